// kernels_util.h -- k_dynamics and the layout conversion kernels
// Part of ilqr_hip.hip (included inside its anonymous namespace, in this order: kernels_common.h,
// kernels_backward.h, kernels_forward.h, kernels_util.h); not a stand-alone header.
// ------------------------------------------------------------------------------------------------
// k_dynamics: the lambdified callables evaluated at `count` points (array-of-structs in and out)
// ------------------------------------------------------------------------------------------------
template <class S>
__global__ void __launch_bounds__(64) k_dynamics(int which, long long count, const double* __restrict__ xs,
                                                 const double* __restrict__ us, double dt, const double* __restrict__ ps,
                                                 double* __restrict__ out)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double x[n], u[m], p[np];
#pragma unroll
    for (int q = 0; q < n; ++q) x[q] = xs[i * n + q];
#pragma unroll
    for (int q = 0; q < m; ++q) u[q] = us[i * m + q];
#pragma unroll
    for (int q = 0; q < np; ++q) p[q] = ps[q];
    if (which == 0) {
        double o[n];
        S::f(x, u, dt, p, o);
#pragma unroll
        for (int q = 0; q < n; ++q) out[i * n + q] = o[q];
    } else if (which == 1) {
        double o[n * n];
        S::A(x, u, dt, p, o);
#pragma unroll
        for (int q = 0; q < n * n; ++q) out[i * n * n + q] = o[q];
    } else {
        double o[n * m];
        S::B(x, u, dt, p, o);
#pragma unroll
        for (int q = 0; q < n * m; ++q) out[i * n * m + q] = o[q];
    }
}

// ------------------------------------------------------------------------------------------------
// layout conversion.  ref: (B, T, C) with problem stride bstride (0 = one shared problem), tiled:
// [tile][T][C][64].  Padding problems (b >= B) replicate problem B-1 so every lane computes on valid data.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_pack(const T* __restrict__ ref, T* __restrict__ tiled, long long B, long long Bpad,
                                              long long TCcopy, long long TCdst, long long bstride)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bpad * TCcopy) return;
    const int lane = (int)(idx & 63);
    const long long rest = idx >> 6;
    const long long tc = rest % TCcopy, tile = rest / TCcopy;
    long long b = tile * TILE + lane;
    if (b >= B) b = B - 1;
    tiled[(tile * TCdst + tc) * TILE + lane] = ref[b * bstride + tc];
}
template <typename T>
__global__ void __launch_bounds__(256) k_unpack(const T* __restrict__ tiled, T* __restrict__ ref, long long B, long long Bpad,
                                                long long TC)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bpad * TC) return;
    const int lane = (int)(idx & 63);
    const long long rest = idx >> 6;
    const long long tc = rest % TC, tile = rest / TC;
    const long long b = tile * TILE + lane;
    if (b < B) ref[b * TC + tc] = tiled[idx];
}
__global__ void __launch_bounds__(256) k_fill_i32(int* p, long long cnt, int v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) p[i] = v;
}
__global__ void __launch_bounds__(256) k_fill_f64(double* p, long long cnt, double v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) p[i] = v;
}

