// stream_layout_probe.hip -- do the lane-per-problem kernels lose HBM efficiency because a wave streams FOUR arrays?
// (development aid, round 3)  k_forward<accept> reads x, u, K, k (n = 4, m = 1: 4 + 1 + 4 + 1 rows of 64 doubles per step)
// from four separate tiled arrays and writes x, u back: 4.85 TB/s on 65 536 acrobot problems, one wave per SIMD.  Here the same
// rows are streamed (a) from four arrays as today, (b) from ONE array with the ten rows of a step adjacent
// ([tile][t][10][64]), same bytes, same prefetch depth, same amount of arithmetic per step, with and without the write-back.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/stream_layout_probe tools/probe/stream_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int PF = 3;

template <bool WRITE>
__global__ void __launch_bounds__(64) k_four(double* __restrict__ X, double* __restrict__ U, const double* __restrict__ K,
                                             const double* __restrict__ kf, double* sink, int T, int work)
{
    const size_t tile = blockIdx.x;
    const int lane = threadIdx.x;
    double* Xp = X + tile * (size_t)T * 4 * 64 + lane;
    double* Up = U + tile * (size_t)T * 64 + lane;
    const double* Kp = K + tile * (size_t)T * 4 * 64 + lane;
    const double* kp = kf + tile * (size_t)T * 64 + lane;
    double v[PF][10], acc = 1.0, s = 0.0;
    auto fetch = [&](int t, double (&o)[10]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = Xp[((size_t)t * 4 + q) * 64];
        o[4] = Up[(size_t)t * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[5 + q] = Kp[((size_t)t * 4 + q) * 64];
        o[9] = kp[(size_t)t * 64];
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, v[d]);
    for (int t = 0; t + PF <= T; t += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < 10; ++q) u += v[d][q];
            const int tl = t + d + PF < T ? t + d + PF : T - 1;
            fetch(tl, v[d]);
            for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
            s += u;
            if constexpr (WRITE) {
                if (t + d > 0) {                               // rows of step t + d - 1: behind every row still to be read
#pragma unroll
                    for (int q = 0; q < 4; ++q) Xp[((size_t)(t + d - 1) * 4 + q) * 64] = acc + q;
                    Up[(size_t)(t + d - 1) * 64] = acc;
                }
            }
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

template <bool WRITE>
__global__ void __launch_bounds__(64) k_one(double* __restrict__ R, double* sink, int T, int work)
{
    const size_t tile = blockIdx.x;
    const int lane = threadIdx.x;
    double* Rp = R + tile * (size_t)T * 10 * 64 + lane;
    double v[PF][10], acc = 1.0, s = 0.0;
    auto fetch = [&](int t, double (&o)[10]) {
#pragma unroll
        for (int q = 0; q < 10; ++q) o[q] = Rp[((size_t)t * 10 + q) * 64];
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, v[d]);
    for (int t = 0; t + PF <= T; t += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < 10; ++q) u += v[d][q];
            const int tl = t + d + PF < T ? t + d + PF : T - 1;
            fetch(tl, v[d]);
            for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
            s += u;
            if constexpr (WRITE) {
                if (t + d > 0) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) Rp[((size_t)(t + d - 1) * 10 + q) * 64] = acc + q;
                }
            }
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

int main(int argc, char** argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 3000;
    const size_t row = (size_t)tiles * T * 64 * sizeof(double);
    double *X, *U, *K, *kf, *R, *sink;
    CK(hipMalloc(&X, 4 * row)); CK(hipMalloc(&U, row)); CK(hipMalloc(&K, 4 * row)); CK(hipMalloc(&kf, row));
    CK(hipMalloc(&R, 10 * row)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(X, 0, 4 * row)); CK(hipMemset(U, 0, row)); CK(hipMemset(K, 0, 4 * row)); CK(hipMemset(kf, 0, row));
    CK(hipMemset(R, 0, 10 * row));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%d tiles x %d steps: %.2f GB read per pass, %.2f GB written with the write-back\n", tiles, T, 10 * row / 1e9, 5 * row / 1e9);
    auto timed = [&](auto&& launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        return best;
    };
    for (int work : {0, 60, 130}) {
        const float f0 = timed([&] { hipLaunchKernelGGL(k_four<false>, dim3(tiles), dim3(64), 0, 0, X, U, K, kf, sink, T, work); });
        const float o0 = timed([&] { hipLaunchKernelGGL(k_one<false>, dim3(tiles), dim3(64), 0, 0, R, sink, T, work); });
        const float f1 = timed([&] { hipLaunchKernelGGL(k_four<true>, dim3(tiles), dim3(64), 0, 0, X, U, K, kf, sink, T, work); });
        const float o1 = timed([&] { hipLaunchKernelGGL(k_one<true>, dim3(tiles), dim3(64), 0, 0, R, sink, T, work); });
        printf("work %3d  read only: four arrays %7.3f ms (%.2f TB/s) | one array %7.3f ms (%.2f TB/s)   read + write-back: four %7.3f ms (%.2f TB/s) | one %7.3f ms (%.2f TB/s)\n",
               work, f0, 10 * row / f0 / 1e9, o0, 10 * row / o0 / 1e9, f1, 15 * row / f1 / 1e9, o1, 15 * row / o1 / 1e9);
    }
    return 0;
}
