// Development probe: semantics of v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast on gfx950 (used by k_riccati_grp).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int L>
__device__ __forceinline__ void fmac_bcast16(double& acc, double a_src, double b)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a_src), "v"(b), "n"(L));
}
template <int L>
__device__ __forceinline__ double mov_bcast16(double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
__global__ void k(double* o)
{
    const int i = threadIdx.x;
    double x = 100.0 + i, y = 2.0 + 0.001 * i;
    double acc = 0.5;
    fmac_bcast16<3>(acc, x, y);        // 0.5 + x[row*16+3] * y[i]
    fmac_bcast16<11>(acc, x, y);       // + x[row*16+11] * y[i]
    o[i] = acc;
    o[64 + i] = mov_bcast16<7>(x);
}
int main()
{
    double* d; double h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int r = i & ~15;
        const double y = 2.0 + 0.001 * i;
        const double e = __builtin_fma(100.0 + r + 11, y, __builtin_fma(100.0 + r + 3, y, 0.5));
        if (h[i] != e) { ++bad; printf("lane %d fmac got %.17g want %.17g\n", i, h[i], e); }
        if (h[64 + i] != 100.0 + r + 7) { ++bad; printf("lane %d mov got %g want %g\n", i, h[64 + i], 100.0 + r + 7); }
    }
    printf("dpp probe: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
