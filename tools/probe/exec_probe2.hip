// Development probe (second part of exec_probe.hip): FP64 FMA issue time of one wave against (a) WHICH lanes are enabled
// and (b) how many waves the chip holds (64 ... 1024 workgroups of one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k_chain(double* o, const unsigned long long mask, const int iters, const double c)
{
    const int lane = threadIdx.x;
    double a0 = 1.0 + 1e-3 * lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a0 = __builtin_fma(a0, c, 1e-9); a1 = __builtin_fma(a1, c, 1e-9);
                a2 = __builtin_fma(a2, c, 1e-9); a3 = __builtin_fma(a3, c, 1e-9);
            }
        }
    }
    o[(size_t)blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main()
{
    double* d;
    CK(hipMalloc(&d, (size_t)4096 * 64 * sizeof(double)));
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long m; } pats[] = {
        {"all 64", ~0ull}, {"lanes 0-15", 0xffffull}, {"lanes 0-7", 0xffull}, {"lanes 0-3 of each row", 0x000f000f000f000full},
        {"lanes 0-7 of each row", 0x00ff00ff00ff00ffull}, {"lane 0 of each row", 0x0001000100010001ull}, {"even lanes", 0x5555555555555555ull},
        {"lanes 0-8", 0x1ffull}, {"lanes 0-11", 0xfffull}};
    for (int blocks : {64, 256, 1024})
        for (auto& p : pats) {
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, p.m, iters, 0.999999);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, d, p.m, iters, 0.999999);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks %4d  %-24s %.3f ms  %.2f ns per FMA instruction\n", blocks, p.name, ms, ms * 1e6 / ((double)iters * 64));
        }
    CK(hipFree(d));
    return 0;
}
