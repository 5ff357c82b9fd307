// Development probe: does a wave64 FP64 VALU instruction get cheaper when only part of EXEC is set?  (The lone waves of the
// late line-search rounds and of the recursion are bound by FP64 issue: 4 cycles per instruction for 64 lanes.  If the SIMD
// skipped the 16-lane passes whose EXEC bits are all zero, spreading a problem over more, emptier waves would pay.)
// One wave per SIMD (1024 workgroups of 64), a dependent chain of v_fma_f64, EXEC = the low ACTIVE lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ void __launch_bounds__(64) k_chain(double* o, const int active, const int iters, const double c)
{
    const int lane = threadIdx.x;
    double a[CHAINS];
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) a[j] = 1.0 + 1e-3 * lane + j;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int j = 0; j < CHAINS; ++j) a[j] = __builtin_fma(a[j], c, 1e-9);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) s += a[j];
    o[(size_t)blockIdx.x * 64 + lane] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int CHAINS>
int run(double* d, const int blocks, const int active, const int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, active, iters, 0.999999);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_chain<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, active, iters, 0.999999);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)iters * 16 * CHAINS;
    printf("chains %d blocks %5d active lanes %2d: %.3f ms, %.2f ns per FMA instruction of one wave\n", CHAINS, blocks, active, ms,
           ms * 1e6 / insts);
    return 0;
}

int main()
{
    double* d;
    const int max_blocks = 8192;
    CK(hipMalloc(&d, (size_t)max_blocks * 64 * sizeof(double)));
    const int iters = 20000;
    for (int blocks : {1024, 2048, 4096})
        for (int active : {64, 48, 32, 16, 8, 1}) {
            if (run<1>(d, blocks, active, iters)) return 2;
            if (run<4>(d, blocks, active, iters)) return 2;
        }
    CK(hipFree(d));
    return 0;
}
