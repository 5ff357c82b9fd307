// ls_load_probe.hip -- what bounds the line search's record stream at large batches?  (development aid, round 3)
// rocprofv3 on acrobot 65 536 / cartpole 131 072 / pendulum 131 072 showed round 0 of k_linesearch (W = 4) at the SAME
// 2.3e10 wave-level load instructions per second = 2.95 TB/s in all three, VALU 35-58 % busy, insensitive to prefetch depth
// and occupancy.  Two candidates: (1) the vector-memory path's rate for instructions whose lanes share addresses in groups
// (one 128-byte line per 64-lane instruction), (2) the memory system's rate for 128-byte requests.  This probe streams the
// same [tile][t][R rows][64 doubles] layout with R = 10 rows per step, a wave covering 16 problems (a 128-byte quarter of each
// row), in three ways with the SAME bytes and lines:
//   A "shared"  : 10 loads per step, lanes in groups of 4 read the same double            (what k_linesearch<W=4> does)
//   C "coop"    : lane l reads row l/16 (+4, +8) of problem l%16: 3 loads per step, 64 distinct doubles each, the same
//                 ten 128-byte lines (what a cooperative load + LDS hand-off would issue)
//   D "coop+lds": C, and the 160 doubles go through LDS: each lane then reads its problem's 10 doubles back (5 x b128)
// plus E "rows": one wave per tile reading whole 512-byte rows (k_forward's pattern), for the HBM ceiling of this layout.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/ls_load_probe tools/probe/ls_load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int R = 10, PF = 3;

// occupancy like the real kernel: 2 workgroups of 4 waves per CU (LDS-limited)
// (the launches pass 60 000 bytes of dynamic LDS: at most 2 workgroups per CU)

__global__ void __launch_bounds__(256, 2) k_shared(const double* __restrict__ src, double* sink, int T, int work)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const unsigned tile = wave >> 2, part = wave & 3;
    const double* p = src + (size_t)tile * T * R * 64 + part * 16 + lane / 4;
    double s = 0.0, acc = 1.0;
    double v[PF][R];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int q = 0; q < R; ++q) v[d][q] = p[((size_t)d * R + q) * 64];
    for (int t = 0; t + PF <= T; t += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) u += v[d][q];
            const int tl = t + d + PF < T ? t + d + PF : T - 1;
#pragma unroll
            for (int q = 0; q < R; ++q) v[d][q] = p[((size_t)tl * R + q) * 64];
            for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);      // dependent fp64 chain per step
            s += u;
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

__global__ void __launch_bounds__(256, 2) k_coop(const double* __restrict__ src, double* sink, int T, int work)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const unsigned tile = wave >> 2, part = wave & 3;
    const double* p = src + (size_t)tile * T * R * 64 + part * 16 + (lane & 15) + (size_t)(lane >> 4) * 64;
    const bool third = (lane >> 4) + 8 < R;
    double s = 0.0, acc = 1.0;
    double v[PF][3];
    auto fetch = [&](int t, double (&o)[3]) {
        o[0] = p[(size_t)t * R * 64];
        o[1] = p[(size_t)t * R * 64 + 4 * 64];
        o[2] = third ? p[(size_t)t * R * 64 + 8 * 64] : 0.0;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, v[d]);
    for (int t = 0; t + PF <= T; t += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const double u = v[d][0] + v[d][1] + v[d][2];
            const int tl = t + d + PF < T ? t + d + PF : T - 1;
            fetch(tl, v[d]);
            for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
            s += u;
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

__global__ void __launch_bounds__(256, 2) k_coop_lds(const double* __restrict__ src, double* sink, int T, int work)
{
    __shared__ double sh[4][2][16 * R + 8];
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
    const unsigned tile = wave >> 2, part = wave & 3;
    const double* p = src + (size_t)tile * T * R * 64 + part * 16 + (lane & 15) + (size_t)(lane >> 4) * 64;
    const bool third = (lane >> 4) + 8 < R;
    const int g = lane >> 2;                                    // the problem this lane works on
    double s = 0.0, acc = 1.0;
    double v[PF][3];
    auto fetch = [&](int t, double (&o)[3]) {
        o[0] = p[(size_t)t * R * 64];
        o[1] = p[(size_t)t * R * 64 + 4 * 64];
        o[2] = third ? p[(size_t)t * R * 64 + 8 * 64] : 0.0;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(d, v[d]);
    for (int t = 0; t + PF <= T; t += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            double* buf = sh[wl][d & 1];
            // record layout in LDS: [problem][row] (stride R): lane (row r = lane >> 4 (+4, +8), problem lane & 15)
            buf[(lane & 15) * R + (lane >> 4)] = v[d][0];
            buf[(lane & 15) * R + (lane >> 4) + 4] = v[d][1];
            if (third) buf[(lane & 15) * R + (lane >> 4) + 8] = v[d][2];
            const int tl = t + d + PF < T ? t + d + PF : T - 1;
            fetch(tl, v[d]);
            __builtin_amdgcn_wave_barrier();
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) u += buf[g * R + q];
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
            s += u;
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

__global__ void __launch_bounds__(64) k_rows(const double* __restrict__ src, double* sink, int T, int work)
{
    const double* p = src + (size_t)blockIdx.x * T * R * 64 + threadIdx.x;
    double s = 0.0, acc = 1.0;
    for (int t = 0; t < T; ++t) {
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) u += p[((size_t)t * R + q) * 64];
        for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
        s += u;
    }
    if (s + acc == 12345.678) sink[0] = s;
}

int main(int argc, char** argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 3000;
    const size_t bytes = (size_t)tiles * T * R * 64 * sizeof(double);
    double *src, *sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%d tiles x %d steps x %d rows: %.2f GB per pass\n", tiles, T, R, bytes / 1e9);
    for (int work : {0, 32, 64, 128}) {
        for (int which = 0; which < 4; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(a));
                if (which == 0) hipLaunchKernelGGL(k_shared, dim3(tiles), dim3(256), 60000, 0, src, sink, T, work);
                if (which == 1) hipLaunchKernelGGL(k_coop, dim3(tiles), dim3(256), 60000, 0, src, sink, T, work);
                if (which == 2) hipLaunchKernelGGL(k_coop_lds, dim3(tiles), dim3(256), 60000, 0, src, sink, T, work);
                if (which == 3) hipLaunchKernelGGL(k_rows, dim3(tiles), dim3(64), 0, 0, src, sink, T, work);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            const char* names[] = {"A shared (10 loads/step, 4 lanes per double)", "C coop   (3 loads/step, 64 distinct doubles)",
                                   "D coop + LDS hand-off", "E rows   (one wave per tile, 512-byte rows)"};
            printf("fma chain %3d per step  %-48s %7.3f ms  %7.1f GB/s\n", work, names[which], best, bytes / best / 1e6);
        }
    }
    CK(hipGetLastError());
    return 0;
}
