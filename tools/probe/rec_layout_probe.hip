// rec_layout_probe.hip -- would a PROBLEM-MAJOR copy of the line search's records pay?  (development aid, round 3)
// In the tiled layout [tile][t][component][64 problems] a 128-byte line holds one component of 16 ADJACENT problems, so the
// compacted rounds of the line search fetch nearly every line whatever share of the problems still searches
// (profiles/r3_{acrobot_big,c4}_pmc_summary.json).  A copy REC[problem][t][10 doubles] would make every line one problem's.
// Two questions, both answered here with the real sizes:
//   (1) read side: a wave of 16 problems x 4 lanes streaming the records of a random share f of the problems (ascending
//       list), `work` dependent FMAs per step, from (a) the tiled layout, cooperative 8-byte loads as closed_loop_gather
//       issues them, (b) REC with 8-byte loads, (c) REC with 16-byte loads (4 steps = 320 bytes = 5 loads per lane);
//   (2) write side: the producer (one lane per problem, walking the horizon backwards like k_backward, `work` FMAs per
//       step) writing (a) 5 tiled rows per step as today, (b) today's rows AND the 80-byte record into REC (5 x 16 bytes
//       per lane and step, every lane its own line).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/rec_layout_probe tools/probe/rec_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int R = 10, PF = 4;

__global__ void __launch_bounds__(256, 2) k_read_tiled(const double* __restrict__ src, const int* __restrict__ list, int cnt,
                                                       double* sink, int T, int work)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if ((long long)wave * 16 >= cnt) return;
    const long long g = (long long)wave * 16 + (lane & 15);
    const int b = list[g < cnt ? g : cnt - 1];
    const int c = lane >> 4;
    const double* p = src + (size_t)(b >> 6) * T * R * 64 + (b & 63) + (size_t)c * 64;
    const bool third = c + 8 < R;
    double s = 0.0, acc = 1.0, v[PF][3];
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

__global__ void __launch_bounds__(256, 2) k_read_rec8(const double* __restrict__ rec, const int* __restrict__ list, int cnt,
                                                      double* sink, int T, int work)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if ((long long)wave * 16 >= cnt) return;
    const long long g = (long long)wave * 16 + (lane >> 2);                     // 4 adjacent lanes = one problem: 32 contiguous bytes
    const int b = list[g < cnt ? g : cnt - 1];
    const int c = lane & 3;
    const double* p = rec + (size_t)b * T * R + c;
    const bool third = c + 8 < R;
    double s = 0.0, acc = 1.0, v[PF][3];
    auto fetch = [&](int t, double (&o)[3]) {
        o[0] = p[(size_t)t * R];
        o[1] = p[(size_t)t * R + 4];
        o[2] = third ? p[(size_t)t * R + 8] : 0.0;
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

// 4 steps = 40 doubles = 20 x 16 bytes per problem: lane c of the problem's 4 loads pieces c, c + 4, ..., c + 16
__global__ void __launch_bounds__(256, 2) k_read_rec16(const double* __restrict__ rec, const int* __restrict__ list, int cnt,
                                                       double* sink, int T, int work)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if ((long long)wave * 16 >= cnt) return;
    const long long g = (long long)wave * 16 + (lane >> 2);
    const int b = list[g < cnt ? g : cnt - 1];
    const int c = lane & 3;
    const double2* p = (const double2*)(rec + (size_t)b * T * R) + c;
    double s = 0.0, acc = 1.0;
    double2 v[2][5];
    auto fetch = [&](int t4, double2 (&o)[5]) {                               // steps 4 t4 .. 4 t4 + 3
#pragma unroll
        for (int q = 0; q < 5; ++q) o[q] = p[(size_t)t4 * 20 + 4 * q];
    };
    const int T4 = T / 4;
    fetch(0, v[0]);
    fetch(1 < T4 ? 1 : 0, v[1]);
    for (int t4 = 0; t4 + 2 <= T4; t4 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) u += v[d][q].x + v[d][q].y;
            const int tl = t4 + d + 2 < T4 ? t4 + d + 2 : T4 - 1;
            fetch(tl, v[d]);
            for (int k = 0; k < 4 * work; ++k) acc = __builtin_fma(acc, 1.0000001, u);   // four steps' worth
            s += u;
        }
    }
    if (s + acc == 12345.678) sink[0] = s;
}

template <bool REC>
__global__ void __launch_bounds__(64) k_write(const double* __restrict__ src, double* __restrict__ out, double* __restrict__ rec,
                                              int T, int work)
{
    const int lane = threadIdx.x;
    const size_t tile = blockIdx.x;
    const double* p = src + tile * T * 5 * 64 + lane;
    double* o = out + tile * T * 5 * 64 + lane;
    double2* r = (double2*)(rec + (tile * 64 + lane) * (size_t)T * R);
    double acc = 1.0;
    double x[5], xn[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) x[q] = p[((size_t)(T - 1) * 5 + q) * 64];
    for (int t = T - 1; t >= 0; --t) {
        const int tn = t > 0 ? t - 1 : 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) xn[q] = p[((size_t)tn * 5 + q) * 64];
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) u += x[q];
        for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, u);
#pragma unroll
        for (int q = 0; q < 5; ++q) o[((size_t)t * 5 + q) * 64] = acc + q;
        if constexpr (REC) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r[(size_t)t * 5 + q] = make_double2(q < 3 ? x[2 * q > 4 ? 4 : 2 * q] : acc, acc + q);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) x[q] = xn[q];
    }
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 2 ? atoi(argv[2]) : 3000;
    const int tiles = B / 64;
    const size_t bytes = (size_t)B * T * R * sizeof(double);
    double *tiled, *rec, *half_in, *half_out, *sink;
    int* list;
    CK(hipMalloc(&tiled, bytes)); CK(hipMalloc(&rec, bytes));
    CK(hipMalloc(&half_in, bytes / 2)); CK(hipMalloc(&half_out, bytes / 2));
    CK(hipMalloc(&sink, 8)); CK(hipMalloc(&list, (size_t)B * 4));
    CK(hipMemset(tiled, 0, bytes)); CK(hipMemset(rec, 0, bytes)); CK(hipMemset(half_in, 0, bytes / 2));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%d problems x %d steps x %d doubles: %.2f GB\n", B, T, R, bytes / 1e9);
    std::mt19937 rng(5);
    auto timed = [&](auto&& launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        return best;
    };
    for (int work : {50, 130}) {
        for (double f : {1.0, 0.7, 0.5, 0.25, 0.1}) {
            std::vector<int> h;
            std::uniform_real_distribution<double> U(0, 1);
            for (int i = 0; i < B; ++i) if (f >= 1.0 || U(rng) < f) h.push_back(i);
            const int cnt = (int)h.size();
            CK(hipMemcpy(list, h.data(), (size_t)cnt * 4, hipMemcpyHostToDevice));
            const unsigned grid = (unsigned)((cnt + 63) / 64);               // 4 waves of 16 problems per workgroup
            const float t0 = timed([&] { hipLaunchKernelGGL(k_read_tiled, dim3(grid), dim3(256), 50000, 0, tiled, list, cnt, sink, T, work); });
            const float t1 = timed([&] { hipLaunchKernelGGL(k_read_rec8, dim3(grid), dim3(256), 50000, 0, rec, list, cnt, sink, T, work); });
            const float t2 = timed([&] { hipLaunchKernelGGL(k_read_rec16, dim3(grid), dim3(256), 50000, 0, rec, list, cnt, sink, T, work); });
            const double need = (double)cnt * T * R * 8 / 1e9;
            printf("read  work %3d  share %.2f (%6d problems, %5.2f GB needed): tiled %7.3f ms | REC 8-byte %7.3f ms | REC 16-byte %7.3f ms  (%.1f / %.1f / %.1f TB/s useful)\n",
                   work, f, cnt, need, t0, t1, t2, need / t0, need / t1, need / t2);
        }
    }
    for (int work : {0, 300, 700}) {
        const float t0 = timed([&] { hipLaunchKernelGGL(k_write<false>, dim3(tiles), dim3(64), 0, 0, half_in, half_out, rec, T, work); });
        const float t1 = timed([&] { hipLaunchKernelGGL(k_write<true>, dim3(tiles), dim3(64), 0, 0, half_in, half_out, rec, T, work); });
        printf("write work %3d: tiled rows only %7.3f ms | rows + 80-byte records %7.3f ms\n", work, t0, t1);
    }
    return 0;
}
