// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for THIS code's read patterns?  (development aid)
// MI355X_MICROARCH.md: FETCH_SIZE is exact x1/2 for wide coalesced streaming reads on gfx950, "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  Three kernels, each reading a 1 GiB buffer
// (beyond the 256 MiB Infinity Cache) exactly once, laid out like the trajectories: [tile][row][64 doubles]:
//   k_rows    : one wave per tile, lane l reads double l of every row (512-byte rows; k_riccati / k_cost / k_accept pattern)
//   k_segments: the line search's pattern with W = 32 -- a workgroup of 4 waves covers 8 problems, i.e. ONE 64-byte
//               segment of every row; the 32 lanes of a trial group read the same double
//   k_quads   : k_riccati_quad's pattern -- a wave covers 16 problems (128 bytes of a row), 4 lanes share a double
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/fetch_calib tools/probe/fetch_calib.hip
// run  : rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/probe/fetch_calib   (bytes per kernel are printed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int UNROLL = 10;   // loads in flight per lane, like one time step's record

__global__ void __launch_bounds__(64) k_rows(const double* __restrict__ src, double* sink, int rows)
{
    const double* p = src + (size_t)blockIdx.x * rows * 64 + threadIdx.x;
    double s = 0.0;
    for (int t = 0; t < rows; t += UNROLL) {
        double v[UNROLL];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) v[q] = p[(size_t)(t + q) * 64];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) s += v[q];
    }
    if (s == 12345.678) sink[0] = s;
}

__global__ void __launch_bounds__(256) k_segments(const double* __restrict__ src, double* sink, int rows)
{
    // logical workgroup -> block id as in k_linesearch: the 8 workgroups of a tile are 8 block ids apart (same XCD)
    const unsigned lw = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    const unsigned tile = lw / 8, seg = lw % 8;
    const double* p = src + (size_t)tile * rows * 64 + seg * 8 + threadIdx.x / 32;
    double s = 0.0;
    for (int t = 0; t < rows; t += UNROLL) {
        double v[UNROLL];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) v[q] = p[(size_t)(t + q) * 64];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) s += v[q];
    }
    if (s == 12345.678) sink[0] = s;
}

__global__ void __launch_bounds__(64) k_quads(const double* __restrict__ src, double* sink, int rows)
{
    const unsigned tile = blockIdx.x / 4, part = blockIdx.x % 4;
    const double* p = src + (size_t)tile * rows * 64 + part * 16 + threadIdx.x / 4;
    double s = 0.0;
    for (int t = 0; t < rows; t += UNROLL) {
        double v[UNROLL];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) v[q] = p[(size_t)(t + q) * 64];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) s += v[q];
    }
    if (s == 12345.678) sink[0] = s;
}

int main()
{
    const int tiles = 64, rows = 32760;                      // 64 * 32760 * 512 B = 1.07 GB
    const size_t bytes = (size_t)tiles * rows * 64 * 8;
    double *src, *sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(src, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_rows, dim3(tiles), dim3(64), 0, 0, src, sink, rows);
        hipLaunchKernelGGL(k_segments, dim3(tiles * 8), dim3(256), 0, 0, src, sink, rows);
        hipLaunchKernelGGL(k_quads, dim3(tiles * 4), dim3(64), 0, 0, src, sink, rows);
    }
    CK(hipDeviceSynchronize());
    printf("BYTES_PER_KERNEL %zu\n", bytes);
    CK(hipFree(src));
    CK(hipFree(sink));
    return 0;
}
