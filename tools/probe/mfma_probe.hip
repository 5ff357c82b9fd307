// mfma_probe.hip -- does the FP64 matrix core buy anything for the n = 12 recursion?  (development aid, round 3; the decision
// DESIGN.md section 3 quotes.)  BASELINE's north star asks for the no-MFMA choice to be justified by measurement.  The n = 12
// backward recursion (k_riccati_grp, ilqr.py:99-103,124-125) is a chain of 12 x 12 x 12 fp64 products per time step, each
// step's value function feeding the next: V_t = f(A_t' V_{t+1} A_t, ...).  This probe runs ONLY that product chain --
//     W = A' V ;  Q = W A ;  V <- Q            (two 12 x 12 x 12 products per step, 400 dependent steps)
// for 32 768 problems, four problems per wave (one 16-lane DPP row each, as k_riccati_grp maps them), in three forms:
//   dpalu  : lane r of a row holds row r of A and column r of V; the operand every lane of the row needs comes through
//            v_fmac_f64_dpp row_newbcast, four instructions behind one s_nop -- k_riccati_grp's form, DENSE (the real kernel
//            skips the structural zeros of A_t: 63 of 144 entries are general for the quadcopter)
//   mfma4  : v_mfma_f64_4x4x4_4b_f64, block b of the instruction = problem b; a lane holds entry (i, k) of each of the nine
//            4 x 4 blocks of a matrix, so the nine accumulators of a product ARE the operand registers of the next product:
//            no data movement at all between the products (optimistic: the real recursion needs A' V, W A, B' V, ... whose
//            operand roles do not all line up like this, and would need transposes through LDS)
//   mfma16 : v_mfma_f64_16x16x4_f64, one problem per instruction zero-padded 12 -> 16, K in three steps of 4, four problems
//            one after the other; accumulator registers reused as operands as they lie (again: no shuffles charged)
// Same clocks, same occupancy (2 waves per SIMD), rocprofv3 --kernel-trace for all three -> profiles/r3_mfma_probe.*
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_probe tools/probe/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

#define DPP_FMAC(acc, src, own, lane) "v_fmac_f64_dpp " acc ", " src ", " own " row_newbcast:" lane " row_mask:0xf bank_mask:0xf"
template <int L>
__device__ __forceinline__ void fmac4(double& a0, double& a1, double& a2, double& a3, const double s0, const double s1,
                                      const double s2, const double s3, const double o)
{
    asm("s_nop 1\n\t" DPP_FMAC("%0", "%4", "%8", "%9") "\n\t" DPP_FMAC("%1", "%5", "%8", "%9") "\n\t"
        DPP_FMAC("%2", "%6", "%8", "%9") "\n\t" DPP_FMAC("%3", "%7", "%8", "%9")
        : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(o), "n"(L));
}
template <int R>
__device__ __forceinline__ void row_times(double (&acc)[12], const double (&src)[12], const double (&own)[12])
{
    // acc[i] += (lane R's src[i]) * own[R] for i < 12
    fmac4<R>(acc[0], acc[1], acc[2], acc[3], src[0], src[1], src[2], src[3], own[R]);
    fmac4<R>(acc[4], acc[5], acc[6], acc[7], src[4], src[5], src[6], src[7], own[R]);
    fmac4<R>(acc[8], acc[9], acc[10], acc[11], src[8], src[9], src[10], src[11], own[R]);
    if constexpr (R + 1 < 12) row_times<R + 1>(acc, src, own);
}

__global__ void __launch_bounds__(64, 2) k_dpalu(const double* a_in, double* out, int steps)
{
    const int lane = threadIdx.x, r = lane & 15;
    double A[12], V[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { A[i] = r < 12 ? a_in[(r * 12 + i) % 144] : 0.0; V[i] = i == r ? 1.0 : 0.0; }
    for (int t = 0; t < steps; ++t) {
        double W[12], Q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { W[i] = 0.0; Q[i] = 0.0; }
        row_times<0>(W, A, V);              // W[i] = sum_r A[r][i] V[r][j]                    (A' V)[:, j]
        row_times<0>(Q, W, A);              // the second product: the first one's result as the broadcast operand
#pragma unroll
        for (int i = 0; i < 12; ++i) V[i] = Q[i];
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += V[i];
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

template <int L>
__device__ __forceinline__ void fmac1(double& a0, const double s0, const double o)
{
    asm("s_nop 1\n\t" DPP_FMAC("%0", "%1", "%2", "%3") : "+&v"(a0) : "v"(s0), "v"(o), "n"(L));
}
// the same with the quadcopter's share of general entries: 63 of the 144 entries of A_t (5 or 6 per row; the others are
// structural zeros / ones / dt that k_riccati_grp skips or adds) -- 126 broadcast-FMAs per step instead of 288
template <int R>
__device__ __forceinline__ void row_times_sparse(double (&acc)[12], const double (&src)[12], const double (&own)[12])
{
    constexpr int o = (R * 5) % 7;                          // where this row's run of general entries starts
    fmac4<R>(acc[o], acc[o + 1], acc[o + 2], acc[o + 3], src[o], src[o + 1], src[o + 2], src[o + 3], own[R]);
    fmac1<R>(acc[o + 4], src[o + 4], own[R]);
    if constexpr (R == 8 || R == 9 || R == 10) fmac1<R>(acc[(o + 5) % 12], src[(o + 5) % 12], own[R]);
    if constexpr (R + 1 < 12) row_times_sparse<R + 1>(acc, src, own);
}
__global__ void __launch_bounds__(64, 2) k_dpalu_sparse(const double* a_in, double* out, int steps)
{
    const int lane = threadIdx.x, r = lane & 15;
    double A[12], V[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { A[i] = r < 12 ? a_in[(r * 12 + i) % 144] : 0.0; V[i] = i == r ? 1.0 : 0.0; }
    for (int t = 0; t < steps; ++t) {
        double W[12], Q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { W[i] = V[i]; Q[i] = 0.0; }     // (unit entries of A_t: plain adds, here one per column)
        row_times_sparse<0>(W, A, V);
#pragma unroll
        for (int i = 0; i < 12; ++i) Q[i] = W[i];
        row_times_sparse<0>(Q, W, A);
#pragma unroll
        for (int i = 0; i < 12; ++i) V[i] = 0.5 * Q[i];
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += V[i];
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

__global__ void __launch_bounds__(64, 2) k_mfma4(const double* a_in, double* out, int steps)
{
    const int lane = threadIdx.x, e = lane & 15;
    double A[3][3], V[3][3];
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) { A[I][J] = a_in[((4 * I + (e & 3)) * 12 + 4 * J + (e >> 2)) % 144]; V[I][J] = (I == J && (e & 3) == (e >> 2)) ? 1.0 : 0.0; }
    for (int t = 0; t < steps; ++t) {
        double W[3][3], Q[3][3];
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J < 3; ++J) {
                double acc = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[K][I], V[K][J], acc, 0, 0, 0);
                W[I][J] = acc;
            }
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J < 3; ++J) {
                double acc = 0.0;
#pragma unroll
                for (int K = 0; K < 3; ++K) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(W[I][K], A[K][J], acc, 0, 0, 0);
                Q[I][J] = acc;
            }
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J < 3; ++J) V[I][J] = Q[I][J];
    }
    double s = 0.0;
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) s += V[I][J];
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

__global__ void __launch_bounds__(64, 2) k_mfma16(const double* a_in, double* out, int steps)
{
    const int lane = threadIdx.x;
    double A[3], V[4][3];                                   // K-steps of the operands; four problems one after the other
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[k] = (lane & 15) < 12 ? a_in[(lane * 3 + k) % 144] : 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) V[p][k] = (lane & 15) == 4 * k + (lane >> 4) ? 1.0 : 0.0;
    }
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            d4 W = {0.0, 0.0, 0.0, 0.0}, Q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 3; ++k) W = __builtin_amdgcn_mfma_f64_16x16x4f64(A[k], V[p][k], W, 0, 0, 0);
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(W.x, A[0], Q, 0, 0, 0);
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(W.y, A[1], Q, 0, 0, 0);
            Q = __builtin_amdgcn_mfma_f64_16x16x4f64(W.z, A[2], Q, 0, 0, 0);
            V[p][0] = Q.x; V[p][1] = Q.y; V[p][2] = Q.z + Q.w;
        }
    }
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) s += V[p][0] + V[p][1] + V[p][2];
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 8192, steps = argc > 2 ? atoi(argv[2]) : 400;
    double h[144];
    for (int i = 0; i < 144; ++i) h[i] = 0.28 * ((i * 37 % 11) - 5) / 5.0;   // |A| small enough that V stays finite
    double *a, *out;
    CK(hipMalloc(&a, sizeof(h)));
    CK(hipMalloc(&out, (size_t)waves * 64 * 8));
    CK(hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"dpalu  (v_fmac_f64_dpp row_newbcast, dense, 288 per step)", "mfma4  (v_mfma_f64_4x4x4_4b, 54 per step)",
                           "mfma16 (v_mfma_f64_16x16x4, 24 per step, 4 problems in turn)",
                           "dpalu, structural zeros skipped as k_riccati_grp does (126 per step)"};
    printf("%d waves x 4 problems, %d dependent steps, two 12x12x12 products per step (%.1f GFLOP useful)\n", waves, steps,
           waves * 4.0 * steps * 2 * 2 * 1728 / 1e9);
    for (int which = 0; which < 4; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_dpalu, dim3(waves), dim3(64), 0, 0, a, out, steps);
            if (which == 1) hipLaunchKernelGGL(k_mfma4, dim3(waves), dim3(64), 0, 0, a, out, steps);
            if (which == 2) hipLaunchKernelGGL(k_mfma16, dim3(waves), dim3(64), 0, 0, a, out, steps);
            if (which == 3) hipLaunchKernelGGL(k_dpalu_sparse, dim3(waves), dim3(64), 0, 0, a, out, steps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double useful = waves * 4.0 * steps * 2 * 2 * 1728;
        printf("%-70s %8.3f ms  %6.1f SIMD-ns per problem and step  %5.1f TFLOP/s dense-equivalent\n", names[which], best,
               best * 1e6 * 1024 / (waves * 4.0 * steps), useful / best / 1e9);
    }
    CK(hipGetLastError());
    return 0;
}
