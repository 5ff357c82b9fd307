// ilqr_hip.hip -- batched iLQR / LTV-LQR hot path for AMD MI355X (gfx950, CDNA4), fp64.
//
// What is here (reference = Kongx231/ilqr, python/ilqr.py and python/ltvlqr.py):
//   k_rollout     ilqr.rollout            ilqr.py:30-47  (+ compute_cost :49-62 fused)
//   k_backward    ilqr.backwards_pass     ilqr.py:64-135 (+ the "optimal" stop test :175)
//   k_linesearch  the while loop of solve ilqr.py:180-202: W line-search trials per problem run concurrently
//                 on W adjacent lanes, the first accepted trial is found with a ballot over the lane group
//   k_forward     ilqr.forwards_pass      ilqr.py:137-158 (also the accept step :194-199, re-rolling the winner)
//   k_ltvlqr      ltvlqr                  ltvlqr.py:3-28
//   k_dynamics    the lambdified f/A/B    symbolic_dynamics.py:23-25,88-90
//   k_pack/k_unpack  reference layout (B,T,C) <-> tiled SoA [b/64][T][C][b%64]
//
// Design: problems are independent, the horizon is sequential.  One lane owns one problem in
// the backward / forward / ltvlqr kernels, with V_xx, V_x, A, B, Q_* held in VGPRs (n <= 4: no LDS, no
// cross-lane traffic); a wave streams its own contiguous tile of 64 problems so every global access is a
// 512-byte coalesced row.  No MFMA: the products are n x n with n <= 12, fp64, and serially dependent in time.
// The translation unit is compiled with -ffp-contract=off; fused multiply-adds appear only where written
// (__builtin_fma in the small dense products), never inside the generated dynamics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/ilqr_hip.h"

#include "fast_trig.h"

#define ILQR_DYN_FN __device__ __forceinline__
#define ILQR_DYN_CONST static constexpr
#define ILQR_DYN_SINCOS(x, s, c) ilqr_sincos_fast((x), (s), (c))
#define ILQR_DYN_SINCOS_SLOW(x, s, c) do { const ilqr_sc sc_ = ilqr_sincos_slow(x); (s) = sc_.sinv; (c) = sc_.cosv; } while (0)
#define ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c) ilqr_sincos_sum((a), (b), (sa), (ca), (sb), (cb), (s), (c))
#define ILQR_DYN_TRIG_BIG(x) (!(__builtin_fabs(x) < 1.0e6))
#define ILQR_DYN_TAN(x) tan(x)
#define ILQR_DYN_TAN_SC(x, s, c) ilqr_div_shared((s), (c), ilqr_rcp(c))   // tan from the pair f computes anyway (<= 2 ulp)
#define ILQR_DYN_RCP(d) ilqr_rcp(d)
#define ILQR_DYN_DIV(a, d, rd) ilqr_div_shared((a), (d), (rd))
#include "dynamics/pendulum.h"
#include "dynamics/acrobot.h"
#include "dynamics/cartpole.h"
#include "dynamics/quadcopter.h"
#include "dynamics/car.h"
#include "dynamics/double_integrator.h"
#include "dynamics/double_pendulum.h"

namespace {

constexpr int TILE = 64;          // problems per tile = lanes per wavefront
constexpr int MAX_TRIALS = 64;    // line-search trials kept in the alpha table
constexpr int N_KERNEL_IDS = 8;

// ------------------------------------------------------------------------------------------------
// systems
// ------------------------------------------------------------------------------------------------
#define ILQR_DEFINE_SYSTEM(NAME, UP)                                                                              \
    struct Sys_##NAME {                                                                                           \
        static constexpr int n = ILQR_DYN_##UP##_N, m = ILQR_DYN_##UP##_M, np = ILQR_DYN_##UP##_NP;               \
        static constexpr int id = ILQR_DYN_##UP##_ID;                                                             \
        static constexpr const char* A_pat = ilqr_dyn_##NAME##_A_pat;                                             \
        static constexpr const char* B_pat = ilqr_dyn_##NAME##_B_pat;                                             \
        static __device__ __forceinline__ char a_kind(int i) { return ilqr_dyn_##NAME##_A_pat[i]; }               \
        static __device__ __forceinline__ char b_kind(int i) { return ilqr_dyn_##NAME##_B_pat[i]; }               \
        static __device__ __forceinline__ void f(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_f(x, u, dt, p, o); }              \
        static __device__ __forceinline__ void A(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_A(x, u, dt, p, o); }              \
        static __device__ __forceinline__ void B(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_B(x, u, dt, p, o); }              \
    };
ILQR_DEFINE_SYSTEM(pendulum, PENDULUM)
ILQR_DEFINE_SYSTEM(acrobot, ACROBOT)
ILQR_DEFINE_SYSTEM(cartpole, CARTPOLE)
ILQR_DEFINE_SYSTEM(quadcopter, QUADCOPTER)
ILQR_DEFINE_SYSTEM(car, CAR)
ILQR_DEFINE_SYSTEM(double_integrator, DOUBLE_INTEGRATOR)
ILQR_DEFINE_SYSTEM(double_pendulum, DOUBLE_PENDULUM)

struct SysInfo { const char* name; int n, m, np; };
constexpr int N_SYSTEMS = 7;
const SysInfo SYSTEMS[N_SYSTEMS] = {{"pendulum", 2, 1, 3}, {"acrobot", 4, 1, 5}, {"cartpole", 4, 1, 4}, {"quadcopter", 12, 4, 8},
                                    {"car", 4, 2, 1}, {"double_integrator", 2, 1, 1}, {"double_pendulum", 4, 2, 5}};

// ------------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------------
struct KArgs {
    int N;                 // time steps
    int it;                // outer iteration index (trace row)
    int n_iter;            // rows in trace
    int zero_first;
    int n_trials;          // line-search trials with alpha > low_learning_rate
    int state_rows;        // rows per problem in X (N+1)
    long long n_tiles;     // tiles this launch covers (a sub-batch of the workspace)
    long long tile0;       // first tile of the sub-batch
    double dt, low_red, armijo_thr, quu_reg;
    double* X;             // [tile][state_rows][n][64]
    double* U;             // [tile][N][m][64]
    const double* G;       // [tile][N][m][64] initial guess (read by k_rollout only)
    double* KFB;           // [tile][N][m*n][64]
    double* KFF;           // [tile][N][m][64]
    double* X2;            // split pipeline: accepted trajectory before it is copied back, same layout as X / U
    double* U2;
    double* CKPT;          // [problem][segment][n][64 trial slots] line-search checkpoints
    double* LIN;           // [tile][N][nA_x + nB_x][64] general entries of A_t, B_t
    int ck;                // checkpoint interval (time steps per segment)
    int n_seg;             // segments = ceil(N / ck)
    double* XO;            // forward outputs (may alias X/U)
    double* UO;
    const double* INIT;    // [tile][n][64]
    const double* TARGET;  // [tile][n][64]
    const double* PARAMS;  // [tile][np][64]
    const double* WTS;     // Qk[n*n], Rk[m*m], QT[n*n]
    const double* alphas;  // [MAX_TRIALS] alpha_j by repeated multiplication
    const double* alpha_pp;// per-problem alpha (forward API) or nullptr
    double* cost;          // [Bpad]
    double* new_cost;      // [Bpad]
    double* dJ;            // [tile][3][64]: total, grad, hess
    int* active;           // [Bpad]
    int* iters;
    int* status;
    int* acc_j;
    int* ls_rounds;
    int ls_round;          // line search: round r evaluates trials [r*W, r*W + W)
    int* ls_list;          // [2][Bpad] problems still searching after a round (ping-pong by round parity)
    int* ls_count;         // [MAX_TRIALS] length of the list each round leaves (zeroed before round 0)
    int* accepts;
    double* trace;         // [tile][n_iter][6][64] or nullptr
    long long Bpad;
};

// ------------------------------------------------------------------------------------------------
// small dense helpers.  Every loop is fully unrolled so all arrays live in VGPRs; `kind` is the
// compile-time sparsity class of the Jacobian entry ('0' zero, '1' one, 'x' general): after unrolling
// the tests fold away and structurally zero / unit entries cost nothing.  Skipping an exact-zero product
// or multiplying by an exact one does not change any rounding.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void acc_term(double& s, bool& first, char kind, double a, double b)
{
    if (kind == '0') return;
    if (kind == '1') s = first ? b : s + b;
    else s = first ? a * b : __builtin_fma(a, b, s);
    first = false;
}

template <class S>
struct Riccati {
    static constexpr int n = S::n, m = S::m;

    // inverse of the m x m matrix Q (row-major).  m == 1: reciprocal (what LAPACK getri does for 1x1).
    // m > 1: Cholesky L L^T, inv = L^-T L^-1.  Returns false when Q is not positive definite / not finite.
    static __device__ __forceinline__ bool inv_spd(const double (&Q)[m * m], double (&Qi)[m * m])
    {
        if constexpr (m == 1) {
            Qi[0] = 1.0 / Q[0];
            return Q[0] != 0.0;
        } else {
            // 1 / L_jj is computed once per pivot (IEEE division) and reused for the diagonal of L^-1.  The off-diagonal
            // divisions stay divisions: replacing them by reciprocal multiplies moved the quadcopter's final K from
            // < 1e-8 to 2.5e-8 of the reference after 21 iterations (the inverse is where rounding is amplified most).
            double L[m * m], Li[m * m], rinv[m];
            bool ok = true;
#pragma unroll
            for (int j = 0; j < m; ++j) {
                double d = Q[j * m + j];
#pragma unroll
                for (int q = 0; q < j; ++q) d = __builtin_fma(-L[j * m + q], L[j * m + q], d);
                ok = ok && (d > 0.0);
                const double ljj = sqrt(d);
                L[j * m + j] = ljj;
                const double r = 1.0 / ljj;
                rinv[j] = r;
#pragma unroll
                for (int i = j + 1; i < m; ++i) {
                    double s = Q[i * m + j];
#pragma unroll
                    for (int q = 0; q < j; ++q) s = __builtin_fma(-L[i * m + q], L[j * m + q], s);
                    L[i * m + j] = s * r;
                }
            }
            // Li = L^-1 (lower triangular)
#pragma unroll
            for (int j = 0; j < m; ++j) {
                Li[j * m + j] = rinv[j];
#pragma unroll
                for (int i = j + 1; i < m; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int q = j; q < i; ++q) s = __builtin_fma(-L[i * m + q], Li[q * m + j], s);
                    Li[i * m + j] = s / L[i * m + i];
                }
            }
            // inv = Li^T Li
#pragma unroll
            for (int i = 0; i < m; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int q = i; q < m; ++q) s = __builtin_fma(Li[q * m + i], Li[q * m + j], s);
                    Qi[i * m + j] = s;
                    Qi[j * m + i] = s;
                }
            return ok;
        }
    }

    // One step of ilqr.backwards_pass (ilqr.py:84-125).  A, Bm: Jacobians at (x_t,u_t).  Updates Vx, Vxx in place,
    // returns k, K and the step's (grad, hess) expected-reduction terms.
    static __device__ __forceinline__ bool step(const double (&A)[n * n], const double (&Bm)[n * m], const double (&u)[m],
                                                const double (&Qk)[n * n], const double (&Rk)[m * m], double quu_reg,
                                                double (&Vx)[n], double (&Vxx)[n * n], double (&k)[m], double (&K)[m * n],
                                                double& g_out, double& h_out)
    {
        double Qx[n], Qu[m], BtV[m * n], Qux[m * n], Quu[m * m], Qxx[n * n], Quu_inv[m * m];
        // :99  Q_x = A.T @ V_x   (l_x = Q_k @ 0 = 0)
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = 0.0; bool first = true;
#pragma unroll
            for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + i), A[r * n + i], Vx[r]);
            Qx[i] = s;
        }
        // :93,:100  Q_u = R_k @ u + B.T @ V_x
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double lu = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) lu = __builtin_fma(Rk[j * m + l], u[l], lu);
            double s = 0.0; bool first = true;
#pragma unroll
            for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + j), Bm[r * m + j], Vx[r]);
            Qu[j] = lu + s;
        }
        // B.T @ V_xx
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + j), Bm[r * m + j], Vxx[r * n + c]);
                BtV[j * n + c] = s;
            }
        // :101  Q_ux = (B.T V_xx) A ; :102  Q_uu = R + (B.T V_xx) B
#pragma unroll
        for (int j = 0; j < m; ++j) {
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + c), A[r * n + c], BtV[j * n + r]);
                Qux[j * n + c] = s;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + l), Bm[r * m + l], BtV[j * n + r]);
                Quu[j * m + l] = Rk[j * m + l] + s;
            }
            Quu[j * m + j] += quu_reg;
        }
        // :103  Q_xx = Q_k + (A.T V_xx) A, one row of A.T V_xx at a time
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double row[n];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + i), A[r * n + i], Vxx[r * n + c]);
                row[c] = s;
            }
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + c), A[r * n + c], row[r]);
                Qxx[i * n + c] = Qk[i * n + c] + s;
            }
        }
        // :106-108  gains
        const bool ok = inv_spd(Quu, Quu_inv);
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) s = __builtin_fma(-Quu_inv[j * m + l], Qu[l], s);
            k[j] = s;
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double t = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) t = __builtin_fma(-Quu_inv[j * m + l], Qux[l * n + c], t);
                K[j * n + c] = t;
            }
        }
        // :115-116  expected reduction terms
        double g = 0.0, h = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) g = __builtin_fma(-Qu[j], k[j], g);
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < m; ++i) s = __builtin_fma(0.5 * k[i], Quu[i * m + j], s);
            h = __builtin_fma(s, k[j], h);
        }
        g_out = g; h_out = h;
        // :124-125  value function update (four-term forms, no symmetrisation)
        double KtQuu[n * m];
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < m; ++j) s = __builtin_fma(K[j * n + i], Quu[j * m + l], s);
                KtQuu[i * m + l] = s;
            }
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) v1 = __builtin_fma(KtQuu[i * m + l], k[l], v1);
#pragma unroll
            for (int j = 0; j < m; ++j) v2 = __builtin_fma(K[j * n + i], Qu[j], v2);
#pragma unroll
            for (int j = 0; j < m; ++j) v3 = __builtin_fma(Qux[j * n + i], k[j], v3);
            Vx[i] = ((Qx[i] + v1) + v2) + v3;
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                for (int j = 0; j < m; ++j) t1 = __builtin_fma(Qux[j * n + i], K[j * n + c], t1);
#pragma unroll
                for (int j = 0; j < m; ++j) t2 = __builtin_fma(K[j * n + i], Qux[j * n + c], t2);
#pragma unroll
                for (int l = 0; l < m; ++l) t3 = __builtin_fma(KtQuu[i * m + l], K[l * n + c], t3);
                Vxx[i * n + c] = ((Qxx[i * n + c] + t1) + t2) + t3;
            }
        }
        return ok;
    }
};

// stage cost of compute_cost (ilqr.py:56): (u^T R) u
template <int m>
__device__ __forceinline__ double input_cost(const double (&u)[m], const double (&Rk)[m * m])
{
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < m; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < m; ++i) s = __builtin_fma(u[i], Rk[i * m + j], s);
        c = __builtin_fma(s, u[j], c);
    }
    return c;
}
// terminal cost (ilqr.py:59-60): d = target - x_N, (d^T Q_T) d
template <int n>
__device__ __forceinline__ double terminal_cost(const double (&x)[n], const double (&target)[n], const double (&QT)[n * n])
{
    double d[n];
#pragma unroll
    for (int i = 0; i < n; ++i) d[i] = target[i] - x[i];
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) s = __builtin_fma(d[i], QT[i * n + j], s);
        c = __builtin_fma(s, d[j], c);
    }
    return c;
}

template <int CNT>
__device__ __forceinline__ void load_col(const double* __restrict__ base, double (&out)[CNT])
{
#pragma unroll
    for (int i = 0; i < CNT; ++i) out[i] = base[(size_t)i * TILE];
}
template <int CNT>
__device__ __forceinline__ void store_col(double* __restrict__ base, const double (&v)[CNT])
{
#pragma unroll
    for (int i = 0; i < CNT; ++i) base[(size_t)i * TILE] = v[i];
}
template <int CNT>
__device__ __forceinline__ void load_uniform(const double* __restrict__ base, double (&out)[CNT])
{
#pragma unroll
    for (int i = 0; i < CNT; ++i) out[i] = base[i];
}

// ------------------------------------------------------------------------------------------------
// k_rollout: ilqr.py:30-47 with compute_cost (:49-62) fused.  One lane per problem, one wave per tile.
// ------------------------------------------------------------------------------------------------
template <class S>
__global__ void __launch_bounds__(64) k_rollout(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    double x[n], p[np], target[n], Rk[m * m], QT[n * n];
    load_col<n>(a.INIT + (size_t)tile * n * TILE + lane, x);
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Gp = a.G + (size_t)tile * a.N * m * TILE + lane;
    double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    {
        double x0[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x0[i] = a.zero_first ? 0.0 : x[i];
        store_col<n>(Xp, x0);
    }
    double cost = 0.0;
    for (int t = 0; t < a.N; ++t) {
        double u[m], xn[n];
        load_col<m>(Gp + (size_t)t * m * TILE, u);
        cost = cost + input_cost<m>(u, Rk);
        S::f(x, u, a.dt, p, xn);
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        store_col<n>(Xp + (size_t)(t + 1) * n * TILE, x);                              // :39
        store_col<m>(Up + (size_t)t * m * TILE, u);                                    // :40
    }
    cost = cost + terminal_cost<n>(x, target, QT);
    a.cost[b] = cost;
}

// k_cost: compute_cost (ilqr.py:49-62) of the stored trajectory.
template <class S>
__global__ void __launch_bounds__(64) k_cost(const KArgs a)
{
    constexpr int n = S::n, m = S::m;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    double x[n], target[n], Rk[m * m], QT[n * n];
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    double cost = 0.0;
    for (int t = 0; t < a.N; ++t) {
        double u[m];
        load_col<m>(Up + (size_t)t * m * TILE, u);
        cost = cost + input_cost<m>(u, Rk);
    }
    load_col<n>(a.X + ((size_t)tile * a.state_rows + a.N) * n * TILE + lane, x);
    cost = cost + terminal_cost<n>(x, target, QT);
    a.cost[tile * TILE + lane] = cost;
}

// terminal condition of backwards_pass (ilqr.py:67-70): V_x = Q_T (x_N - target); Vxx already holds Q_T
template <class S>
__device__ __forceinline__ void terminal_value(const double* xNp, const double (&target)[S::n],
                                               const double (&Vxx)[S::n * S::n], double (&Vx)[S::n])
{
    constexpr int n = S::n;
    double xN[n], ed[n];
    load_col<n>(xNp, xN);
#pragma unroll
    for (int i = 0; i < n; ++i) ed[i] = xN[i] - target[i];                             // :68
#pragma unroll
    for (int i = 0; i < n; ++i) {                                                      // :70
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < n; ++c) s = __builtin_fma(Vxx[i * n + c], ed[c], s);
        Vx[i] = s;
    }
}

// end of backwards_pass for one active problem: expected reductions (:128-130), iteration count, the "optimal"
// stop test of solve (:175), trace row
__device__ __forceinline__ void backward_epilogue(const KArgs& a, long long tile, int lane, long long b, double ecr,
                                                  double ecr_g, double ecr_h, bool ok)
{
    double* dj = a.dJ + (size_t)tile * 3 * TILE + lane;
    dj[0] = ecr; dj[TILE] = ecr_g; dj[2 * TILE] = ecr_h;
    a.iters[b] += 1;
    int still = 1;
    if (!ok) { a.status[b] = ILQR_STATUS_NUMERICAL; still = 0; }
    else if (fabs(ecr) < a.low_red) { a.status[b] = ILQR_STATUS_OPTIMAL; still = 0; }
    if (!still) a.active[b] = 0;
    if (a.trace && a.it < a.n_iter) {
        double* tr = a.trace + ((size_t)tile * a.n_iter + a.it) * 6 * TILE + lane;
        tr[0 * TILE] = a.cost[b]; tr[1 * TILE] = ecr; tr[2 * TILE] = ecr_g; tr[3 * TILE] = ecr_h;
        tr[4 * TILE] = -1.0; tr[5 * TILE] = __builtin_nan("");
    }
}

// ------------------------------------------------------------------------------------------------
// k_backward: ilqr.py:64-135.  One lane per problem; V_x, V_xx and the step's Q_* stay in VGPRs.
// ------------------------------------------------------------------------------------------------
template <class S>
__global__ void __launch_bounds__(64) k_backward(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    const bool act = a.active[b] != 0;
    if (__ballot(act) == 0ull) return;   // whole tile finished
    double p[np], target[n], Qk[n * n], Rk[m * m], Vx[n], Vxx[n * n];
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<n * n>(a.WTS, Qk);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, Vxx);                                   // :67  V_xx = Q_T
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    double* kp = a.KFF + (size_t)tile * a.N * m * TILE + lane;
    terminal_value<S>(Xp + (size_t)a.N * n * TILE, target, Vxx, Vx);
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    bool ok = true;
    double x[n], u[m];
    load_col<n>(Xp + (size_t)(a.N - 1) * n * TILE, x);
    load_col<m>(Up + (size_t)(a.N - 1) * m * TILE, u);
    for (int t = a.N - 1; t >= 0; --t) {                                               // :82
        double xnext[n], unext[m];                                                     // prefetch step t-1
        const int tp = t > 0 ? t - 1 : 0;
        load_col<n>(Xp + (size_t)tp * n * TILE, xnext);
        load_col<m>(Up + (size_t)tp * m * TILE, unext);
        double A[n * n], Bm[n * m], k[m], K[m * n], g, h;
        S::A(x, u, a.dt, p, A);                                                        // :96
        S::B(x, u, a.dt, p, Bm);                                                       // :97
        ok = Riccati<S>::step(A, Bm, u, Qk, Rk, a.quu_reg, Vx, Vxx, k, K, g, h) && ok;
        if (act) {
            store_col<m>(kp + (size_t)t * m * TILE, k);                                // :111
            store_col<m * n>(Kp + (size_t)t * (m * n) * TILE, K);                      // :112
        }
        const double cur = g + h;                                                      // :117
        ecr_g += g; ecr_h += h; ecr += cur;                                            // :119-121
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xnext[i];
#pragma unroll
        for (int j = 0; j < m; ++j) u[j] = unext[j];
    }
    if (act) backward_epilogue(a, tile, lane, b, ecr, ecr_g, ecr_h, ok);
}

// ------------------------------------------------------------------------------------------------
// closed-loop rollout of one lane over steps [t0, t1) (ilqr.py:144-157), accumulating the input cost of
// compute_cost; optionally storing the new trajectory (STORE) and the state at every segment boundary (CKPT).
// ------------------------------------------------------------------------------------------------
#ifndef CLOSED_LOOP_PREFETCH
#define CLOSED_LOOP_PREFETCH 1
#endif
template <class S, bool STORE, bool CKPT>
__device__ __forceinline__ double closed_loop_range(const KArgs& a, const double* Xp, const double* Up, const double* Kp,
                                                    const double* kp, double* XOp, double* UOp, double* ckp,
                                                    double (&x)[S::n], const double (&p)[S::np],
                                                    const double (&Rk)[S::m * S::m], const double alpha, const int t0,
                                                    const int t1)
{
    constexpr int n = S::n, m = S::m;
    // Rows of the reference trajectory and gains for the next D steps are in flight while step t computes -- in a ring
    // of registers when the per-step record is small (n <= 4: 10 doubles per slot), else (quadcopter: 68 doubles) only
    // the trajectory row is one step ahead and the gains are left to the scheduler.  A slot is refilled as soon as
    // its step's input is formed, i.e. row t+D of X is read before the STORE of row t+1 (in-place accept, D >= 1).
    constexpr bool PREFETCH = (n + 2 * m + m * n) <= 16;
    constexpr int D = PREFETCH ? CLOSED_LOOP_PREFETCH : 1;
    double cost = 0.0;
    double xb[D][n], ub[D][m], K[D][m * n], kf[D][m];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int tl = t0 + d < t1 ? t0 + d : t1 - 1;
        load_col<n>(Xp + (size_t)tl * n * TILE, xb[d]); load_col<m>(Up + (size_t)tl * m * TILE, ub[d]);
        if constexpr (PREFETCH) { load_col<m * n>(Kp + (size_t)tl * (m * n) * TILE, K[d]); load_col<m>(kp + (size_t)tl * m * TILE, kf[d]); }
    }
    int next_ck = CKPT ? a.ck : 0;
    for (int tb = t0; tb < t1; tb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = tb + d;
            if (t >= t1) break;
            if constexpr (!PREFETCH) {
                load_col<m * n>(Kp + (size_t)t * (m * n) * TILE, K[d]);
                load_col<m>(kp + (size_t)t * m * TILE, kf[d]);
            }
            if constexpr (CKPT) {
                if (t == next_ck) {                                                    // state entering segment t / ck
                    double* c = ckp + (size_t)(t / a.ck) * n * TILE;
#pragma unroll
                    for (int i = 0; i < n; ++i) c[(size_t)i * TILE] = x[i];
                    next_ck += a.ck;
                }
            }
            double u[m], xn[n];
#pragma unroll
            for (int j = 0; j < m; ++j) {
                const double ff = alpha * kf[d][j];                                    // :146
                double fb = 0.0;                                                       // :147
#pragma unroll
                for (int i = 0; i < n; ++i) fb = __builtin_fma(K[d][j * n + i], x[i] - xb[d][i], fb);
                u[j] = (ub[d][j] + fb) + ff;                                           // :148
            }
            {
                const int tl = t + D < t1 ? t + D : t1 - 1;                            // slot d next holds step t + D
                load_col<n>(Xp + (size_t)tl * n * TILE, xb[d]); load_col<m>(Up + (size_t)tl * m * TILE, ub[d]);
                if constexpr (PREFETCH) { load_col<m * n>(Kp + (size_t)tl * (m * n) * TILE, K[d]); load_col<m>(kp + (size_t)tl * m * TILE, kf[d]); }
            }
            cost = cost + input_cost<m>(u, Rk);
            S::f(x, u, a.dt, p, xn);                                                   // :151
#pragma unroll
            for (int i = 0; i < n; ++i) x[i] = xn[i];
            if constexpr (STORE) {
                store_col<m>(UOp + (size_t)t * m * TILE, u);                           // :154
                store_col<n>(XOp + (size_t)(t + 1) * n * TILE, x);                     // :153
            }
        }
    }
    return cost;
}

template <class S, bool STORE, bool CKPT = false>
__device__ __forceinline__ double closed_loop(const KArgs& a, const double* Xp, const double* Up, const double* Kp,
                                              const double* kp, double* XOp, double* UOp, double (&x)[S::n],
                                              const double (&p)[S::np], const double (&target)[S::n],
                                              const double (&Rk)[S::m * S::m], const double (&QT)[S::n * S::n],
                                              const double alpha, double* ckp = nullptr)
{
    double cost = closed_loop_range<S, STORE, CKPT>(a, Xp, Up, Kp, kp, XOp, UOp, ckp, x, p, Rk, alpha, 0, a.N);
    cost = cost + terminal_cost<S::n>(x, target, QT);
    return cost;
}

// ------------------------------------------------------------------------------------------------
// Line-search rollout for systems whose per-step record (x_ref, u_ref, K, k: n + 2m + mn doubles, 68 for the
// quadcopter) is too large to prefetch in every lane's registers.  The W lanes of a problem need the SAME record, so the
// wave loads each record once -- lane l fetches component l/G (+W, +2W, ...) of problem l%G, 64-byte runs of a tile row
// -- and passes it through LDS (double buffered, one step ahead); every lane then reads its problem's record as
// broadcasts.  10 vector loads per step and lane instead of 68 (W = 8), all in flight during the previous step.
// Wave-synchronous: only this wave touches its LDS region and its LDS operations execute in order.
// ------------------------------------------------------------------------------------------------
template <class S, int W>
struct Staged {
    static constexpr int n = S::n, m = S::m, G = TILE / W;
    static constexpr int REC = n + 2 * m + m * n, RECP = REC | 1;        // odd stride: the G records start in distinct banks
    static constexpr int oX = 0, oU = n, oK = n + m, ok = n + m + m * n;
    static constexpr int qX = (n + W - 1) / W, qU = (m + W - 1) / W, qK = (m * n + W - 1) / W;
    static constexpr int WAVE_DOUBLES = 2 * G * RECP;
    static constexpr bool use = (n + 2 * m + m * n) > 16 && W >= 8;      // smaller records are prefetched in registers
};

template <class S, int W, bool CKPT>
__device__ __forceinline__ double closed_loop_staged(const KArgs& a, double* sw, const int lane, const int grp,
                                                     const long long b_duty, double* ckp, const bool valid,
                                                     double (&x)[S::n], const double (&p)[S::np],
                                                     const double (&Rk)[S::m * S::m], const double alpha)
{
    using T = Staged<S, W>;
    constexpr int n = S::n, m = S::m, G = T::G, RECP = T::RECP;
    const int gd = lane % G, cd = lane / G;                                            // load duty: problem gd, component cd (+ W q)
    const long long tile_d = b_duty >> 6;
    const int pl_d = (int)(b_duty & 63);
    const double* Xd = a.X + (size_t)tile_d * a.state_rows * n * TILE + pl_d;
    const double* Ud = a.U + (size_t)tile_d * a.N * m * TILE + pl_d;
    const double* Kd = a.KFB + (size_t)tile_d * a.N * (m * n) * TILE + pl_d;
    const double* kd = a.KFF + (size_t)tile_d * a.N * m * TILE + pl_d;
    double rX[T::qX], rU[T::qU], rK[T::qK], rk[T::qU];
    auto fetch = [&](const int t) {
#pragma unroll
        for (int q = 0; q < T::qX; ++q) { const int c = cd + W * q; rX[q] = Xd[((size_t)t * n + (c < n ? c : n - 1)) * TILE]; }
#pragma unroll
        for (int q = 0; q < T::qU; ++q) { const int c = cd + W * q; rU[q] = Ud[((size_t)t * m + (c < m ? c : m - 1)) * TILE]; }
#pragma unroll
        for (int q = 0; q < T::qK; ++q) { const int c = cd + W * q; rK[q] = Kd[((size_t)t * (m * n) + (c < m * n ? c : m * n - 1)) * TILE]; }
#pragma unroll
        for (int q = 0; q < T::qU; ++q) { const int c = cd + W * q; rk[q] = kd[((size_t)t * m + (c < m ? c : m - 1)) * TILE]; }
    };
    auto stash = [&](const int t) {
        double* d = sw + (t & 1) * (G * RECP) + gd * RECP;
#pragma unroll
        for (int q = 0; q < T::qX; ++q) { const int c = cd + W * q; if (c < n) d[T::oX + c] = rX[q]; }
#pragma unroll
        for (int q = 0; q < T::qU; ++q) { const int c = cd + W * q; if (c < m) d[T::oU + c] = rU[q]; }
#pragma unroll
        for (int q = 0; q < T::qK; ++q) { const int c = cd + W * q; if (c < m * n) d[T::oK + c] = rK[q]; }
#pragma unroll
        for (int q = 0; q < T::qU; ++q) { const int c = cd + W * q; if (c < m) d[T::ok + c] = rk[q]; }
    };
    fetch(0);
    stash(0);
    __builtin_amdgcn_wave_barrier();
    double cost = 0.0;
    int next_ck = CKPT ? a.ck : 0;
    for (int t = 0; t < a.N; ++t) {
        fetch(t + 1 < a.N ? t + 1 : t);                                                // next record, in flight during this step
        const double* rec = sw + (t & 1) * (G * RECP) + grp * RECP;
        if constexpr (CKPT) {
            if (t == next_ck) {                                                        // state entering segment t / ck
                if (valid) {
                    double* c = ckp + (size_t)(t / a.ck) * n * TILE;
#pragma unroll
                    for (int i = 0; i < n; ++i) c[(size_t)i * TILE] = x[i];
                }
                next_ck += a.ck;
            }
        }
        double u[m], xn[n], dx[n];
#pragma unroll
        for (int i = 0; i < n; ++i) dx[i] = x[i] - rec[T::oX + i];
#pragma unroll
        for (int j = 0; j < m; ++j) {
            const double ff = alpha * rec[T::ok + j];                                  // :146
            double fb = 0.0;                                                           // :147
#pragma unroll
            for (int i = 0; i < n; ++i) fb = __builtin_fma(rec[T::oK + j * n + i], dx[i], fb);
            u[j] = (rec[T::oU + j] + fb) + ff;                                         // :148
        }
        cost = cost + input_cost<m>(u, Rk);
        S::f(x, u, a.dt, p, xn);                                                       // :151
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        __builtin_amdgcn_wave_barrier();                                               // reads of this parity's other buffer are done
        stash(t + 1);
        __builtin_amdgcn_wave_barrier();
    }
    return cost;
}

// ------------------------------------------------------------------------------------------------
// k_linesearch: ilqr.py:180-202.  W adjacent lanes evaluate W consecutive trials alpha_j of one problem
// (64/W problems per wave); the accepted trial is the FIRST j whose Armijo ratio passes (:193), found with a
// ballot over the lane group -- not the lowest cost.  Rounds of W trials repeat until accepted or exhausted.
// ------------------------------------------------------------------------------------------------
// register budget of the line search = 512 / (waves per SIMD): what each system's time step needs without spilling
#ifndef LS_WAVES_SMALL
#define LS_WAVES_SMALL 2
#endif
template <class S>
constexpr int ls_waves_per_simd()
{
    return S::n > 4 ? 1 : (S::id == ILQR_DYN_ACROBOT_ID ? 2 : LS_WAVES_SMALL);   // (4 for cartpole spills 14 VGPRs and gains nothing)
}
template <class S, int W, bool CKPT>
__global__ void __launch_bounds__(W == 64 ? 512 : 256, ls_waves_per_simd<S>()) k_linesearch(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    constexpr int G = TILE / W;
    // logical block: neighbours (2k, 2k+1) sit 8 physical block ids apart (same XCD under round-robin dispatch)
    const unsigned pb = blockIdx.x, nb16 = gridDim.x & ~15u;
    const unsigned lb = pb < nb16 ? (pb & ~15u) + ((pb & 7u) << 1) + ((pb >> 3) & 1u) : pb;
    const long long wave = ((long long)lb * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane / W, al = lane % W;
    const int r = a.ls_round;
    long long b;
    bool act;
    if (r == 0) {                                            // every active problem of the (sub-)batch, in place
        if (wave * G >= a.n_tiles * TILE) return;
        b = a.tile0 * TILE + wave * G + grp;
        act = a.active[b] != 0;
    } else {                                                 // the problems round r-1 left searching, packed densely
        const long long cnt = a.ls_count[r - 1], g = wave * G + grp;
        if (wave * G >= cnt) return;
        act = g < cnt;
        b = act ? a.ls_list[(size_t)((r - 1) & 1) * a.Bpad + a.tile0 * TILE + g] : a.tile0 * TILE;
    }
    const long long tile = b >> 6;
    const int pl = (int)(b & 63);
    if (__ballot(act) == 0ull) return;
    double x0[n], p[np], target[n], Rk[m * m], QT[n * n];
    load_col<n>(a.INIT + (size_t)tile * n * TILE + pl, x0);
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + pl, p);
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + pl, target);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double cur_cost = a.cost[b];
    const double dg = a.dJ[((size_t)tile * 3 + 1) * TILE + pl], dh = a.dJ[((size_t)tile * 3 + 2) * TILE + pl];
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + pl;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + pl;
    const double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + pl;
    const double* kp = a.KFF + (size_t)tile * a.N * m * TILE + pl;
    const int j = r * W + al;
    const bool valid = act && j < a.n_trials;
    double c = __builtin_nan("");
    bool okj = false;
    if constexpr (Staged<S, W>::use) {
        // whole wave together (the record loads are shared); lanes without a trial compute on and are ignored
        __shared__ double sh[(W == 64 ? 8 : 4) * Staged<S, W>::WAVE_DOUBLES];
        double* sw = sh + (threadIdx.x >> 6) * Staged<S, W>::WAVE_DOUBLES;
        const long long b_duty = __shfl(b, (lane % G) * W);
        const double alpha = a.alphas[j < a.n_trials ? j : 0];
        double x[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        double* ckp = CKPT ? a.CKPT + (size_t)b * a.n_seg * n * TILE + j : nullptr;
        c = closed_loop_staged<S, W, CKPT>(a, sw, lane, grp, b_duty, ckp, valid, x, p, Rk, alpha);
        c = c + terminal_cost<n>(x, target, QT);
        const double cost_difference = cur_cost - c;                               // :191
        const double expected = alpha * dg + (alpha * alpha) * dh;                 // :192
        okj = valid && (cost_difference / expected) > a.armijo_thr;                // :193 (NaN -> false)
        if (!valid) c = __builtin_nan("");
    } else if (valid) {
        const double alpha = a.alphas[j];
        double x[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        double* ckp = CKPT ? a.CKPT + (size_t)b * a.n_seg * n * TILE + j : nullptr;
        c = closed_loop<S, false, CKPT>(a, Xp, Up, Kp, kp, nullptr, nullptr, x, p, target, Rk, QT, alpha, ckp);
        const double cost_difference = cur_cost - c;                               // :191
        const double expected = alpha * dg + (alpha * alpha) * dh;                 // :192
        okj = (cost_difference / expected) > a.armijo_thr;                         // :193 (NaN -> false)
    }
    const unsigned long long mask = __ballot(okj);
    const unsigned long long gm = (W == 64) ? mask : ((mask >> (grp * W)) & ((1ull << (W % 64)) - 1ull));
    const int first = gm != 0ull ? __ffsll((long long)gm) - 1 : 0;
    const double cwin = __shfl(c, grp * W + first);
    if (al == 0 && act) {
        a.ls_rounds[b] += 1;
        if (gm != 0ull) {
            a.acc_j[b] = r * W + first;
            a.new_cost[b] = cwin;
        } else if ((r + 1) * W < a.n_trials) {                                     // :200-202, next round's list
            const int slot = atomicAdd(a.ls_count + r, 1);
            a.ls_list[(size_t)(r & 1) * a.Bpad + a.tile0 * TILE + slot] = (int)b;
        } else {                                                                   // :204 learning rate ran out
            a.acc_j[b] = -1;
            a.new_cost[b] = __builtin_nan("");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_forward: ilqr.forwards_pass (ilqr.py:137-158), one lane per problem.
//   ACCEPT = true : the accept step of solve (:194-207): problems with acc_j >= 0 re-roll the winning alpha
//                   in place into X/U, take new_cost as their cost; the others stop with "low learning rate".
//   ACCEPT = false: plain forwards_pass with a per-problem alpha into XO/UO (+ its cost into new_cost).
// ------------------------------------------------------------------------------------------------
template <class S, bool ACCEPT>
__global__ void __launch_bounds__(64) k_forward(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    bool run = true;
    double alpha;
    int accj = -1;
    if constexpr (ACCEPT) {
        const bool act = a.active[b] != 0;
        accj = act ? a.acc_j[b] : -1;
        run = act && accj >= 0;
        alpha = a.alphas[accj >= 0 ? accj : 0];
        if (act && accj < 0) { a.status[b] = ILQR_STATUS_LOW_LR; a.active[b] = 0; }    // :204-207
        if (__ballot(run) == 0ull) return;
    } else {
        alpha = a.alpha_pp[b];
    }
    double x[n], p[np], target[n], Rk[m * m], QT[n * n];
    load_col<n>(a.INIT + (size_t)tile * n * TILE + lane, x);
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    const double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    const double* kp = a.KFF + (size_t)tile * a.N * m * TILE + lane;
    double* XOp = a.XO + (size_t)tile * a.state_rows * n * TILE + lane;
    double* UOp = a.UO + (size_t)tile * a.N * m * TILE + lane;
    if (run) {
        if constexpr (!ACCEPT) {
            double x0[n];
#pragma unroll
            for (int i = 0; i < n; ++i) x0[i] = a.zero_first ? 0.0 : x[i];             // :138,:143 (row 0 stays zero)
            store_col<n>(XOp, x0);
        }
        const double c = closed_loop<S, true>(a, Xp, Up, Kp, kp, XOp, UOp, x, p, target, Rk, QT, alpha);
        if constexpr (ACCEPT) {
            const double nc = a.new_cost[b];          // cost of the winning trial as the line search saw it
            a.cost[b] = nc;                                                            // :197
            a.accepts[b] += 1;
            if (a.trace && a.it < a.n_iter) {
                double* tr = a.trace + ((size_t)tile * a.n_iter + a.it) * 6 * TILE + lane;
                tr[4 * TILE] = (double)accj; tr[5 * TILE] = nc;
            }
            (void)c;
        } else {
            a.new_cost[b] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split pipeline (small batches: the horizon-sequential kernels would leave most SIMDs idle).
//   k_accept_par : the accept step (:194-207) in parallel over (tile, segment): each lane restarts the winning
//                  trial from the checkpoint the line search stored at its segment boundary and re-rolls ck steps
//                  into X2/U2 (same arithmetic, same order => same bits as the trial itself).
//   k_linearize  : parallel over (tile, block of time steps): evaluates A_t, B_t (ilqr.py:96-97) at the stored
//                  trajectory and keeps their general entries in LIN; with COPY it first moves the accepted
//                  trajectory X2/U2 -> X/U.
//   k_riccati    : the horizon-sequential part of backwards_pass (:99-125) reading LIN instead of calling A(), B().
// ------------------------------------------------------------------------------------------------
template <class S>
__global__ void __launch_bounds__(64) k_accept_par(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + blockIdx.x / a.n_seg;
    const int seg = (int)(blockIdx.x % a.n_seg);
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    const bool act = a.active[b] != 0;
    const int accj = act ? a.acc_j[b] : -1;
    const bool run = act && accj >= 0;
    if (__ballot(act) == 0ull) return;
    if (run) {
        const double alpha = a.alphas[accj];
        double x[n], p[np], Rk[m * m];
        load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
        load_uniform<m * m>(a.WTS + n * n, Rk);
        if (seg == 0) {
            load_col<n>(a.INIT + (size_t)tile * n * TILE + lane, x);
        } else {
            const double* c = a.CKPT + ((size_t)b * a.n_seg + seg) * n * TILE + accj;
#pragma unroll
            for (int i = 0; i < n; ++i) x[i] = c[(size_t)i * TILE];
        }
        const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
        const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
        const double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
        const double* kp = a.KFF + (size_t)tile * a.N * m * TILE + lane;
        double* XOp = a.X2 + (size_t)tile * a.state_rows * n * TILE + lane;
        double* UOp = a.U2 + (size_t)tile * a.N * m * TILE + lane;
        const int t0 = seg * a.ck;
        const int t1 = t0 + a.ck < a.N ? t0 + a.ck : a.N;
        (void)closed_loop_range<S, true, false>(a, Xp, Up, Kp, kp, XOp, UOp, nullptr, x, p, Rk, alpha, t0, t1);
    }
    if (seg == 0 && act) {                                                             // per-problem bookkeeping, once
        if (accj < 0) { a.status[b] = ILQR_STATUS_LOW_LR; a.active[b] = 0; }           // :204-207
        else {
            const double nc = a.new_cost[b];
            a.cost[b] = nc;                                                            // :197
            a.accepts[b] += 1;
            if (a.trace && a.it < a.n_iter) {
                double* tr = a.trace + ((size_t)tile * a.n_iter + a.it) * 6 * TILE + lane;
                tr[4 * TILE] = (double)accj; tr[5 * TILE] = nc;
            }
        }
    }
}

template <class S>
struct Lin {
    static constexpr int count_x(const char* pat, int len)
    {
        int c = 0;
        for (int i = 0; i < len; ++i) c += pat[i] == 'x';
        return c;
    }
    static constexpr int nA = count_x(S::A_pat, S::n * S::n);
    static constexpr int nB = count_x(S::B_pat, S::n * S::m);
    static constexpr int n_lin = nA + nB;
};

constexpr int LIN_TS = 8;   // time steps per wave in k_linearize

template <class S, bool COPY>
__global__ void __launch_bounds__(64) k_linearize(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const int chunks = (a.N + LIN_TS - 1) / LIN_TS;
    const long long tile = a.tile0 + blockIdx.x / chunks;
    const int chunk = (int)(blockIdx.x % chunks);
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    const bool act = a.active[b] != 0;
    const bool run = COPY ? (act && a.acc_j[b] >= 0) : act;
    if (__ballot(run) == 0ull) return;
    if (!run) return;
    double p[np];
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    const double* X2p = a.X2 + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* U2p = a.U2 + (size_t)tile * a.N * m * TILE + lane;
    double* Lp = a.LIN + (size_t)tile * a.N * Lin<S>::n_lin * TILE + lane;
    const int t0 = chunk * LIN_TS;
    const int t1 = t0 + LIN_TS < a.N ? t0 + LIN_TS : a.N;
    for (int t = t0; t < t1; ++t) {
        double x[n], u[m];
        if constexpr (COPY) {
            if (t == 0) load_col<n>(Xp, x);                 // row 0 never changes (zero or init_state)
            else { load_col<n>(X2p + (size_t)t * n * TILE, x); store_col<n>(Xp + (size_t)t * n * TILE, x); }
            load_col<m>(U2p + (size_t)t * m * TILE, u);
            store_col<m>(Up + (size_t)t * m * TILE, u);
            if (t == a.N - 1) {
                double xN[n];
                load_col<n>(X2p + (size_t)a.N * n * TILE, xN);
                store_col<n>(Xp + (size_t)a.N * n * TILE, xN);
            }
        } else {
            load_col<n>(Xp + (size_t)t * n * TILE, x);
            load_col<m>(Up + (size_t)t * m * TILE, u);
        }
        double A[n * n], Bm[n * m];
        S::A(x, u, a.dt, p, A);                                                        // :96
        S::B(x, u, a.dt, p, Bm);                                                       // :97
        double* l = Lp + (size_t)t * Lin<S>::n_lin * TILE;
        int e = 0;
#pragma unroll
        for (int i = 0; i < n * n; ++i) if (S::a_kind(i) == 'x') { l[(size_t)e * TILE] = A[i]; ++e; }
#pragma unroll
        for (int i = 0; i < n * m; ++i) if (S::b_kind(i) == 'x') { l[(size_t)e * TILE] = Bm[i]; ++e; }
    }
}

template <class S>
__device__ __forceinline__ void load_lin(const double* l, double (&A)[S::n * S::n], double (&Bm)[S::n * S::m])
{
    constexpr int n = S::n, m = S::m;
    int e = 0;
#pragma unroll
    for (int i = 0; i < n * n; ++i) {
        const char kd = S::a_kind(i);
        if (kd == 'x') { A[i] = l[(size_t)e * TILE]; ++e; } else A[i] = kd == '1' ? 1.0 : 0.0;
    }
#pragma unroll
    for (int i = 0; i < n * m; ++i) {
        const char kd = S::b_kind(i);
        if (kd == 'x') { Bm[i] = l[(size_t)e * TILE]; ++e; } else Bm[i] = kd == '1' ? 1.0 : 0.0;
    }
}

#ifndef RICCATI_GRP_WAVES
#define RICCATI_GRP_WAVES 2
#endif
#ifndef RICCATI_PREFETCH
#define RICCATI_PREFETCH 4
#endif
template <class S>
__global__ void __launch_bounds__(64) k_riccati(const KArgs a)
{
    constexpr int n = S::n, m = S::m;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    const bool act = a.active[b] != 0;
    if (__ballot(act) == 0ull) return;
    double target[n], Qk[n * n], Rk[m * m], Vx[n], Vxx[n * n];
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<n * n>(a.WTS, Qk);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, Vxx);                                   // :67
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    const double* Lp = a.LIN + (size_t)tile * a.N * Lin<S>::n_lin * TILE + lane;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    double* kp = a.KFF + (size_t)tile * a.N * m * TILE + lane;
    terminal_value<S>(Xp + (size_t)a.N * n * TILE, target, Vxx, Vx);
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    bool ok = true;
    // A_t, B_t, u_t of the next D steps are in flight while step t computes: one step (~0.5 us of dependent arithmetic)
    // is shorter than an HBM round trip, and one wave per SIMD has nothing else to hide it behind.
    constexpr int D = Lin<S>::n_lin + m <= 16 ? RICCATI_PREFETCH : 1;
    double Ar[D][n * n], Br[D][n * m], ur[D][m];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int tl = a.N - 1 - d > 0 ? a.N - 1 - d : 0;
        load_lin<S>(Lp + (size_t)tl * Lin<S>::n_lin * TILE, Ar[d], Br[d]);
        load_col<m>(Up + (size_t)tl * m * TILE, ur[d]);
    }
    for (int t0 = a.N - 1; t0 >= 0; t0 -= D) {                                         // :82
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 - d;
            if (t < 0) break;
            double k[m], K[m * n], g, h;
            ok = Riccati<S>::step(Ar[d], Br[d], ur[d], Qk, Rk, a.quu_reg, Vx, Vxx, k, K, g, h) && ok;
            const int tl = t - D > 0 ? t - D : 0;                                      // slot d next holds step t - D
            load_lin<S>(Lp + (size_t)tl * Lin<S>::n_lin * TILE, Ar[d], Br[d]);
            load_col<m>(Up + (size_t)tl * m * TILE, ur[d]);
            if (act) {
                store_col<m>(kp + (size_t)t * m * TILE, k);                            // :111
                store_col<m * n>(Kp + (size_t)t * (m * n) * TILE, K);                  // :112
            }
            const double cur = g + h;                                                  // :117
            ecr_g += g; ecr_h += h; ecr += cur;                                        // :119-121
        }
    }
    if (act) backward_epilogue(a, tile, lane, b, ecr, ecr_g, ecr_h, ok);
}

// ------------------------------------------------------------------------------------------------
// k_riccati_grp: the same recursion as k_riccati with a ROW of 16 lanes per problem (4 problems per wave), for
// systems whose V_xx does not fit one lane's registers (n = 12: V_xx, Q_xx and A^T V_xx are 3 x 144 doubles).
//   * lane r of a row keeps ROW r of A_t and of B_t in registers (gathered straight from LIN, next step in flight);
//     between steps lane j holds COLUMN j of V_xx and V_x[j].
//   * every product needs one operand that is the same for the whole row of lanes (A[r][i], K[l][j], V_x[r], ...).
//     Those are never staged in LDS: the DPALU broadcast `v_fmac_f64_dpp ... row_newbcast:r` multiplies lane r's
//     register into every lane of its row inside the FMA itself.  (The first version read them from LDS: 262 LDS
//     instructions per step, and the CU's single LDS pipe -- shared by the four SIMDs -- was the bound, VALU 33 % busy.)
//   * column phase (lane j): (A^T V_xx)[:, j], (B^T V_xx)[:, j] over the structural non-zeros; three small LDS
//     transposes remain (A^T V_xx columns -> rows, A rows -> columns for Q_ux / Q_x, new V_xx rows -> columns);
//   * row phase (lane i): row i of Q_xx = (A^T V_xx) A and of the four-term V_xx update (ilqr.py:125), column i of
//     Q_ux and K, Q_x[i]; Q_uu, Q_u, the inverse and k are small and computed by every lane.
// One wave per workgroup: the LDS hand-offs are ordered by program order (the barriers only pin the compiler).
// Operation order follows the reference ((A^T V) A, (B^T V) A, ...), as Riccati<S>::step does.
// ------------------------------------------------------------------------------------------------
template <class S>
struct Grp {
    static constexpr int n = S::n, m = S::m;
    static constexpr int GL = 16;                       // one DPP row
    static constexpr int P = TILE / GL;                 // problems per wave
    static constexpr int LD = n + 1;                    // padded leading dimension of the transpose buffers
    static constexpr int oT = 0, oAT = oT + n * LD, SLOT = ((oAT + n * LD + 1) | 1);
    static_assert(n <= GL, "k_riccati_grp: one lane per state");
};

// index of dense entry e of A (or B) in the step's LIN row; -1 structural zero, -2 structural one
template <class S, bool IS_B>
__device__ __forceinline__ int lin_index(const int e)
{
    constexpr int cnt_all = IS_B ? S::n * S::m : S::n * S::n;
    int cnt = IS_B ? Lin<S>::nA : 0;
    int res = -1;
#pragma unroll
    for (int q = 0; q < cnt_all; ++q) {
        const char kd = IS_B ? S::b_kind(q) : S::a_kind(q);
        if (q == e) res = kd == 'x' ? cnt : (kd == '1' ? -2 : -1);
        if (kd == 'x') ++cnt;
    }
    return res;
}

// acc += (lane L of this lane's 16-lane row of `src`) * own.  The wait states a DPP read of a just-written VGPR needs
// are not inserted around inline asm by the compiler, hence the s_nop.
template <int L>
__device__ __forceinline__ void fmac_row(double& acc, const double src, const double own)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(L));
}
template <int L>
__device__ __forceinline__ double bcast_row(const double src)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(L));
    return r;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <class S>
__global__ void __launch_bounds__(64, RICCATI_GRP_WAVES) k_riccati_grp(const KArgs a)
{
    using G = Grp<S>;
    constexpr int n = S::n, m = S::m, GL = G::GL, P = G::P, NL = Lin<S>::n_lin, LD = G::LD;
    __shared__ double sh[P * G::SLOT + n * n];
    const int lane = threadIdx.x;
    const int grp = lane / GL, c = lane % GL;
    // The 16 workgroups of a tile touch the same 128-byte lines (4 problems = 32 bytes of each): workgroups are dealt
    // round-robin over the 8 XCDs, so logical ids are chosen such that a tile's workgroups share one XCD's L2.
    const unsigned pb = blockIdx.x, nb8 = gridDim.x & ~7u;
    const unsigned lb = pb < nb8 ? (pb & 7u) * (nb8 >> 3) + (pb >> 3) : pb;
    const long long b = a.tile0 * TILE + (long long)lb * P + grp;
    const long long tile = b >> 6;
    const int pl = (int)(b & 63);
    const bool act = a.active[b] != 0;
    if (__ballot(act) == 0ull) return;
    double* s = sh + grp * G::SLOT;
    const bool row = c < n;
    const int cc = row ? c : n - 1;                     // surplus lanes shadow the last row / column and never store
    // where row cc of A_t / B_t sits in a LIN record: byte offsets of the general entries (structural zeros and ones
    // load a harmless valid entry instead; their constants are written to the transpose buffer once, below)
    int aoff[n], boff[m];
    unsigned amask = 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int q = lin_index<S, false>(cc * n + i);
        aoff[i] = (q > 0 ? q : 0) * (TILE * (int)sizeof(double));
        amask |= (q >= 0 ? 1u : 0u) << i;
        if (row && q < 0) s[G::oAT + i * LD + c] = q == -2 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int l = 0; l < m; ++l) {
        const int q = lin_index<S, true>(cc * m + l);
        boff[l] = (q > 0 ? q : 0) * (TILE * (int)sizeof(double));
    }
    double Rk[m * m], Vc[n], vx;
    double* sQk = sh + P * G::SLOT;                                                    // Q_k, shared by the rows
    for (int e = lane; e < n * n; e += TILE) sQk[e] = a.WTS[e];
    load_uniform<m * m>(a.WTS + n * n, Rk);
    const double* QT = a.WTS + n * n + m * m;
#pragma unroll
    for (int r = 0; r < n; ++r) Vc[r] = QT[r * n + cc];                                // :67 column j of Q_T
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + pl;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + pl;
    const double* Lp = a.LIN + (size_t)tile * a.N * NL * TILE + pl;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + pl;
    double* kp = a.KFF + (size_t)tile * a.N * m * TILE + pl;
    {
        double target[n], xN[n];
        load_col<n>(a.TARGET + (size_t)tile * n * TILE + pl, target);
        load_col<n>(Xp + (size_t)a.N * n * TILE, xN);
        double acc = 0.0;                                                              // :68-70 V_x[j] = (Q_T ed)[j]
#pragma unroll
        for (int q = 0; q < n; ++q) acc = __builtin_fma(QT[cc * n + q], xN[q] - target[q], acc);
        vx = acc;
    }
    auto load_rows = [&](int t, double (&Ar)[n], double (&Br)[m], double (&u)[m]) {
        const char* l = (const char*)(Lp + (size_t)t * NL * TILE);
#pragma unroll
        for (int i = 0; i < n; ++i) {
            int o = aoff[i];
            asm volatile("" : "+v"(o));                  // keep base + offset out of the loop-invariant (spilled) set
            Ar[i] = *(const double*)(l + o);
        }
#pragma unroll
        for (int q = 0; q < m; ++q) {
            int o = boff[q];
            asm volatile("" : "+v"(o));
            Br[q] = *(const double*)(l + o);
        }
#pragma unroll
        for (int q = 0; q < m; ++q) u[q] = Up[((size_t)t * m + q) * TILE];
    };
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    bool ok = true;
    __syncthreads();
    // one time step with rows (Ar, Br, u) of step t in registers; the rows of step t-1 are requested into (An, Bn, un)
    auto step = [&](const int t, const double (&Ar)[n], const double (&Br)[m], const double (&u)[m], double (&An)[n],
                    double (&Bn)[m], double (&un)[m]) {
        load_rows(t > 0 ? t - 1 : 0, An, Bn, un);
        // ---- column phase (lane j): A^T V_xx and B^T V_xx, column j
        double AtV[n], BtV[m];
#pragma unroll
        for (int i = 0; i < n; ++i) AtV[i] = 0.0;
#pragma unroll
        for (int l = 0; l < m; ++l) BtV[l] = 0.0;
        static_for<0, n>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            static_for<0, n>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr char kd = S::A_pat[r * n + i];
                if constexpr (kd == 'x') fmac_row<r>(AtV[i], Ar[i], Vc[r]);            // A[r][i] * V[r][j]
                else if constexpr (kd == '1') AtV[i] = AtV[i] + Vc[r];
            });
            static_for<0, m>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                constexpr char kd = S::B_pat[r * m + l];
                if constexpr (kd == 'x') fmac_row<r>(BtV[l], Br[l], Vc[r]);            // B[r][l] * V[r][j]
                else if constexpr (kd == '1') BtV[l] = BtV[l] + Vc[r];
            });
        });
        if (row) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
                s[G::oT + i * LD + c] = AtV[i];
                if ((amask >> i) & 1u) s[G::oAT + i * LD + c] = Ar[i];
            }
        }
        __syncthreads();
        // ---- gains (every lane: Q_uu, Q_u, the inverse, k; lane i: column i of Q_ux and K, Q_x[i]).  Done before the
        // row phase so that Q_uu and its inverse (2 m^2 doubles) are dead when the rows of A^T V_xx come into registers.
        double Qux[m], Qu[m], k[m], Kc[m], KtQuu[m], g = 0.0, h = 0.0;
        {
            double Quu[m * m], Quu_inv[m * m], Acol[n], Qx = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) Acol[r] = s[G::oAT + cc * LD + r];             // A[r][i]
#pragma unroll
            for (int q = 0; q < m * m; ++q) Quu[q] = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) Qu[l] = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) Qux[l] = 0.0;
            static_for<0, n>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                static_for<0, m>([&](auto l2c) {
                    constexpr int l2 = decltype(l2c)::value;
                    constexpr char kd = S::B_pat[r * m + l2];
                    if constexpr (kd == 'x') {
                        const double Brl = bcast_row<r>(Br[l2]);                       // B[r][l2], the same in every lane
#pragma unroll
                        for (int l = 0; l < m; ++l) fmac_row<r>(Quu[l * m + l2], BtV[l], Brl);       // :102
                        fmac_row<r>(Qu[l2], vx, Brl);                                  // :100 B[r][l2] * V_x[r]
                    } else if constexpr (kd == '1') {
#pragma unroll
                        for (int l = 0; l < m; ++l) Quu[l * m + l2] = Quu[l * m + l2] + bcast_row<r>(BtV[l]);
                        Qu[l2] = Qu[l2] + bcast_row<r>(vx);
                    }
                });
#pragma unroll
                for (int l = 0; l < m; ++l) fmac_row<r>(Qux[l], BtV[l], Acol[r]);       // :101 (B^T V_xx)[l][r] * A[r][i]
                fmac_row<r>(Qx, vx, Acol[r]);                                          // :99  A[r][i] * V_x[r]
            });
#pragma unroll
            for (int l = 0; l < m; ++l) {
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) Quu[l * m + l2] = Rk[l * m + l2] + Quu[l * m + l2];
                Quu[l * m + l] += a.quu_reg;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {                                              // :93 l_u = R_k u
                double lu = 0.0;
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) lu = __builtin_fma(Rk[l * m + l2], u[l2], lu);
                Qu[l] = lu + Qu[l];
            }
                ok = Riccati<S>::inv_spd(Quu, Quu_inv) && ok;                              // :106
    #pragma unroll
            for (int l = 0; l < m; ++l) {                                              // :107-108
                double acc = 0.0, acc2 = 0.0;
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) {
                    acc = __builtin_fma(-Quu_inv[l * m + l2], Qu[l2], acc);
                    acc2 = __builtin_fma(-Quu_inv[l * m + l2], Qux[l2], acc2);
                }
                k[l] = acc;
                Kc[l] = acc2;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) g = __builtin_fma(-Qu[l], k[l], g);            // :115
#pragma unroll
            for (int l = 0; l < m; ++l) {                                              // :116
                double acc = 0.0;
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) acc = __builtin_fma(0.5 * k[l2], Quu[l2 * m + l], acc);
                h = __builtin_fma(acc, k[l], h);
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double acc = 0.0;
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) acc = __builtin_fma(Kc[l2], Quu[l2 * m + l], acc);
                KtQuu[l] = acc;
            }
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;                                       // :124
#pragma unroll
            for (int l = 0; l < m; ++l) v1 = __builtin_fma(KtQuu[l], k[l], v1);
#pragma unroll
            for (int l = 0; l < m; ++l) v2 = __builtin_fma(Kc[l], Qu[l], v2);
#pragma unroll
            for (int l = 0; l < m; ++l) v3 = __builtin_fma(Qux[l], k[l], v3);
            vx = ((Qx + v1) + v2) + v3;
        }
        if (act) {
            if (row) {
#pragma unroll
                for (int l = 0; l < m; ++l) Kp[((size_t)t * (m * n) + l * n + c) * TILE] = Kc[l];   // :112
            }
#pragma unroll
            for (int l = 0; l < m; ++l) if (c == l) kp[((size_t)t * m + l) * TILE] = k[l];          // :111
        }
        const double cur = g + h;                                                      // :117
        ecr_g += g; ecr_h += h; ecr += cur;                                            // :119-121
        // ---- row phase (lane i): row i of Q_xx (:103) and of the new V_xx (:125), one entry at a time, written
        // straight into the lane's own row of the transpose buffer (whose old contents are in Ri by then)
        double Ri[n], Aw[n];
#pragma unroll
        for (int r = 0; r < n; ++r) Ri[r] = s[G::oT + cc * LD + r];                    // (A^T V_xx)[i][:]
#pragma unroll
        for (int j = 0; j < n; ++j) Aw[j] = s[G::oAT + j * LD + cc];                   // own row of A_t again (its
        static_for<0, n>([&](auto jc) {                                                // registers ended with the column phase)
            constexpr int j = decltype(jc)::value;
            double acc = 0.0;
            static_for<0, n>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr char kd = S::A_pat[r * n + j];
                if constexpr (kd == 'x') fmac_row<r>(acc, Aw[j], Ri[r]);               // (A^T V_xx)[i][r] * A[r][j]
                else if constexpr (kd == '1') acc = acc + Ri[r];
            });
            const double Qxx = sQk[cc * n + j] + acc;
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) fmac_row<j>(t1, Kc[l], Qux[l]);                // Q_ux[l][i] * K[l][j]
#pragma unroll
            for (int l = 0; l < m; ++l) fmac_row<j>(t2, Qux[l], Kc[l]);                // K[l][i] * Q_ux[l][j]
#pragma unroll
            for (int l = 0; l < m; ++l) fmac_row<j>(t3, Kc[l], KtQuu[l]);              // (K^T Q_uu)[i][l] * K[l][j]
            const double Vij = ((Qxx + t1) + t2) + t3;
            if (row) s[G::oT + c * LD + j] = Vij;                                      // rows -> columns, part 1
        });
        __syncthreads();
#pragma unroll
        for (int r = 0; r < n; ++r) Vc[r] = s[G::oT + r * LD + cc];
        __syncthreads();
    };
    double A0[n], B0[m], u0[m], A1[n], B1[m], u1[m];
    load_rows(a.N - 1, A0, B0, u0);
    for (int t = a.N - 1; t >= 0; t -= 2) {                                            // :82
        step(t, A0, B0, u0, A1, B1, u1);
        if (t > 0) step(t - 1, A1, B1, u1, A0, B0, u0);
    }
    if (act && c == 0) backward_epilogue(a, tile, pl, b, ecr, ecr_g, ecr_h, ok);
}

// ------------------------------------------------------------------------------------------------
// k_ltvlqr: ltvlqr.py:3-28.  One lane per problem; P stays in VGPRs.
// K = -inv(R + B'PB) B'PA,  P <- Q + A'PA - (A'PB) inv(R + B'PB) (B'PA)
// ------------------------------------------------------------------------------------------------
template <class S>
__global__ void __launch_bounds__(64) k_ltvlqr(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    double p[np], Qk[n * n], Rk[m * m], P[n * n];
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    load_uniform<n * n>(a.WTS, Qk);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, P);                                     // :14
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    bool ok = true;
    // ring of D trajectory rows in flight (as in k_riccati): at 65 536 problems every SIMD holds one wave, and one step
    // of dependent arithmetic is shorter than an HBM round trip
    constexpr int D = (n + m) <= 8 ? RICCATI_PREFETCH : 1;
    double xr[D][n], ur[D][m];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int tl = a.N - 1 - d > 0 ? a.N - 1 - d : 0;
        load_col<n>(Xp + (size_t)tl * n * TILE, xr[d]);
        load_col<m>(Up + (size_t)tl * m * TILE, ur[d]);
    }
    for (int t0 = a.N - 1; t0 >= 0; t0 -= D) {                                         // :15
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int t = t0 - d;
        if (t < 0) break;
        double x[n], u[m];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xr[d][i];
#pragma unroll
        for (int j = 0; j < m; ++j) u[j] = ur[d][j];
        double A[n * n], Bm[n * m];
        S::A(x, u, a.dt, p, A);                                                        // :21
        S::B(x, u, a.dt, p, Bm);                                                       // :22
        {
            const int tl = t - D > 0 ? t - D : 0;                                      // slot d next holds step t - D
            load_col<n>(Xp + (size_t)tl * n * TILE, xr[d]);
            load_col<m>(Up + (size_t)tl * m * TILE, ur[d]);
        }
        double BtP[m * n], BtPA[m * n], G[m * m], Gi[m * m], K[m * n];
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + j), Bm[r * m + j], P[r * n + c]);
                BtP[j * n + c] = s;
            }
#pragma unroll
        for (int j = 0; j < m; ++j) {
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + c), A[r * n + c], BtP[j * n + r]);
                BtPA[j * n + c] = s;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + l), Bm[r * m + l], BtP[j * n + r]);
                G[j * m + l] = Rk[j * m + l] + s;
            }
        }
        ok = Riccati<S>::inv_spd(G, Gi) && ok;
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
            for (int c = 0; c < n; ++c) {                                              // :24
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) s = __builtin_fma(-Gi[j * m + l], BtPA[l * n + c], s);
                K[j * n + c] = s;
            }
        store_col<m * n>(Kp + (size_t)t * (m * n) * TILE, K);                          // :25
        // :27  P = Q + A'PA - (A'PB) Gi (B'PA);  (A'PB)[i][l] is formed from the rows of A'P
        double Pn[n * n];
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double row[n], apb[m];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + i), A[r * n + i], P[r * n + c]);
                row[c] = s;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + l), Bm[r * m + l], row[r]);
                apb[l] = s;
            }
            double w[m];                                                               // (A'PB Gi)[i][:]
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < m; ++q) s = __builtin_fma(apb[q], Gi[q * m + l], s);
                w[l] = s;
            }
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + c), A[r * n + c], row[r]);
                double corr = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) corr = __builtin_fma(w[l], BtPA[l * n + c], corr);
                Pn[i * n + c] = (Qk[i * n + c] + s) - corr;
            }
        }
#pragma unroll
        for (int i = 0; i < n * n; ++i) P[i] = Pn[i];
      }
    }
    if (!ok) a.status[tile * TILE + lane] = ILQR_STATUS_NUMERICAL;
}

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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(-100 - (int)e_, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum { KID_ROLLOUT = 0, KID_BACKWARD = 1, KID_LINESEARCH = 2, KID_ACCEPT = 3, KID_LTVLQR = 4, KID_FORWARD = 5, KID_COST = 6,
       KID_LINEARIZE = 7 };
constexpr int CKPT_INTERVAL = 64;   // time steps between line-search checkpoints (split pipeline)

}  // namespace

struct ilqr_ws {
    ilqr_desc d;
    int n, m, np, N, W, n_trials, want_trace;
    long long B, Bpad, tiles;
    hipStream_t stream;
    int n_sub;                       // sub-batches of solve(), each a contiguous tile range on its own stream
    hipStream_t sub_stream[4];
    hipEvent_t sub_start, sub_done[4];
    long long sub_tile0[5];
    double *X2, *U2, *CKPT, *LIN;
    int split, grp, ck, n_seg, n_lin;
    double *X, *U, *G, *KFB, *KFF, *XO, *UO, *INIT, *TARGET, *PARAMS, *WTS, *alphas, *alpha_pp, *cost, *new_cost, *dJ, *trace;
    int *active, *iters, *status, *acc_j, *ls_rounds, *accepts, *ls_list, *ls_count;
    void* staging;
    size_t staging_bytes;
    size_t hbm_bytes;
    int state_rows;
    bool profiling;
    struct Ev { hipEvent_t a, b; int kid; };
    std::vector<Ev> events;
    double prof_ms[N_KERNEL_IDS];
    long long prof_cnt[N_KERNEL_IDS];
};

namespace {

KArgs make_args(ilqr_ws* ws, int it)
{
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.N = ws->N; a.it = it; a.n_iter = ws->d.n_iterations; a.zero_first = ws->d.zero_first_state;
    a.n_trials = ws->n_trials; a.state_rows = ws->state_rows; a.n_tiles = ws->tiles; a.tile0 = 0;
    a.dt = ws->d.dt; a.low_red = ws->d.low_expected_reduction; a.armijo_thr = ws->d.armijo_threshold;
    a.quu_reg = ws->d.quu_reg;
    a.X2 = ws->X2; a.U2 = ws->U2; a.CKPT = ws->CKPT; a.LIN = ws->LIN; a.ck = ws->ck; a.n_seg = ws->n_seg;
    a.X = ws->X; a.U = ws->U; a.G = ws->G; a.KFB = ws->KFB; a.KFF = ws->KFF; a.XO = ws->X; a.UO = ws->U;
    a.INIT = ws->INIT; a.TARGET = ws->TARGET; a.PARAMS = ws->PARAMS; a.WTS = ws->WTS; a.alphas = ws->alphas;
    a.alpha_pp = ws->alpha_pp; a.cost = ws->cost; a.new_cost = ws->new_cost; a.dJ = ws->dJ; a.active = ws->active;
    a.iters = ws->iters; a.status = ws->status; a.acc_j = ws->acc_j; a.ls_rounds = ws->ls_rounds;
    a.ls_round = 0; a.ls_list = ws->ls_list; a.ls_count = ws->ls_count;
    a.accepts = ws->accepts; a.trace = ws->want_trace ? ws->trace : nullptr; a.Bpad = ws->Bpad;
    return a;
}

template <class F>
int with_system(int sys, F&& fn)
{
    switch (sys) {
        case 0: return fn(Sys_pendulum{});
        case 1: return fn(Sys_acrobot{});
        case 2: return fn(Sys_cartpole{});
        case 3: return fn(Sys_quadcopter{});
        case 4: return fn(Sys_car{});
        case 5: return fn(Sys_double_integrator{});
        case 6: return fn(Sys_double_pendulum{});
    }
    return fail(-2, "unknown system id %d", sys);
}

struct ProfScope {
    ilqr_ws* ws; int kid; hipStream_t st; hipEvent_t a, b; bool on;
    ProfScope(ilqr_ws* w, int k, hipStream_t s) : ws(w), kid(k), st(s), a(nullptr), b(nullptr), on(w && w->profiling)
    {
        if (on) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(b, st);
            ws->events.push_back({a, b, kid});
        }
    }
};

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-100 - (int)e, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// Line-search launch geometry.  A workgroup covers 8 consecutive problems, i.e. one full 64-byte segment of every
// 512-byte tile row, so the rows it streams are fetched once per CU (W = 64: 8 waves, else 4 waves x 64/W problems;
// W < 32 covers more).  Workgroups are dealt round-robin over the 8 XCDs, so logical neighbours (which share 128-byte
// L2 lines) are mapped to block ids 8 apart = the same XCD's L2.  Placement is a speed matter only.
template <class S, int W>
void launch_ls(hipStream_t st, const KArgs& a, bool ckpt)
{
    const long long waves = a.n_tiles * W;          // tiles * 64 problems * W lanes / 64
    constexpr int wpb = W == 64 ? 8 : 4;
    const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
    if (ckpt) hipLaunchKernelGGL((k_linesearch<S, W, true>), dim3(grid), dim3(64 * wpb), 0, st, a);
    else hipLaunchKernelGGL((k_linesearch<S, W, false>), dim3(grid), dim3(64 * wpb), 0, st, a);
}

int set_device(const ilqr_ws* ws)
{
    HIPCHK(hipSetDevice(ws->d.device));
    return 0;
}

size_t field_elems(const ilqr_ws* ws, int field, int* is_int, long long* tc_out, double** dptr, int** iptr)
{
    const long long N = ws->N, n = ws->n, m = ws->m, np = ws->np;
    *is_int = 0; *dptr = nullptr; *iptr = nullptr;
    long long tc = 0;
    switch (field) {
        case ILQR_F_INIT: tc = n; *dptr = ws->INIT; break;
        case ILQR_F_TARGET: tc = n; *dptr = ws->TARGET; break;
        case ILQR_F_PARAMS: tc = np; *dptr = ws->PARAMS; break;
        case ILQR_F_STATES: tc = (long long)ws->state_rows * n; *dptr = ws->X; break;
        case ILQR_F_INPUTS: tc = N * m; *dptr = ws->U; break;
        case ILQR_F_GUESS: tc = N * m; *dptr = ws->G; break;
        case ILQR_F_KFF: tc = N * m; *dptr = ws->KFF; break;
        case ILQR_F_KFB: tc = N * m * n; *dptr = ws->KFB; break;
        case ILQR_F_COST: tc = 1; *dptr = ws->cost; break;
        case ILQR_F_NEW_COST: tc = 1; *dptr = ws->new_cost; break;
        case ILQR_F_ALPHA: tc = 1; *dptr = ws->alpha_pp; break;
        case ILQR_F_DJ: tc = 3; *dptr = ws->dJ; break;
        case ILQR_F_ITERS: tc = 1; *is_int = 1; *iptr = ws->iters; break;
        case ILQR_F_STATUS: tc = 1; *is_int = 1; *iptr = ws->status; break;
        case ILQR_F_LS_ROUNDS: tc = 1; *is_int = 1; *iptr = ws->ls_rounds; break;
        case ILQR_F_ACCEPTS: tc = 1; *is_int = 1; *iptr = ws->accepts; break;
        case ILQR_F_TRACE: tc = (long long)ws->d.n_iterations * 6; *dptr = ws->trace; break;
        case ILQR_F_STATES_OUT: tc = (long long)ws->state_rows * n; *dptr = ws->XO; break;
        case ILQR_F_INPUTS_OUT: tc = N * m; *dptr = ws->UO; break;
        default: return 0;
    }
    *tc_out = tc;
    return (size_t)tc;
}

int ensure_staging(ilqr_ws* ws, size_t bytes)
{
    if (ws->staging_bytes >= bytes) return 0;
    if (ws->staging) { HIPCHK(hipFree(ws->staging)); ws->staging = nullptr; ws->staging_bytes = 0; }
    HIPCHK(hipMalloc(&ws->staging, bytes));
    ws->staging_bytes = bytes;
    return 0;
}

int ensure_forward_out(ilqr_ws* ws)
{
    if (ws->XO) return 0;
    const size_t xs = (size_t)ws->Bpad * ws->state_rows * ws->n * sizeof(double);
    const size_t us = (size_t)ws->Bpad * ws->N * ws->m * sizeof(double);
    HIPCHK(hipMalloc((void**)&ws->XO, xs));
    HIPCHK(hipMalloc((void**)&ws->UO, us));
    ws->hbm_bytes += xs + us;
    return 0;
}

// every per-problem array is [tile][TC][64] on the device and (B,TC) at the boundary
int do_upload(ilqr_ws* ws, int field, const void* src, bool src_on_device, long long rows)
{
    int rc = set_device(ws);
    if (rc) return rc;
    if (field == ILQR_F_STATES_OUT || field == ILQR_F_INPUTS_OUT) { rc = ensure_forward_out(ws); if (rc) return rc; }
    int is_int; long long tc; double* dp; int* ip;
    if (!field_elems(ws, field, &is_int, &tc, &dp, &ip)) return fail(-3, "upload: unknown field %d", field);
    if (is_int || field == ILQR_F_DJ || field == ILQR_F_TRACE) return fail(-3, "upload: field %d is output only", field);
    if (!dp) return fail(-3, "upload: field %d not allocated", field);
    long long bstride = tc, copy_elems;
    long long tc_eff = tc;
    if (field == ILQR_F_TARGET && !ws->d.target_batched) bstride = 0;
    if (field == ILQR_F_PARAMS && !ws->d.params_batched) bstride = 0;
    if (field == ILQR_F_STATES) {
        const long long r = rows > 0 ? rows : ws->state_rows;
        if (r < ws->N) return fail(-3, "upload: states needs at least N=%d rows per problem, got %lld", ws->N, r);
        // only min(r, state_rows) rows are packed; the source stride is r rows
        bstride = r * ws->n;
        if (r < ws->state_rows) tc_eff = r * ws->n;
    }
    copy_elems = bstride == 0 ? tc : bstride * ws->B;
    const double* dsrc = (const double*)src;
    if (!src_on_device) {
        rc = ensure_staging(ws, (size_t)copy_elems * sizeof(double));
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ws->staging, src, (size_t)copy_elems * sizeof(double), hipMemcpyHostToDevice, ws->stream));
        dsrc = (const double*)ws->staging;
    }
    {
        // tc_eff < tc only for states given with exactly N rows (ltvlqr): the last row of the buffer stays untouched
        const long long total = ws->Bpad * tc_eff;
        hipLaunchKernelGGL(k_pack<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ws->stream, dsrc, dp, ws->B,
                           ws->Bpad, tc_eff, tc, bstride);
    }
    rc = check_launch("k_pack");
    if (rc) return rc;
    if (!src_on_device) HIPCHK(hipStreamSynchronize(ws->stream));   // staging is reused by the next upload
    return 0;
}

int do_download(ilqr_ws* ws, int field, void* dst, bool dst_on_device)
{
    int rc = set_device(ws);
    if (rc) return rc;
    int is_int; long long tc; double* dp; int* ip;
    if (!field_elems(ws, field, &is_int, &tc, &dp, &ip)) return fail(-3, "download: unknown field %d", field);
    if (field == ILQR_F_TRACE && !ws->want_trace) return fail(-3, "download: workspace was created without a trace");
    if (!dp && !ip) return fail(-3, "download: field %d not allocated", field);
    const size_t esz = is_int ? sizeof(int) : sizeof(double);
    const size_t bytes = (size_t)ws->B * tc * esz;
    void* ddst = dst;
    if (!dst_on_device) {
        rc = ensure_staging(ws, bytes);
        if (rc) return rc;
        ddst = ws->staging;
    }
    const long long total = ws->Bpad * tc;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (is_int) {
        hipLaunchKernelGGL(k_unpack<int>, dim3(grid), dim3(256), 0, ws->stream, ip, (int*)ddst, ws->B, ws->Bpad, tc);
    } else {
        hipLaunchKernelGGL(k_unpack<double>, dim3(grid), dim3(256), 0, ws->stream, dp, (double*)ddst, ws->B, ws->Bpad, tc);
    }
    rc = check_launch("k_unpack");
    if (rc) return rc;
    if (!dst_on_device) {
        HIPCHK(hipMemcpyAsync(dst, ws->staging, bytes, hipMemcpyDeviceToHost, ws->stream));
        HIPCHK(hipStreamSynchronize(ws->stream));
    }
    return 0;
}

int fill_i32(ilqr_ws* ws, int* p, int v)
{
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((ws->Bpad + 255) / 256)), dim3(256), 0, ws->stream, p, ws->Bpad, v);
    return check_launch("k_fill_i32");
}

// launch one pass over the whole batch (sub < 0, workspace stream) or over sub-batch `sub` (its own stream)
template <class F>
int run_pass(ilqr_ws* ws, int kid, int it, const char* name, F&& launch, int sub = -1, bool prof = true)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    KArgs a = make_args(ws, it);
    hipStream_t st = ws->stream;
    if (sub >= 0) {
        a.tile0 = ws->sub_tile0[sub];
        a.n_tiles = ws->sub_tile0[sub + 1] - ws->sub_tile0[sub];
        st = ws->sub_stream[sub];
    }
    if (a.n_tiles <= 0) return 0;
    ProfScope ps(prof ? ws : nullptr, kid, st);
    return with_system(ws->d.system_id, [&](auto S) {
        launch(S, a, st);
        return check_launch(name);
    });
}

int ws_linearize(ilqr_ws* ws, bool copy, int sub = -1)
{
    const int chunks = (ws->N + LIN_TS - 1) / LIN_TS;
    return run_pass(ws, KID_LINEARIZE, 0, "k_linearize", [&](auto S, const KArgs& a, hipStream_t st) {
        const unsigned grid = (unsigned)(a.n_tiles * chunks);
        if (copy) hipLaunchKernelGGL((k_linearize<decltype(S), true>), dim3(grid), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_linearize<decltype(S), false>), dim3(grid), dim3(64), 0, st, a);
    }, sub);
}
int ws_riccati(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_BACKWARD, it, "k_riccati", [&](auto S, const KArgs& a, hipStream_t st) {
        using T = decltype(S);
        if (ws->grp) hipLaunchKernelGGL(k_riccati_grp<T>, dim3((unsigned)(a.n_tiles * Grp<T>::GL)), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(k_riccati<T>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    }, sub);
}
int ws_accept_par(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_ACCEPT, it, "k_accept_par", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_accept_par<decltype(S)>, dim3((unsigned)(a.n_tiles * ws->n_seg)), dim3(64), 0, st, a);
    }, sub);
}
int ws_rollout(ilqr_ws* ws, int sub = -1)
{
    return run_pass(ws, KID_ROLLOUT, 0, "k_rollout", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_rollout<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    }, sub);
}
int ws_backward(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_BACKWARD, it, "k_backward", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_backward<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    }, sub);
}
// One launch per round of W trials.  Round 0 runs every active problem in place; a problem none of whose W trials passed
// appends itself to a list, and the next round runs exactly that list, packed into full waves (a round that is mostly
// finished then occupies few SIMDs once each instead of trailing behind half-empty waves).  The host does not know the
// list lengths: every round is launched at full size and the surplus workgroups return on their first load.
int ws_linesearch(ilqr_ws* ws, int sub = -1)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    hipStream_t st = sub >= 0 ? ws->sub_stream[sub] : ws->stream;
    int* cnt = ws->ls_count + (size_t)(sub + 1) * MAX_TRIALS;
    const int rounds = (ws->n_trials + ws->W - 1) / ws->W;
    ProfScope ps(ws, KID_LINESEARCH, st);
    if (rounds > 1) HIPCHK(hipMemsetAsync(cnt, 0, MAX_TRIALS * sizeof(int), st));
    for (int r = 0; r < rounds; ++r) {
        rc = run_pass(ws, KID_LINESEARCH, 0, "k_linesearch", [&](auto S, const KArgs& a0, hipStream_t s) {
            using T = decltype(S);
            KArgs a = a0;
            a.ls_round = r;
            a.ls_count = cnt;
            const bool ck = ws->split != 0;
            switch (ws->W) {
                case 1: launch_ls<T, 1>(s, a, ck); break;
                case 2: launch_ls<T, 2>(s, a, ck); break;
                case 4: launch_ls<T, 4>(s, a, ck); break;
                case 8: launch_ls<T, 8>(s, a, ck); break;
                case 16: launch_ls<T, 16>(s, a, ck); break;
                case 32: launch_ls<T, 32>(s, a, ck); break;
                default: launch_ls<T, 64>(s, a, ck); break;
            }
        }, sub, false);
        if (rc) return rc;
    }
    return 0;
}
int ws_accept(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_ACCEPT, it, "k_forward<accept>", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL((k_forward<decltype(S), true>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    }, sub);
}
// solve() of one sub-batch (or of the whole batch when sub < 0): ilqr.py:162-207
int ws_solve_chain(ilqr_ws* ws, int sub)
{
    int rc;
    if ((rc = ws_rollout(ws, sub))) return rc;                       // :162-164
    if (ws->split) {
        if ((rc = ws_linearize(ws, false, sub))) return rc;          // A_t, B_t of the initial trajectory (:96-97)
        for (int it = 0; it < ws->d.n_iterations; ++it) {            // :170
            if ((rc = ws_riccati(ws, it, sub))) return rc;           // :99-135, :173-179
            if ((rc = ws_linesearch(ws, sub))) return rc;            // :180-202 (+ checkpoints)
            if ((rc = ws_accept_par(ws, it, sub))) return rc;        // :194-207, segments in parallel
            if ((rc = ws_linearize(ws, true, sub))) return rc;       // accepted trajectory back + A_t, B_t for the next pass
        }
        return 0;
    }
    for (int it = 0; it < ws->d.n_iterations; ++it) {                // :170
        if ((rc = ws_backward(ws, it, sub))) return rc;              // :173-179
        if ((rc = ws_linesearch(ws, sub))) return rc;                // :180-202
        if ((rc = ws_accept(ws, it, sub))) return rc;                // :194-207
    }
    return 0;
}

int validate_desc(const ilqr_desc* d)
{
    if (!d) return fail(-1, "null descriptor");
    if (d->system_id < 0 || d->system_id >= N_SYSTEMS) return fail(-2, "unknown system id %d", d->system_id);
    if (d->batch < 1) return fail(-1, "batch must be >= 1 (got %lld)", (long long)d->batch);
    if (d->n_timesteps < 1) return fail(-1, "n_timesteps must be >= 1 (got %d)", d->n_timesteps);
    if (d->n_iterations < 0) return fail(-1, "n_iterations must be >= 0");
    const int w = d->n_alpha_parallel;
    if (w != 0 && (w < 1 || w > 64 || (w & (w - 1)) != 0)) return fail(-1, "n_alpha_parallel must be 0 or a power of two <= 64");
    if (d->n_streams < 0 || d->n_streams > 4) return fail(-1, "n_streams must be 0 (default: 1) to 4");
    if (d->riccati < 0 || d->riccati > 2) return fail(-1, "riccati must be 0 (auto), 1 (lane per problem) or 2 (lane group)");
    if (d->pipeline < 0 || d->pipeline > 2) return fail(-1, "pipeline must be 0 (auto), 1 (fused) or 2 (split)");
    if (!(d->learning_speed > 0.0 && d->learning_speed < 1.0)) return fail(-1, "learning_speed must be in (0,1)");
    if (!(d->low_learning_rate > 0.0 && d->low_learning_rate < 1.0)) return fail(-1, "low_learning_rate must be in (0,1)");
    return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

void ilqr_desc_default(ilqr_desc* d)
{
    memset(d, 0, sizeof(*d));
    d->batch = 1;
    d->n_alpha_parallel = 0;
    d->zero_first_state = 1;
    d->learning_speed = 0.95;
    d->low_learning_rate = 0.05;
    d->low_expected_reduction = 1e-3;
    d->armijo_threshold = 0.1;
    d->quu_reg = 0.0;
    d->pipeline = 0;
    d->n_streams = 0;
    d->riccati = 0;
}

int ilqr_abi_version(void) { return ILQR_ABI_VERSION; }

int ilqr_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

const char* ilqr_last_error(void) { return g_err; }

int ilqr_system_dims(int32_t sys, int32_t* n, int32_t* m, int32_t* np)
{
    if (sys < 0 || sys >= N_SYSTEMS) return fail(-2, "unknown system id %d", sys);
    if (n) *n = SYSTEMS[sys].n;
    if (m) *m = SYSTEMS[sys].m;
    if (np) *np = SYSTEMS[sys].np;
    return 0;
}

int ilqr_system_lookup(const char* name, int32_t* n, int32_t* m, int32_t* np)
{
    if (!name) return fail(-2, "null system name");
    for (int i = 0; i < N_SYSTEMS; ++i)
        if (strcmp(name, SYSTEMS[i].name) == 0) {
            ilqr_system_dims(i, n, m, np);
            return i;
        }
    return fail(-2, "unknown system '%s' (compiled in: pendulum, acrobot, cartpole, quadcopter, car, double_integrator, double_pendulum)", name);
}

int ilqr_dynamics_eval(int32_t sys, int32_t which, int64_t count, const double* x, const double* u, double dt,
                       const double* params, double* out, int32_t device)
{
    if (sys < 0 || sys >= N_SYSTEMS) return fail(-2, "unknown system id %d", sys);
    if (which < 0 || which > 2) return fail(-1, "which must be 0 (f), 1 (A) or 2 (B)");
    if (count < 0) return fail(-1, "negative count");
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(device));
    const int n = SYSTEMS[sys].n, m = SYSTEMS[sys].m, np = SYSTEMS[sys].np;
    const size_t osz = which == 0 ? n : (which == 1 ? n * n : n * m);
    double *dx = nullptr, *du = nullptr, *dp = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void**)&dx, count * n * sizeof(double)));
    HIPCHK(hipMalloc((void**)&du, count * m * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dp, np * sizeof(double)));
    HIPCHK(hipMalloc((void**)&dout, count * osz * sizeof(double)));
    HIPCHK(hipMemcpy(dx, x, count * n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(du, u, count * m * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp, params, np * sizeof(double), hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((count + 63) / 64);
    int rc = with_system(sys, [&](auto S) {
        using T = decltype(S);
        hipLaunchKernelGGL(k_dynamics<T>, dim3(grid), dim3(64), 0, 0, which, (long long)count, dx, du, dt, dp, dout);
        return check_launch("k_dynamics");
    });
    if (rc == 0) {
        hipError_t e = hipMemcpy(out, dout, count * osz * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(-100 - (int)e, "hipMemcpy D2H: %s", hipGetErrorString(e));
    }
    (void)hipFree(dx); (void)hipFree(du); (void)hipFree(dp); (void)hipFree(dout);
    return rc;
}

// ---------------------------------------------------------------------------------------- workspace
int ilqr_ws_create(const ilqr_desc* d, int32_t want_trace, ilqr_ws** out)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!out) return fail(-1, "null output pointer");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(-1, "device %d out of range (%d visible)", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));
    ilqr_ws* ws = new (std::nothrow) ilqr_ws();
    if (!ws) return fail(-4, "out of host memory");
    ws->d = *d;
    ws->n = SYSTEMS[d->system_id].n; ws->m = SYSTEMS[d->system_id].m; ws->np = SYSTEMS[d->system_id].np;
    ws->N = d->n_timesteps; ws->B = d->batch;
    ws->tiles = (ws->B + TILE - 1) / TILE; ws->Bpad = ws->tiles * TILE;
    ws->state_rows = ws->N + 1;
    ws->want_trace = want_trace && d->n_iterations > 0;
    ws->profiling = false;
    memset(ws->prof_ms, 0, sizeof(ws->prof_ms));
    memset(ws->prof_cnt, 0, sizeof(ws->prof_cnt));
    // alpha table by repeated multiplication (ilqr.py:180,202) and the number of trials with alpha > low_lr (:185)
    double alphas[MAX_TRIALS];
    double lr = 1.0;
    int n_trials = 0;
    for (int j = 0; j < MAX_TRIALS; ++j) {
        alphas[j] = lr;
        if (lr > d->low_learning_rate && n_trials == j) n_trials = j + 1;
        lr = d->learning_speed * lr;
    }
    if (n_trials == MAX_TRIALS && lr > d->low_learning_rate) {
        delete ws;
        return fail(-1, "learning_speed/low_learning_rate need more than %d line-search trials", MAX_TRIALS);
    }
    ws->n_trials = n_trials;
    int W = d->n_alpha_parallel;
    if (W == 0) {   // auto: about two waves per SIMD (256 CUs x 4 SIMDs), 8 to 32 trials per problem and round
        W = 8;
        while (W < 32 && ws->Bpad * (long long)(W * 2) / TILE <= 4096) W *= 2;
    }
    ws->W = W;
    // pipeline: with few tiles the horizon-sequential kernels occupy one wave per tile and leave most of the 1024 SIMDs
    // idle, so the parallel parts (accept re-roll, Jacobians) are split off; large batches fill the chip either way and
    // keep the fused kernels (no LIN / checkpoint traffic).
    ws->split = d->pipeline == 2 || (d->pipeline == 0 && ws->tiles < 1024);
    if (d->n_iterations == 0) ws->split = 0;
    // Riccati recursion: one lane per problem while V_xx fits its registers (n <= 4), a lane group with LDS hand-offs
    // beyond (always through the split schedule, which provides LIN); riccati = 1 / 2 forces lane / group
    ws->grp = d->riccati == 2 || (d->riccati == 0 && ws->n > 4);
    if (ws->grp) ws->split = 1;
    ws->ck = CKPT_INTERVAL;
    ws->n_seg = (ws->N + ws->ck - 1) / ws->ck;
    ws->n_lin = with_system(d->system_id, [&](auto S) { return Lin<decltype(S)>::n_lin; });
    const size_t Bp = (size_t)ws->Bpad, n = ws->n, m = ws->m, np = ws->np, N = ws->N;
    struct { void** p; size_t bytes; } allocs[] = {
        {(void**)&ws->X, Bp * (N + 1) * n * 8}, {(void**)&ws->U, Bp * N * m * 8}, {(void**)&ws->G, Bp * N * m * 8}, {(void**)&ws->KFB, Bp * N * m * n * 8},
        {(void**)&ws->KFF, Bp * N * m * 8}, {(void**)&ws->INIT, Bp * n * 8}, {(void**)&ws->TARGET, Bp * n * 8},
        {(void**)&ws->PARAMS, Bp * np * 8}, {(void**)&ws->WTS, (2 * n * n + m * m) * 8}, {(void**)&ws->alphas, MAX_TRIALS * 8},
        {(void**)&ws->alpha_pp, Bp * 8}, {(void**)&ws->cost, Bp * 8}, {(void**)&ws->new_cost, Bp * 8}, {(void**)&ws->dJ, 3 * Bp * 8},
        {(void**)&ws->active, Bp * 4}, {(void**)&ws->iters, Bp * 4}, {(void**)&ws->status, Bp * 4}, {(void**)&ws->acc_j, Bp * 4},
        {(void**)&ws->ls_rounds, Bp * 4}, {(void**)&ws->accepts, Bp * 4}, {(void**)&ws->ls_list, 2 * Bp * 4},
        {(void**)&ws->ls_count, 5 * MAX_TRIALS * 4},
        {(void**)&ws->trace, ws->want_trace ? Bp * (size_t)d->n_iterations * 6 * 8 : 0},
        {(void**)&ws->X2, ws->split ? Bp * (N + 1) * n * 8 : 0}, {(void**)&ws->U2, ws->split ? Bp * N * m * 8 : 0},
        {(void**)&ws->CKPT, ws->split ? Bp * (size_t)ws->n_seg * n * TILE * 8 : 0},
        {(void**)&ws->LIN, ws->split ? Bp * N * (size_t)ws->n_lin * 8 : 0},
    };
    ws->hbm_bytes = 0;
    for (auto& al : allocs) {
        *al.p = nullptr;
        if (al.bytes == 0) continue;
        hipError_t e = hipMalloc(al.p, al.bytes);
        if (e != hipSuccess) {
            const int code = fail(-100 - (int)e, "hipMalloc of %zu bytes failed: %s", al.bytes, hipGetErrorString(e));
            ilqr_ws_destroy(ws);
            return code;
        }
        ws->hbm_bytes += al.bytes;
    }
    hipError_t e = hipStreamCreate(&ws->stream);
    if (e != hipSuccess) { ilqr_ws_destroy(ws); return fail(-100 - (int)e, "hipStreamCreate: %s", hipGetErrorString(e)); }
    ws->n_sub = d->n_streams > 0 ? d->n_streams : 1;
    if (ws->n_sub > ws->tiles) ws->n_sub = (int)ws->tiles;
    for (int sb = 0; sb <= ws->n_sub; ++sb) ws->sub_tile0[sb] = ws->tiles * sb / ws->n_sub;
    if (ws->n_sub > 1) {
        HIPCHK(hipEventCreateWithFlags(&ws->sub_start, hipEventDisableTiming));
        for (int sb = 0; sb < ws->n_sub; ++sb) {
            HIPCHK(hipStreamCreate(&ws->sub_stream[sb]));
            HIPCHK(hipEventCreateWithFlags(&ws->sub_done[sb], hipEventDisableTiming));
        }
    }
    HIPCHK(hipMemcpy(ws->alphas, alphas, sizeof(alphas), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(ws->WTS, 0, (2 * n * n + m * m) * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->dJ, 0, 3 * Bp * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->KFB, 0, Bp * N * m * n * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->KFF, 0, Bp * N * m * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->X, 0, Bp * (N + 1) * n * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->U, 0, Bp * N * m * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->G, 0, Bp * N * m * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->cost, 0, Bp * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->new_cost, 0, Bp * 8, ws->stream));
    HIPCHK(hipMemsetAsync(ws->TARGET, 0, Bp * n * 8, ws->stream));
    *out = ws;
    rc = ilqr_ws_reset(ws);
    if (rc) { ilqr_ws_destroy(ws); *out = nullptr; return rc; }
    HIPCHK(hipStreamSynchronize(ws->stream));
    return 0;
}

int ilqr_ws_destroy(ilqr_ws* ws)
{
    if (!ws) return 0;
    (void)hipSetDevice(ws->d.device);
    if (ws->stream) (void)hipStreamSynchronize(ws->stream);
    for (auto& ev : ws->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    void* ptrs[] = {ws->X2, ws->U2, ws->CKPT, ws->LIN, ws->X, ws->U, ws->G, ws->KFB, ws->KFF, ws->XO, ws->UO, ws->INIT, ws->TARGET, ws->PARAMS, ws->WTS, ws->alphas,
                    ws->alpha_pp, ws->cost, ws->new_cost, ws->dJ, ws->trace, ws->active, ws->iters, ws->status, ws->acc_j,
                    ws->ls_rounds, ws->accepts, ws->ls_list, ws->ls_count, ws->staging};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (ws->n_sub > 1) {
        if (ws->sub_start) (void)hipEventDestroy(ws->sub_start);
        for (int sb = 0; sb < ws->n_sub; ++sb) {
            if (ws->sub_stream[sb]) { (void)hipStreamSynchronize(ws->sub_stream[sb]); (void)hipStreamDestroy(ws->sub_stream[sb]); }
            if (ws->sub_done[sb]) (void)hipEventDestroy(ws->sub_done[sb]);
        }
    }
    if (ws->stream) (void)hipStreamDestroy(ws->stream);
    delete ws;
    return 0;
}

int ilqr_ws_set_weights(ilqr_ws* ws, const double* Qk, const double* Rk, const double* QT)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    const size_t n = ws->n, m = ws->m;
    std::vector<double> w(2 * n * n + m * m, 0.0);
    if (Qk) memcpy(w.data(), Qk, n * n * 8);
    if (Rk) memcpy(w.data() + n * n, Rk, m * m * 8);
    if (QT) memcpy(w.data() + n * n + m * m, QT, n * n * 8);
    HIPCHK(hipMemcpyAsync(ws->WTS, w.data(), w.size() * 8, hipMemcpyHostToDevice, ws->stream));
    HIPCHK(hipStreamSynchronize(ws->stream));
    return 0;
}

int ilqr_ws_upload(ilqr_ws* ws, int32_t field, const void* src, int64_t rows)
{
    if (!ws || !src) return fail(-1, "null argument");
    return do_upload(ws, field, src, false, rows);
}
int ilqr_ws_upload_device(ilqr_ws* ws, int32_t field, const void* src, int64_t rows)
{
    if (!ws || !src) return fail(-1, "null argument");
    return do_upload(ws, field, src, true, rows);
}
int ilqr_ws_download(ilqr_ws* ws, int32_t field, void* dst)
{
    if (!ws || !dst) return fail(-1, "null argument");
    return do_download(ws, field, dst, false);
}
int ilqr_ws_download_device(ilqr_ws* ws, int32_t field, void* dst)
{
    if (!ws || !dst) return fail(-1, "null argument");
    return do_download(ws, field, dst, true);
}

int ilqr_ws_reset(ilqr_ws* ws)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    if ((rc = fill_i32(ws, ws->active, 1))) return rc;
    if ((rc = fill_i32(ws, ws->iters, 0))) return rc;
    if ((rc = fill_i32(ws, ws->status, ILQR_STATUS_MAX_ITER))) return rc;
    if ((rc = fill_i32(ws, ws->acc_j, -1))) return rc;
    if ((rc = fill_i32(ws, ws->ls_rounds, 0))) return rc;
    if ((rc = fill_i32(ws, ws->accepts, 0))) return rc;
    if (ws->want_trace) {
        const long long cnt = ws->Bpad * (long long)ws->d.n_iterations * 6;
        hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ws->stream, ws->trace, cnt,
                           __builtin_nan(""));
        if ((rc = check_launch("k_fill_f64"))) return rc;
    }
    return 0;
}

int ilqr_ws_rollout(ilqr_ws* ws) { return ws_rollout(ws); }
int ilqr_ws_cost(ilqr_ws* ws)
{
    return run_pass(ws, KID_COST, 0, "k_cost", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_cost<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
}
int ilqr_ws_backward(ilqr_ws* ws, int32_t iteration)
{
    if (ws && ws->split) {
        int rc = ws_linearize(ws, false);
        return rc ? rc : ws_riccati(ws, iteration);
    }
    return ws_backward(ws, iteration);
}
int ilqr_ws_linesearch(ilqr_ws* ws) { return ws_linesearch(ws); }
int ilqr_ws_accept(ilqr_ws* ws, int32_t iteration)
{
    if (ws && ws->split) return fail(-1, "ilqr_ws_accept is the fused schedule's accept step; this workspace uses the split "
                                         "schedule (create it with pipeline = 1 to drive the passes one by one)");
    return ws_accept(ws, iteration);
}
int ilqr_ws_forward(ilqr_ws* ws, int32_t with_cost)
{
    (void)with_cost;   // the cost of the new trajectory is always produced (NEW_COST)
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    if ((rc = ensure_forward_out(ws))) return rc;
    return run_pass(ws, KID_FORWARD, 0, "k_forward", [&](auto S, const KArgs& a0, hipStream_t st) {
        KArgs a = a0;
        a.XO = ws->XO; a.UO = ws->UO;
        hipLaunchKernelGGL((k_forward<decltype(S), false>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
}
int ilqr_ws_ltvlqr(ilqr_ws* ws)
{
    return run_pass(ws, KID_LTVLQR, 0, "k_ltvlqr", [&](auto S, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_ltvlqr<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
}

int ilqr_ws_solve(ilqr_ws* ws)
{
    int rc;
    if ((rc = ilqr_ws_reset(ws))) return rc;
    if (ws->n_sub <= 1) return ws_solve_chain(ws, -1);
    // sub-batches are independent tile ranges: each runs its own launch chain on its own stream, so one sub-batch's
    // horizon-sequential kernels (few waves) overlap another's line search (many waves)
    HIPCHK(hipEventRecord(ws->sub_start, ws->stream));
    for (int sb = 0; sb < ws->n_sub; ++sb) HIPCHK(hipStreamWaitEvent(ws->sub_stream[sb], ws->sub_start, 0));
    for (int sb = 0; sb < ws->n_sub; ++sb)
        if ((rc = ws_solve_chain(ws, sb))) return rc;
    for (int sb = 0; sb < ws->n_sub; ++sb) {
        HIPCHK(hipEventRecord(ws->sub_done[sb], ws->sub_stream[sb]));
        HIPCHK(hipStreamWaitEvent(ws->stream, ws->sub_done[sb], 0));
    }
    return 0;
}

int ilqr_ws_sync(ilqr_ws* ws)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ws->stream));
    return 0;
}

int ilqr_ws_profile(ilqr_ws* ws, int32_t enable)
{
    if (!ws) return fail(-1, "null workspace");
    for (auto& ev : ws->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    ws->events.clear();
    memset(ws->prof_ms, 0, sizeof(ws->prof_ms));
    memset(ws->prof_cnt, 0, sizeof(ws->prof_cnt));
    ws->profiling = enable != 0;
    return 0;
}

int ilqr_ws_profile_read(ilqr_ws* ws, double* ms, int64_t* launches)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ws->stream));
    for (auto& ev : ws->events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) { ws->prof_ms[ev.kid] += t; ws->prof_cnt[ev.kid] += 1; }
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
    }
    ws->events.clear();
    for (int i = 0; i < N_KERNEL_IDS; ++i) { if (ms) ms[i] = ws->prof_ms[i]; if (launches) launches[i] = ws->prof_cnt[i]; }
    return 0;
}

int ilqr_ws_info(ilqr_ws* ws, int32_t* W, int32_t* n_trials, int64_t* hbm_bytes)
{
    if (!ws) return fail(-1, "null workspace");
    if (W) *W = ws->W;
    if (n_trials) *n_trials = ws->n_trials;
    if (hbm_bytes) *hbm_bytes = (int64_t)ws->hbm_bytes;
    return 0;
}

int ilqr_ws_pipeline(ilqr_ws* ws)
{
    if (!ws) return fail(-1, "null workspace");
    return ws->split ? 2 : 1;
}

// ------------------------------------------------------------------------------ one-shot host API
#define TRY(expr) do { int rc_ = (expr); if (rc_) { if (ws) ilqr_ws_destroy(ws); return rc_; } } while (0)

int ilqr_rollout_batched(const ilqr_desc* d, const double* init_state, const double* inputs_in, const double* params,
                         double* states, double* inputs)
{
    ilqr_ws* ws = nullptr;
    if (!init_state || !inputs_in || !params || !states || !inputs) return fail(-1, "null argument");
    TRY(ilqr_ws_create(d, 0, &ws));
    TRY(ilqr_ws_upload(ws, ILQR_F_INIT, init_state, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_GUESS, inputs_in, 0));
    TRY(ilqr_ws_rollout(ws));
    TRY(ilqr_ws_download(ws, ILQR_F_STATES, states));
    TRY(ilqr_ws_download(ws, ILQR_F_INPUTS, inputs));
    return ilqr_ws_destroy(ws);
}

int ilqr_cost_batched(const ilqr_desc* d, const double* states, const double* inputs, const double* Rk, const double* QT,
                      const double* target, double* cost)
{
    ilqr_ws* ws = nullptr;
    if (!states || !inputs || !Rk || !QT || !target || !cost) return fail(-1, "null argument");
    TRY(ilqr_ws_create(d, 0, &ws));
    TRY(ilqr_ws_set_weights(ws, nullptr, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_TARGET, target, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_STATES, states, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_INPUTS, inputs, 0));
    TRY(ilqr_ws_cost(ws));
    TRY(ilqr_ws_download(ws, ILQR_F_COST, cost));
    return ilqr_ws_destroy(ws);
}

int ilqr_backward_batched(const ilqr_desc* d, const double* states, const double* inputs, const double* target,
                          const double* Qk, const double* Rk, const double* QT, const double* params, double* k, double* K,
                          double* dJ, int32_t* status)
{
    ilqr_ws* ws = nullptr;
    if (!states || !inputs || !target || !Qk || !Rk || !QT || !params || !k || !K || !dJ) return fail(-1, "null argument");
    TRY(ilqr_ws_create(d, 0, &ws));
    TRY(ilqr_ws_set_weights(ws, Qk, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_TARGET, target, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_STATES, states, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_INPUTS, inputs, 0));
    // the stand-alone pass must not apply solve()'s "optimal" stop test: low_red = 0 never triggers it
    ws->d.low_expected_reduction = 0.0;
    TRY(ilqr_ws_backward(ws, 0));
    TRY(ilqr_ws_download(ws, ILQR_F_KFF, k));
    TRY(ilqr_ws_download(ws, ILQR_F_KFB, K));
    TRY(ilqr_ws_download(ws, ILQR_F_DJ, dJ));
    if (status) TRY(ilqr_ws_download(ws, ILQR_F_STATUS, status));
    return ilqr_ws_destroy(ws);
}

int ilqr_forward_batched(const ilqr_desc* d, const double* init_state, const double* ref_states, const double* ref_inputs,
                         const double* k, const double* K, const double* learning_rate, const double* params,
                         double* states, double* inputs, const double* Rk, const double* QT, const double* target,
                         double* cost)
{
    ilqr_ws* ws = nullptr;
    if (!init_state || !ref_states || !ref_inputs || !k || !K || !learning_rate || !params || !states || !inputs)
        return fail(-1, "null argument");
    const bool with_cost = cost != nullptr;
    if (with_cost && (!Rk || !QT || !target)) return fail(-1, "cost requested without Rk/QT/target");
    TRY(ilqr_ws_create(d, 0, &ws));
    if (with_cost) {
        TRY(ilqr_ws_set_weights(ws, nullptr, Rk, QT));
        TRY(ilqr_ws_upload(ws, ILQR_F_TARGET, target, 0));
    }
    TRY(ilqr_ws_upload(ws, ILQR_F_INIT, init_state, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_STATES, ref_states, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_INPUTS, ref_inputs, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_KFF, k, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_KFB, K, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_ALPHA, learning_rate, 0));
    TRY(ilqr_ws_forward(ws, with_cost));
    TRY(ilqr_ws_download(ws, ILQR_F_STATES_OUT, states));
    TRY(ilqr_ws_download(ws, ILQR_F_INPUTS_OUT, inputs));
    if (with_cost) TRY(ilqr_ws_download(ws, ILQR_F_NEW_COST, cost));
    return ilqr_ws_destroy(ws);
}

int ilqr_solve_batched(const ilqr_desc* d, const double* init_state, const double* target, const double* guess,
                       const double* Qk, const double* Rk, const double* QT, const double* params, double* states,
                       double* inputs, double* k, double* K, double* cost, int32_t* iters, int32_t* status, double* trace)
{
    ilqr_ws* ws = nullptr;
    if (!init_state || !target || !guess || !Qk || !Rk || !QT || !params) return fail(-1, "null argument");
    TRY(ilqr_ws_create(d, trace != nullptr, &ws));
    TRY(ilqr_ws_set_weights(ws, Qk, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_INIT, init_state, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_TARGET, target, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_GUESS, guess, 0));
    TRY(ilqr_ws_solve(ws));
    if (states) TRY(ilqr_ws_download(ws, ILQR_F_STATES, states));
    if (inputs) TRY(ilqr_ws_download(ws, ILQR_F_INPUTS, inputs));
    if (k) TRY(ilqr_ws_download(ws, ILQR_F_KFF, k));
    if (K) TRY(ilqr_ws_download(ws, ILQR_F_KFB, K));
    if (cost) TRY(ilqr_ws_download(ws, ILQR_F_COST, cost));
    if (iters) TRY(ilqr_ws_download(ws, ILQR_F_ITERS, iters));
    if (status) TRY(ilqr_ws_download(ws, ILQR_F_STATUS, status));
    if (trace && ws->want_trace) TRY(ilqr_ws_download(ws, ILQR_F_TRACE, trace));
    return ilqr_ws_destroy(ws);
}

int ltvlqr_batched(const ilqr_desc* d, int32_t n_state_rows, const double* states, const double* inputs, const double* Qk,
                   const double* Rk, const double* QT, const double* params, double* K)
{
    ilqr_ws* ws = nullptr;
    if (!states || !inputs || !Qk || !Rk || !QT || !params || !K) return fail(-1, "null argument");
    if (!d) return fail(-1, "null descriptor");
    if (n_state_rows < d->n_timesteps) return fail(-1, "states needs at least N=%d rows (got %d)", d->n_timesteps, n_state_rows);
    TRY(ilqr_ws_create(d, 0, &ws));
    TRY(ilqr_ws_set_weights(ws, Qk, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_STATES, states, n_state_rows));
    TRY(ilqr_ws_upload(ws, ILQR_F_INPUTS, inputs, 0));
    TRY(ilqr_ws_ltvlqr(ws));
    TRY(ilqr_ws_download(ws, ILQR_F_KFB, K));
    return ilqr_ws_destroy(ws);
}

}  // extern "C"
