// kernels_common.h -- tile constants, the compiled systems, kernel arguments, the per-step Riccati algebra, small load/store helpers
// Part of ilqr_hip.hip (included inside its anonymous namespace, in this order: kernels_common.h,
// kernels_backward.h, kernels_forward.h, kernels_util.h); not a stand-alone header.

constexpr int TILE = 64;          // problems per tile = lanes per wavefront
constexpr int MAX_TRIALS = 64;    // line-search trials kept in the alpha table
constexpr int N_KERNEL_IDS = 8;

// ------------------------------------------------------------------------------------------------
// systems
// ------------------------------------------------------------------------------------------------
#ifndef ILQR_LIN_SKIP_DT
#define ILQR_LIN_SKIP_DT 1
#endif
#ifndef ILQR_TRIG_CARRY
#define ILQR_TRIG_CARRY 1
#endif
#define ILQR_DEFINE_SYSTEM(NAME, UP, CARRY, LIBTRIG)                                                              \
    struct Sys_##NAME {                                                                                           \
        /* re-seeds of the carried pairs (and everything else in this system's dynamics) use the library's sin / cos */ \
        static constexpr bool lib_trig = (LIBTRIG);                                                               \
        static constexpr int n = ILQR_DYN_##UP##_N, m = ILQR_DYN_##UP##_M, np = ILQR_DYN_##UP##_NP;               \
        static constexpr int id = ILQR_DYN_##UP##_ID;                                                             \
        /* time-step loops carry the sin/cos pairs of these state components from step to step (TrigCarry) */     \
        static constexpr int n_ang = (ILQR_TRIG_CARRY && (CARRY)) ? ILQR_DYN_##UP##_F_NANG : 0;                   \
        static __device__ __forceinline__ int ang(int k) { return ilqr_dyn_##NAME##_f_ang[k]; }                   \
        static __device__ __forceinline__ void f_sc(const double* x, const double* u, double dt, const double* p, \
                                                    const double* sc, double* o)                                  \
        { ilqr_dyn_##NAME##_f_sc(x, u, dt, p, sc, o); }                                                           \
        static constexpr const char* A_pat = ilqr_dyn_##NAME##_A_pat;                                             \
        static constexpr const char* B_pat = ilqr_dyn_##NAME##_B_pat;                                             \
        static __device__ __forceinline__ char a_kind(int i) { return ilqr_dyn_##NAME##_A_pat[i]; }               \
        static __device__ __forceinline__ char b_kind(int i) { return ilqr_dyn_##NAME##_B_pat[i]; }               \
        /* general entries that are exactly dt are uniform: LIN does not store them (ILQR_LIN_SKIP_DT) */         \
        static constexpr const char* A_dt = ilqr_dyn_##NAME##_A_dt;                                               \
        static constexpr const char* B_dt = ilqr_dyn_##NAME##_B_dt;                                               \
        static constexpr bool a_is_dt(int i) { return ILQR_LIN_SKIP_DT && ilqr_dyn_##NAME##_A_dt[i] == '1'; }     \
        static constexpr bool b_is_dt(int i) { return ILQR_LIN_SKIP_DT && ilqr_dyn_##NAME##_B_dt[i] == '1'; }     \
        static constexpr bool a_stored(int i) { return ilqr_dyn_##NAME##_A_pat[i] == 'x' && !a_is_dt(i); }        \
        static constexpr bool b_stored(int i) { return ilqr_dyn_##NAME##_B_pat[i] == 'x' && !b_is_dt(i); }        \
        static __device__ __forceinline__ void f(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_f(x, u, dt, p, o); }              \
        static __device__ __forceinline__ void A(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_A(x, u, dt, p, o); }              \
        static __device__ __forceinline__ void B(const double* x, const double* u, double dt, const double* p,   \
                                                 double* o) { ilqr_dyn_##NAME##_B(x, u, dt, p, o); }              \
        /* the Jacobians with the carried pairs handed in (a trajectory walked step by step: k_ltvlqr) */         \
        static __device__ __forceinline__ void A_sc(const double* x, const double* u, double dt, const double* p, \
                                                    const double* sc, double* o)                                  \
        { ilqr_dyn_##NAME##_A_sc(x, u, dt, p, sc, o); }                                                           \
        static __device__ __forceinline__ void B_sc(const double* x, const double* u, double dt, const double* p, \
                                                    const double* sc, double* o)                                  \
        { ilqr_dyn_##NAME##_B_sc(x, u, dt, p, sc, o); }                                                           \
    };
ILQR_DEFINE_SYSTEM(pendulum, PENDULUM, 1, false)
ILQR_DEFINE_SYSTEM(acrobot, ACROBOT, 1, false)
ILQR_DEFINE_SYSTEM(cartpole, CARTPOLE, 1, false)
// The quadcopter carries nothing: its 21-iteration example solve amplifies last-bit rounding most, and
// with carried pairs (re-seeded from the library's sin / cos) the final K came out 2.6e-10 from the reference with the
// carry in the accept step's rollouts only and 1.4e-8 with it in the line search as well -- the noise floor of that
// solve straddles the 1e-8 bar, the plain library evaluation (3.8e-9) is the variant that makes no choice.  7 % of C5.
#ifndef ILQR_QUAD_CARRY
#define ILQR_QUAD_CARRY 0
#endif
ILQR_DEFINE_SYSTEM(quadcopter, QUADCOPTER, ILQR_QUAD_CARRY, ILQR_QUAD_TRIG_LIB != 0)
ILQR_DEFINE_SYSTEM(car, CAR, 1, false)
ILQR_DEFINE_SYSTEM(double_integrator, DOUBLE_INTEGRATOR, 1, false)
ILQR_DEFINE_SYSTEM(double_pendulum, DOUBLE_PENDULUM, 1, false)
ILQR_DEFINE_SYSTEM(floating_base, FLOATING_BASE, 1, false)

struct SysInfo { const char* name; int n, m, np; };
constexpr int N_SYSTEMS = 8;
const SysInfo SYSTEMS[N_SYSTEMS] = {{"pendulum", 2, 1, 3}, {"acrobot", 4, 1, 5}, {"cartpole", 4, 1, 4}, {"quadcopter", 12, 4, 8},
                                    {"car", 4, 2, 1}, {"double_integrator", 2, 1, 1}, {"double_pendulum", 4, 2, 5}, {"floating_base", 12, 12, 17}};
// One lane per problem holds Q_uu, its inverse and K in registers: up to 4 inputs.  Beyond (floating base, 12 inputs) only the
// lane-group recursion (k_riccati_grp) exists, and the kernels below are not instantiated.
template <class S> constexpr bool lane_recursion_ok = S::m <= 4;

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
    double low_lr;         // ilqr.py:204: a search that ends with learning_rate < low_lr ends the solve
    double alpha_end;      // learning_rate an exhausted search ends with: the first alpha_j <= search_floor (:185)
    double* X;             // [tile][state_rows][n][64]
    double* U;             // [tile][N][m][64]
    const double* G;       // [tile][N][m][64] initial guess (read by k_rollout only)
    // tracking cost (ilqr_desc.tracking, matlab/smooth-ilqr/ilqr_mpc.m): per-step targets of the horizon window
    const double* XD;      // [tile][N+1][n][64] x_des: rows 0..N-1 stage targets, row N the terminal target (ilqr_mpc.m:203,213,247,255)
    const double* UD;      // [tile][N][m][64]   u_des (ilqr_mpc.m:204,256)
    int fwd_to_cost;       // k_forward<false>: the cost of the new trajectory also becomes the current cost (ilqr_mpc.m:127-134)
    double* KFB;           // [tile][N][m*n][64]
    double* KFF;           // [tile][N][m][64]
    double* XB;            // split pipeline: [tile][segment][n][64] reference-state row at every segment boundary, saved before the in-place accept
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
    int ls_round;          // line search: round r evaluates trials [ls_j0, ls_j0 + W) (W is the launched kernel's width)
    int ls_j0;
    int cost_monotone;     // R_k and Q_T are diagonal with non-negative entries: the running cost of a rollout never decreases
    int all_problems;      // k_linearize / the ltvlqr recursion: ignore the per-problem `active` flags of solve()
    int ls_last;           // no round follows: a search that fails here has run out of learning rate
    int* ls_list;          // [2][Bpad] problems still searching after a round (ping-pong by round parity)
    int* ls_count;         // [MAX_TRIALS] length of the list each round leaves (zeroed before round 0)
    // overlapped accept / Riccati (ilqr_desc.overlap): k_accept_lin covers segments [seg_lo, seg_lo + seg_cnt), k_riccati_quad
    // steps [t_lo, t_hi) with the value function handed from chunk to chunk through VST
    int seg_lo, seg_cnt, t_lo, t_hi, piped;
    double* VST;           // [tile][n*n + n + 4][TILE]: V_xx, V_x, the three expected-reduction sums, ok
    int* accepts;
    double* trace;         // [tile][n_iter][6][64] or nullptr
    long long Bpad;
};

// ------------------------------------------------------------------------------------------------
// TrigCarry: sin/cos of the angle states, carried along a rollout instead of evaluated afresh every step.
// A rollout step changes an angle by d = x_{t+1} - x_t, a few 1e-3 rad (dt * velocity): sin and cos of the new angle follow
// from the old pair by angle addition with sin d, cos d - 1 from short Taylor polynomials -- ~20 fp64 instructions per
// angle instead of ~55 (range reduction, two minimax kernels, quadrant selects); the acrobot's line-search step drops
// from ~265 to ~195 instructions.
//   * d is formed by one subtraction: its rounding error is 2^-53 |d| ~ 1e-19, far below the pair's own rounding.
//   * |d| <= 3/32 (TRIG_CARRY_MAX_STEP): the neglected Taylor terms are d^11/11! < 2e-19 (sin) and d^10/10! < 2e-17 (cos).  The update is
//     written s + (small), c + (small): ONE rounding of the pair per step (half an ulp), i.e. a random walk of a few ulp
//     between re-seeds -- and in f every trig value is scaled by dt before it meets the state (<= 1e-18 of an O(1) state
//     per step: below the rounding of the state update itself).
//   * re-seeded with the full evaluation (ilqr_sincos_fast; ocml for huge / non-finite angles) at every multiple of the
//     checkpoint interval -- so a rollout resumed from a line-search checkpoint (k_accept_lin) sees exactly the pairs
//     the trial saw: same bits whichever kernel, schedule or segment computes a step -- and at once for any lane whose
//     step is larger than 3/32 rad or not finite (diverging trials).
// The Jacobians A_t, B_t keep their direct evaluations (their entries are not scaled against a large addend).
// ------------------------------------------------------------------------------------------------
// Largest |angle step| the recurrence takes before the pair is evaluated afresh.  The series below stop at d^9 (sin) and d^8
// (cos): at 3/32 rad their truncation errors are d^11/11! = 1.2e-19 and d^10/10! = 1.4e-17, an eighth of the half-ulp of a
// value near 1 and far below the recurrence's own roundings.  (Round 3: 1/16 before; the cartpole at dt = 0.005 steps past
// 1/16 rad in ordinary swing-up motion and re-evaluated -- the whole wave waiting -- on 5 % of its line-search steps:
// 131 072 cartpoles 10.33 -> 10.0 ms of search per pass; the acrobot at dt = 0.001 hardly ever does.)
#ifndef TRIG_CARRY_MAX_STEP
#define TRIG_CARRY_MAX_STEP 0.09375
#endif
template <class S>
struct TrigCarry {
    static constexpr int K = S::n_ang;
    double sc[K > 0 ? 2 * K : 1];

    static __device__ __forceinline__ void full(const double x, double& s, double& c)
    {
        if constexpr (S::lib_trig) {
            s = sin(x); c = cos(x);
        } else {
            ilqr_sincos_fast(x, s, c);
            if (!(__builtin_fabs(x) < 1.0e6)) { const ilqr_sc r = ilqr_sincos_slow(x); s = r.sinv; c = r.cosv; }
        }
    }
    __device__ __forceinline__ void seed(const double (&x)[S::n])
    {
#pragma unroll
        for (int k = 0; k < K; ++k) full(x[S::ang(k)], sc[2 * k], sc[2 * k + 1]);
    }
    // the pairs of xn from those of xo; reseed (uniform): evaluate afresh.  dont_care: this lane's rollout no longer
    // matters (a line-search trial already known to fail): it keeps the recurrence whatever its step, so that it does not
    // drag its wave through the full evaluation at every step.  A NaN step is left to the recurrence as well (NaN in, NaN
    // out, as the full evaluation would give).
    __device__ __forceinline__ void advance(const double (&xo)[S::n], const double (&xn)[S::n], const bool reseed,
                                            const bool dont_care = false)
    {
        if constexpr (K > 0) {
            constexpr double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0, S3 = -1.0 / 5040.0, S4 = 1.0 / 362880.0;
            constexpr double C1 = -0.5, C2 = 1.0 / 24.0, C3 = -1.0 / 720.0, C4 = 1.0 / 40320.0;
            bool ok = !reseed;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double d = xn[S::ang(k)] - xo[S::ang(k)];
                ok = ok && (dont_care || !(__builtin_fabs(d) > TRIG_CARRY_MAX_STEP));
                const double z = d * d;
                const double sdt = (d * z) * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S4, S3), S2), S1);   // sin d - d
                const double cm1 = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C4, C3), C2), C1);         // cos d - 1
                const double s = sc[2 * k], c = sc[2 * k + 1];
                const double sd = d + sdt;                                             // sin d (its rounding: 2^-53 |d| < 1e-17)
                sc[2 * k] = s + __builtin_fma(c, sd, s * cm1);
                sc[2 * k + 1] = c + __builtin_fma(-s, sd, c * cm1);
            }
            if (!ok) seed(xn);
        }
    }
};

// ilqr.py:204 after the line search of one problem: accj >= 0 accepted trial (learning_rate stays alpha_accj), accj < 0
// exhausted (learning_rate has dropped to alpha_end).  True: "low learning rate", the solve of this problem ends.
__device__ __forceinline__ bool ls_ends_solve(const KArgs& a, const int accj)
{
    return (accj >= 0 ? a.alphas[accj] : a.alpha_end) < a.low_lr;
}

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

    // np.linalg.inv (ilqr.py:106, ltvlqr.py:24) for a matrix the Cholesky path below cannot take: LU with partial (row)
    // pivoting, then one solve per identity column -- LAPACK getrf/getri's rule, the arithmetic of oracle/ilqr_oracle.c
    // inv_small.  Returns false only on an exactly zero pivot, the one case in which numpy raises LinAlgError; a NaN or
    // indefinite Q is inverted like any other (NaN in, NaN out: the line search then rejects every trial and the
    // problem ends "low learning rate" as it does in the reference).  Row swaps are selects: everything stays in VGPRs.
    static __device__ __forceinline__ bool inv_lu(const double (&Q)[m * m], double (&Qi)[m * m])
    {
        double a[m * m];
        int piv[m];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < m * m; ++i) a[i] = Q[i];
#pragma unroll
        for (int c = 0; c < m; ++c) {
            int pr = c;
            double best = __builtin_fabs(a[c * m + c]);
#pragma unroll
            for (int r = c + 1; r < m; ++r) {
                const double v = __builtin_fabs(a[r * m + c]);
                if (v > best) { best = v; pr = r; }
            }
            piv[c] = pr;
            ok = ok && !(best == 0.0);
#pragma unroll
            for (int r = c + 1; r < m; ++r)
#pragma unroll
                for (int j = 0; j < m; ++j) {
                    const double x = a[c * m + j], y = a[r * m + j];
                    a[c * m + j] = pr == r ? y : x;
                    a[r * m + j] = pr == r ? x : y;
                }
#pragma unroll
            for (int r = c + 1; r < m; ++r) {
                a[r * m + c] = a[r * m + c] / a[c * m + c];
#pragma unroll
                for (int j = c + 1; j < m; ++j) a[r * m + j] = a[r * m + j] - a[r * m + c] * a[c * m + j];
            }
        }
#pragma unroll
        for (int col = 0; col < m; ++col) {
            double b[m];
#pragma unroll
            for (int i = 0; i < m; ++i) b[i] = i == col ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < m; ++c)
#pragma unroll
                for (int r = c + 1; r < m; ++r) {
                    const double x = b[c], y = b[r];
                    b[c] = piv[c] == r ? y : x;
                    b[r] = piv[c] == r ? x : y;
                }
#pragma unroll
            for (int i = 0; i < m; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) b[i] = b[i] - a[i * m + j] * b[j];
#pragma unroll
            for (int i = m - 1; i >= 0; --i) {
#pragma unroll
                for (int j = i + 1; j < m; ++j) b[i] = b[i] - a[i * m + j] * b[j];
                b[i] = b[i] / a[i * m + i];
            }
#pragma unroll
            for (int i = 0; i < m; ++i) Qi[i * m + col] = b[i];
        }
        return ok;
    }

    // inverse of the m x m matrix Q (row-major).  m == 1: reciprocal (what LAPACK getri does for 1x1).
    // m > 1: Cholesky L L^T, inv = L^-T L^-1 while Q is positive definite (every Q_uu of a well-posed problem); a Q that
    // is not (indefinite weights, a diverged trajectory's NaN) takes inv_lu, numpy's own rule.  Returns false only where
    // np.linalg.inv raises: an exactly zero pivot.
    static __device__ __forceinline__ bool inv_spd(const double (&Q)[m * m], double (&Qi)[m * m])
    {
        if constexpr (m == 1) {
            Qi[0] = 1.0 / Q[0];
            return Q[0] != 0.0;
        } else {
            if (inv_chol(Q, Qi)) return true;
            return inv_lu(Q, Qi);
        }
    }
    static __device__ __forceinline__ bool inv_chol(const double (&Q)[m * m], double (&Qi)[m * m])
    {
        {
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
    // TRACK (tracking cost, matlab/smooth-ilqr/ilqr_mpc.m:260-272): l_x = Q_k (x - x_des) is handed in as `lx` and added to
    // Q_x; `u` is then u - u_des, so that l_u = R_k (u - u_des) comes out of the same product (:263).
    template <bool TRACK = false>
    static __device__ __forceinline__ bool step(const double (&A)[n * n], const double (&Bm)[n * m], const double (&u)[m],
                                                const double (&Qk)[n * n], const double (&Rk)[m * m], double quu_reg,
                                                double (&Vx)[n], double (&Vxx)[n * n], double (&k)[m], double (&K)[m * n],
                                                double& g_out, double& h_out, const double (&lx)[n])
    {
        double Qx[n], Qu[m], BtV[m * n], Qux[m * n], Quu[m * m], Qxx[n * n], Quu_inv[m * m];
        // :99  Q_x = A.T @ V_x   (l_x = Q_k @ 0 = 0)
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = 0.0; bool first = true;
#pragma unroll
            for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + i), A[r * n + i], Vx[r]);
            Qx[i] = TRACK ? lx[i] + s : s;                                             // ilqr_mpc.m:271
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

    static __device__ __forceinline__ bool step(const double (&A)[n * n], const double (&Bm)[n * m], const double (&u)[m],
                                                const double (&Qk)[n * n], const double (&Rk)[m * m], double quu_reg,
                                                double (&Vx)[n], double (&Vxx)[n * n], double (&k)[m], double (&K)[m * n],
                                                double& g_out, double& h_out)
    {
        const double none[n] = {};
        return step<false>(A, Bm, u, Qk, Rk, quu_reg, Vx, Vxx, k, K, g_out, h_out, none);
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
// the same for the time-step loops.  With many inputs (floating base: 144 entries of R_k) and an R_k that set_weights found
// diagonal (a.cost_monotone) only the diagonal is walked: the skipped terms are fma(u_i, 0, s) = s, so the value is the
// same bit for bit for finite inputs (and non-finite either way for a diverged trial).
template <int m>
__device__ __forceinline__ double stage_input_cost(const KArgs& a, const double (&u)[m], const double (&Rk)[m * m])
{
    if constexpr (m > 4) {
        if (a.cost_monotone) {
            double c = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) c = __builtin_fma(__builtin_fma(u[j], Rk[j * m + j], 0.0), u[j], c);
            return c;
        }
    }
    return input_cost<m>(u, Rk);
}
// tracking stage cost (matlab/smooth-ilqr/ilqr_mpc.m:207-209): (u_des - u)' R (u_des - u) + (x_des - x)' Q_k (x_des - x)
template <int n, int m>
__device__ __forceinline__ double tracking_stage_cost(const double (&x)[n], const double (&u)[m], const double (&xd)[n],
                                                      const double (&ud)[m], const double (&Qk)[n * n], const double (&Rk)[m * m])
{
    double du[m], dx[n];
#pragma unroll
    for (int j = 0; j < m; ++j) du[j] = ud[j] - u[j];
#pragma unroll
    for (int i = 0; i < n; ++i) dx[i] = xd[i] - x[i];
    double cx = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) s = __builtin_fma(dx[i], Qk[i * n + j], s);
        cx = __builtin_fma(s, dx[j], cx);
    }
    return input_cost<m>(du, Rk) + cx;
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

