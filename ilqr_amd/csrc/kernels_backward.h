// kernels_backward.h -- backward side: k_backward, k_linearize, k_riccati, k_riccati_grp, k_ltvlqr (ilqr.py:64-135, ltvlqr.py:3-28)
// Part of ilqr_hip.hip (included inside its anonymous namespace, in this order: kernels_common.h,
// kernels_backward.h, kernels_forward.h, kernels_util.h); not a stand-alone header.
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
// TRACK: the tracking cost of matlab/smooth-ilqr/ilqr_mpc.m:236-308 -- per-step targets x_des, u_des (XD, UD), l_x = Q_k (x - x_des),
// l_u = R_k (u - u_des), terminal target = the last row of x_des (:247).
template <class S, bool TRACK = false>
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
    const double* XDp = TRACK ? a.XD + (size_t)tile * a.state_rows * n * TILE + lane : nullptr;
    const double* UDp = TRACK ? a.UD + (size_t)tile * a.N * m * TILE + lane : nullptr;
    load_col<n>(TRACK ? XDp + (size_t)a.N * n * TILE : a.TARGET + (size_t)tile * n * TILE + lane, target);
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
        if constexpr (TRACK) {
            double xd[n], ud[m], ex[n], du[m], lx[n];
            load_col<n>(XDp + (size_t)t * n * TILE, xd);
            load_col<m>(UDp + (size_t)t * m * TILE, ud);
#pragma unroll
            for (int i = 0; i < n; ++i) ex[i] = x[i] - xd[i];
#pragma unroll
            for (int j = 0; j < m; ++j) du[j] = u[j] - ud[j];
            // ilqr_mpc.m:260  l_x = Q_k (x - x_des).  (Round 2 rolled this product for n > 4 after "wrong gains at -O3": round 3
            // found the recursion itself unstable for the Q_k that was used -- the CPU restatement against itself, states 1e-13
            // apart, ends O(1) apart after the quadcopter's 400 steps once Q_k ~ 0.1 -- and both forms equal to the restatement
            // wherever it is conditioned: tests/test_gpu_mpc.py::test_tracking_backward_pass_and_cost_with_a_dense_Qk.)
#pragma unroll
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < n; ++c) s = __builtin_fma(Qk[i * n + c], ex[c], s);
                lx[i] = s;
            }
            ok = Riccati<S>::template step<true>(A, Bm, du, Qk, Rk, a.quu_reg, Vx, Vxx, k, K, g, h, lx) && ok;
        } else {
            ok = Riccati<S>::step(A, Bm, u, Qk, Rk, a.quu_reg, Vx, Vxx, k, K, g, h) && ok;
        }
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

template <class S>
struct Lin {
    // stored per step: the general entries of A_t, B_t except those that are exactly dt (uniform; S::a_stored)
    static constexpr int count_a() { int c = 0; for (int i = 0; i < S::n * S::n; ++i) c += S::a_stored(i); return c; }
    static constexpr int count_b() { int c = 0; for (int i = 0; i < S::n * S::m; ++i) c += S::b_stored(i); return c; }
    static constexpr int nA = count_a();
    static constexpr int nB = count_b();
    static constexpr int n_lin = nA + nB;
};

constexpr int LIN_TS = 8;   // time steps per wave in k_linearize

// the general entries of A_t, B_t, in pattern order, into the step's LIN record
template <class S>
__device__ __forceinline__ void store_lin(double* l, const double (&A)[S::n * S::n], const double (&Bm)[S::n * S::m])
{
    constexpr int n = S::n, m = S::m;
    int e = 0;
#pragma unroll
    for (int i = 0; i < n * n; ++i) if (S::a_stored(i)) { l[(size_t)e * TILE] = A[i]; ++e; }
#pragma unroll
    for (int i = 0; i < n * m; ++i) if (S::b_stored(i)) { l[(size_t)e * TILE] = Bm[i]; ++e; }
}

template <class S>
__global__ void __launch_bounds__(64) k_linearize(const KArgs a)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const int chunks = (a.N + LIN_TS - 1) / LIN_TS;
    const long long tile = a.tile0 + blockIdx.x / chunks;
    const int chunk = (int)(blockIdx.x % chunks);
    const int lane = threadIdx.x;
    const long long b = tile * TILE + lane;
    const bool run = a.all_problems != 0 || a.active[b] != 0;
    if (__ballot(run) == 0ull) return;
    if (!run) return;
    double p[np];
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    double* Lp = a.LIN + (size_t)tile * a.N * Lin<S>::n_lin * TILE + lane;
    const int t0 = chunk * LIN_TS;
    const int t1 = t0 + LIN_TS < a.N ? t0 + LIN_TS : a.N;
    for (int t = t0; t < t1; ++t) {
        double x[n], u[m];
        load_col<n>(Xp + (size_t)t * n * TILE, x);
        load_col<m>(Up + (size_t)t * m * TILE, u);
        double A[n * n], Bm[n * m];
        S::A(x, u, a.dt, p, A);                                                        // :96
        S::B(x, u, a.dt, p, Bm);                                                       // :97
        store_lin<S>(Lp + (size_t)t * Lin<S>::n_lin * TILE, A, Bm);
    }
}

template <class S>
__device__ __forceinline__ void load_lin(const double* l, double (&A)[S::n * S::n], double (&Bm)[S::n * S::m], const double dt)
{
    constexpr int n = S::n, m = S::m;
    int e = 0;
#pragma unroll
    for (int i = 0; i < n * n; ++i) {
        const char kd = S::a_kind(i);
        if (S::a_stored(i)) { A[i] = l[(size_t)e * TILE]; ++e; } else A[i] = S::a_is_dt(i) ? dt : (kd == '1' ? 1.0 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < n * m; ++i) {
        const char kd = S::b_kind(i);
        if (S::b_stored(i)) { Bm[i] = l[(size_t)e * TILE]; ++e; } else Bm[i] = S::b_is_dt(i) ? dt : (kd == '1' ? 1.0 : 0.0);
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
    auto one_step = [&](const int t, const double (&A)[n * n], const double (&Bm)[n * m], const double (&u)[m], auto&& refill) {
        double k[m], K[m * n], g, h;
        ok = Riccati<S>::step(A, Bm, u, Qk, Rk, a.quu_reg, Vx, Vxx, k, K, g, h) && ok;
        refill();
        if (act) {
            store_col<m>(kp + (size_t)t * m * TILE, k);                                // :111
            store_col<m * n>(Kp + (size_t)t * (m * n) * TILE, K);                      // :112
        }
        const double cur = g + h;                                                      // :117
        ecr_g += g; ecr_h += h; ecr += cur;                                            // :119-121
    };
    // the first N mod D steps one by one, then whole groups of D (no exit inside the unrolled group: an early exit
    // there costs a register shuffle per step)
    int t0 = a.N - 1;
    for (const int rem = a.N % D; t0 >= a.N - rem; --t0) {
        double A1[n * n], B1[n * m], u1[m];
        load_lin<S>(Lp + (size_t)t0 * Lin<S>::n_lin * TILE, A1, B1, a.dt);
        load_col<m>(Up + (size_t)t0 * m * TILE, u1);
        one_step(t0, A1, B1, u1, [] {});
    }
    double Ar[D][n * n], Br[D][n * m], ur[D][m];
    if (t0 >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_lin<S>(Lp + (size_t)(t0 - d) * Lin<S>::n_lin * TILE, Ar[d], Br[d], a.dt);
            load_col<m>(Up + (size_t)(t0 - d) * m * TILE, ur[d]);
        }
    }
    for (; t0 >= 0; t0 -= D) {                                                         // :82
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 - d;
            one_step(t, Ar[d], Br[d], ur[d], [&] {
                const int tl = t - D > 0 ? t - D : 0;                                  // slot d next holds step t - D
                load_lin<S>(Lp + (size_t)tl * Lin<S>::n_lin * TILE, Ar[d], Br[d], a.dt);
                load_col<m>(Up + (size_t)tl * m * TILE, ur[d]);
            });
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
    // m > 4 (floating base: 12 inputs): Q_uu no longer fits every lane's registers m x m -- lane l owns COLUMN l of Q_uu,
    // the factorisation and the solves for K and k run distributed over the row of lanes (see the kernel), and the
    // columns of B_t come through a third transpose buffer
    static constexpr bool BIGM = m > 4;
    static constexpr int oT = 0, oAT = oT + n * LD, oBT = oAT + n * LD, SLOT = (((BIGM ? oBT + m * LD : oAT + n * LD) + 1) | 1);
    static_assert(n <= GL, "k_riccati_grp: one lane per state");
    static_assert(!BIGM || m < GL, "k_riccati_grp: one lane per input and one spare lane for the feed-forward solve");
};

// index of dense entry e of A (or B) in the step's LIN row; -1 structural zero, -2 structural one, -3 exactly dt (not stored)
template <class S, bool IS_B>
__device__ __forceinline__ int lin_index(const int e)
{
    constexpr int cnt_all = IS_B ? S::n * S::m : S::n * S::n;
    int cnt = IS_B ? Lin<S>::nA : 0;
    int res = -1;
#pragma unroll
    for (int q = 0; q < cnt_all; ++q) {
        const char kd = IS_B ? S::b_kind(q) : S::a_kind(q);
        const bool st = IS_B ? S::b_stored(q) : S::a_stored(q);
        if (q == e) res = st ? cnt : (kd == 'x' ? -3 : (kd == '1' ? -2 : -1));
        if (st) ++cnt;
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
// Groups of up to four such instructions behind ONE s_nop (accumulators are early-clobber operands: no input of the group may
// share a register with one of them).  The wait states are needed once per group (the group is one asm
// statement: no register copy can come between its instructions), and an s_nop is an issue slot like any other -- with
// one per DPP instruction a third of the group kernel's instruction stream was s_nop.
//   fmac_row_acc<L>: K accumulators, acc_k += (lane L of src_k) * own
//   fmac_row_dot<L>: one accumulator, acc += (lane L of src_k) * own_k for k = 0, 1, ... in this order
#define ILQR_DPP_FMAC(acc, src, own, lane) "v_fmac_f64_dpp " acc ", " src ", " own " row_newbcast:" lane " row_mask:0xf bank_mask:0xf"
template <int L>
__device__ __forceinline__ void fmac_row_acc(double& a0, double& a1, const double s0, const double s1, const double o)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%2", "%4", "%5") "\n\t" ILQR_DPP_FMAC("%1", "%3", "%4", "%5")
        : "+&v"(a0), "+&v"(a1) : "v"(s0), "v"(s1), "v"(o), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row_acc(double& a0, double& a1, double& a2, const double s0, const double s1, const double s2,
                                             const double o)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%3", "%6", "%7") "\n\t" ILQR_DPP_FMAC("%1", "%4", "%6", "%7") "\n\t"
        ILQR_DPP_FMAC("%2", "%5", "%6", "%7")
        : "+&v"(a0), "+&v"(a1), "+&v"(a2) : "v"(s0), "v"(s1), "v"(s2), "v"(o), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row_acc(double& a0, double& a1, double& a2, double& a3, const double s0, const double s1,
                                             const double s2, const double s3, const double o)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%4", "%8", "%9") "\n\t" ILQR_DPP_FMAC("%1", "%5", "%8", "%9") "\n\t"
        ILQR_DPP_FMAC("%2", "%6", "%8", "%9") "\n\t" ILQR_DPP_FMAC("%3", "%7", "%8", "%9")
        : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(o), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row_dot(double& acc, const double s0, const double s1, const double o0, const double o1)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%3", "%5") "\n\t" ILQR_DPP_FMAC("%0", "%2", "%4", "%5")
        : "+&v"(acc) : "v"(s0), "v"(s1), "v"(o0), "v"(o1), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row_dot(double& acc, const double s0, const double s1, const double s2, const double o0,
                                             const double o1, const double o2)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%4", "%7") "\n\t" ILQR_DPP_FMAC("%0", "%2", "%5", "%7") "\n\t"
        ILQR_DPP_FMAC("%0", "%3", "%6", "%7")
        : "+&v"(acc) : "v"(s0), "v"(s1), "v"(s2), "v"(o0), "v"(o1), "v"(o2), "n"(L));
}
template <int L>
__device__ __forceinline__ void fmac_row_dot(double& acc, const double s0, const double s1, const double s2, const double s3,
                                             const double o0, const double o1, const double o2, const double o3)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%5", "%9") "\n\t" ILQR_DPP_FMAC("%0", "%2", "%6", "%9") "\n\t"
        ILQR_DPP_FMAC("%0", "%3", "%7", "%9") "\n\t" ILQR_DPP_FMAC("%0", "%4", "%8", "%9")
        : "+&v"(acc) : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "n"(L));
}
// acc[OFF + k * STRIDE] += (lane L of src[k]) * own for k < M, in groups of up to four
template <int L, int M, int STRIDE = 1, int OFF = 0, int K0 = 0, int NA, int NS>
__device__ __forceinline__ void fmac_row_vec(double (&acc)[NA], const double (&src)[NS], const double own)
{
    constexpr int K = M - K0 < 4 ? M - K0 : 4;
    constexpr int e0 = OFF + K0 * STRIDE, e1 = e0 + STRIDE, e2 = e1 + STRIDE, e3 = e2 + STRIDE;
    if constexpr (K == 1) fmac_row<L>(acc[e0], src[K0], own);
    else if constexpr (K == 2) fmac_row_acc<L>(acc[e0], acc[e1], src[K0], src[K0 + 1], own);
    else if constexpr (K == 3) fmac_row_acc<L>(acc[e0], acc[e1], acc[e2], src[K0], src[K0 + 1], src[K0 + 2], own);
    else if constexpr (K == 4) fmac_row_acc<L>(acc[e0], acc[e1], acc[e2], acc[e3], src[K0], src[K0 + 1], src[K0 + 2], src[K0 + 3], own);
    if constexpr (K0 + K < M) fmac_row_vec<L, M, STRIDE, OFF, K0 + K>(acc, src, own);
}
// acc += (lane L of src[k]) * own[k] for k = 0 .. M-1, in this order, in groups of up to four
template <int L, int M, int K0 = 0>
__device__ __forceinline__ void fmac_row_sum(double& acc, const double (&src)[M], const double (&own)[M])
{
    constexpr int K = M - K0 < 4 ? M - K0 : 4;
    if constexpr (K == 1) fmac_row<L>(acc, src[K0], own[K0]);
    else if constexpr (K == 2) fmac_row_dot<L>(acc, src[K0], src[K0 + 1], own[K0], own[K0 + 1]);
    else if constexpr (K == 3) fmac_row_dot<L>(acc, src[K0], src[K0 + 1], src[K0 + 2], own[K0], own[K0 + 1], own[K0 + 2]);
    else if constexpr (K == 4) fmac_row_dot<L>(acc, src[K0], src[K0 + 1], src[K0 + 2], src[K0 + 3], own[K0], own[K0 + 1], own[K0 + 2], own[K0 + 3]);
    if constexpr (K0 + K < M) fmac_row_sum<L, M, K0 + K>(acc, src, own);
}
// position of the k-th general ('x') entry in row r of A (IS_B: of B) -- the column phase groups a row's entries
template <class S, bool IS_B>
constexpr int row_x_count(const int r)
{
    int c = 0;
    for (int i = 0; i < (IS_B ? S::m : S::n); ++i) c += (IS_B ? S::B_pat[r * S::m + i] : S::A_pat[r * S::n + i]) == 'x' ? 1 : 0;
    return c;
}
template <class S, bool IS_B>
constexpr int row_x_index(const int r, const int k)
{
    int c = 0;
    for (int i = 0; i < (IS_B ? S::m : S::n); ++i)
        if ((IS_B ? S::B_pat[r * S::m + i] : S::A_pat[r * S::n + i]) == 'x') { if (c == k) return i; ++c; }
    return 0;
}
// acc[i_k] += (lane r of src[i_k]) * own over the general entries i_k of row r, in groups of up to four
template <class S, bool IS_B, int r, int K0 = 0, int NA>
__device__ __forceinline__ void fmac_row_pattern(double (&acc)[NA], const double (&src)[NA], const double own)
{
    constexpr int M = row_x_count<S, IS_B>(r);
    if constexpr (K0 < M) {
        constexpr int K = M - K0 < 4 ? M - K0 : 4;
        constexpr int i0 = row_x_index<S, IS_B>(r, K0), i1 = row_x_index<S, IS_B>(r, K0 + 1), i2 = row_x_index<S, IS_B>(r, K0 + 2),
                      i3 = row_x_index<S, IS_B>(r, K0 + 3);
        if constexpr (K == 1) fmac_row<r>(acc[i0], src[i0], own);
        else if constexpr (K == 2) fmac_row_acc<r>(acc[i0], acc[i1], src[i0], src[i1], own);
        else if constexpr (K == 3) fmac_row_acc<r>(acc[i0], acc[i1], acc[i2], src[i0], src[i1], src[i2], own);
        else fmac_row_acc<r>(acc[i0], acc[i1], acc[i2], acc[i3], src[i0], src[i1], src[i2], src[i3], own);
        fmac_row_pattern<S, IS_B, r, K0 + K>(acc, src, own);
    }
}

// one accumulator, one source register, lanes L0, L1, ...: acc += (lane L_k of src) * own_k, in this order
template <int L0, int L1>
__device__ __forceinline__ void fmac_lanes(double& acc, const double src, const double o0, const double o1)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%2", "%4") "\n\t" ILQR_DPP_FMAC("%0", "%1", "%3", "%5")
        : "+&v"(acc) : "v"(src), "v"(o0), "v"(o1), "n"(L0), "n"(L1));
}
template <int L0, int L1, int L2>
__device__ __forceinline__ void fmac_lanes(double& acc, const double src, const double o0, const double o1, const double o2)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%2", "%5") "\n\t" ILQR_DPP_FMAC("%0", "%1", "%3", "%6") "\n\t"
        ILQR_DPP_FMAC("%0", "%1", "%4", "%7")
        : "+&v"(acc) : "v"(src), "v"(o0), "v"(o1), "v"(o2), "n"(L0), "n"(L1), "n"(L2));
}
template <int L0, int L1, int L2, int L3>
__device__ __forceinline__ void fmac_lanes(double& acc, const double src, const double o0, const double o1, const double o2,
                                           const double o3)
{
    asm("s_nop 1\n\t" ILQR_DPP_FMAC("%0", "%1", "%2", "%6") "\n\t" ILQR_DPP_FMAC("%0", "%1", "%3", "%7") "\n\t"
        ILQR_DPP_FMAC("%0", "%1", "%4", "%8") "\n\t" ILQR_DPP_FMAC("%0", "%1", "%5", "%9")
        : "+&v"(acc) : "v"(src), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "n"(L0), "n"(L1), "n"(L2), "n"(L3));
}
// column j of A walked over its rows r in order: general entries acc += (lane r of src) * own[r] in runs of up to four that
// no unit entry interrupts (a unit entry is acc += own[r] and keeps its place in the order of the sum)
template <class S>
constexpr int col_next(const int j, int r)            // first row >= r whose entry in column j is not a structural zero
{
    while (r < S::n && S::A_pat[r * S::n + j] == '0') ++r;
    return r;
}
template <class S, int j, int r0 = 0>
__device__ __forceinline__ void fmac_col_pattern(double& acc, const double src, const double (&own)[S::n])
{
    constexpr int n = S::n;
    constexpr int a = col_next<S>(j, r0);
    if constexpr (a < n) {
        if constexpr (S::A_pat[a * n + j] == '1') {
            acc = acc + own[a];
            fmac_col_pattern<S, j, a + 1>(acc, src, own);
        } else {
            constexpr int b = col_next<S>(j, a + 1);
            constexpr bool xb = b < n && S::A_pat[(b < n ? b : 0) * n + j] == 'x';
            constexpr int c = xb ? col_next<S>(j, b + 1) : n;
            constexpr bool xc = xb && c < n && S::A_pat[(c < n ? c : 0) * n + j] == 'x';
            constexpr int d = xc ? col_next<S>(j, c + 1) : n;
            constexpr bool xd = xc && d < n && S::A_pat[(d < n ? d : 0) * n + j] == 'x';
            if constexpr (xd) { fmac_lanes<a, b, c, d>(acc, src, own[a], own[b], own[c], own[d]); fmac_col_pattern<S, j, d + 1>(acc, src, own); }
            else if constexpr (xc) { fmac_lanes<a, b, c>(acc, src, own[a], own[b], own[c]); fmac_col_pattern<S, j, c + 1>(acc, src, own); }
            else if constexpr (xb) { fmac_lanes<a, b>(acc, src, own[a], own[b]); fmac_col_pattern<S, j, b + 1>(acc, src, own); }
            else { fmac_row<a>(acc, src, own[a]); fmac_col_pattern<S, j, a + 1>(acc, src, own); }
        }
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// k_riccati_quad: the same recursion as k_riccati with FOUR lanes per problem (n <= 4), for batches that leave the chip
// mostly idle (4096 problems = 64 waves of k_riccati on 1024 SIMDs, a pure instruction-issue chain per step).
//   * lane c of a quad owns COLUMN c: V_xx[:, c], (A^T V_xx)[:, c], (B^T V_xx)[:, c], Q_xx[:, c], Q_ux[:, c], K[:, c],
//     V_x[c].  The two places where a column needs the other lanes' columns -- (B^T V_xx) A, (A^T V_xx) A -- and the
//     new V_x are all-gathers inside the quad: v_mov_b32_dpp quad_perm (DPP on 64-bit operands only has row_newbcast).
//   * A_t, B_t, u_t are loaded whole by every lane (the quad reads the same addresses); in addition lane c loads ITS
//     column of A_t through per-lane offsets, structural zeros / ones patched in from per-lane constants, and uses it
//     densely: a product with an exact 0 or 1 rounds like the skipped / plain term of Riccati<S>::step, so K, k and the
//     expected reductions are bit-identical to k_riccati for finite data (tested).
//   * the small quantities (Q_u, Q_uu, its inverse, k, K, Q_ux) are computed by every lane.
// A workgroup is one wave = 16 problems; a tile is 4 workgroups.
// ------------------------------------------------------------------------------------------------
#ifndef RICCATI_QUAD_PRIO
#define RICCATI_QUAD_PRIO 1
#endif
#ifndef RICCATI_QUAD_PREFETCH
#define RICCATI_QUAD_PREFETCH 6
#endif
template <int Q>
__device__ __forceinline__ double bcast_quad(const double v)
{
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);                           // quad_perm:[Q,Q,Q,Q]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// dense dot product in the order of acc_term: first term a product, the rest fused
template <int CNT>
__device__ __forceinline__ double dot_dense(const double (&a)[CNT], const double (&b)[CNT])
{
    double s = a[0] * b[0];
#pragma unroll
    for (int r = 1; r < CNT; ++r) s = __builtin_fma(a[r], b[r], s);
    return s;
}

// DEEP: the chunks that run alongside the accept step see several times the memory latency (its 3.4 TB/s of traffic
// queue ahead of this kernel's loads): a ring of 10 steps instead of 6 -- slower alone (1.35 -> 1.41 ms, more register
// traffic), faster where it is used (62.2 -> 60.5 ms per overlapped solve)
#ifndef RICCATI_QUAD_PREFETCH_DEEP
#define RICCATI_QUAD_PREFETCH_DEEP 10
#endif
// (the body is a device function of the workgroup index so that k_accept_riccati -- kernels_forward.h -- can run it next to
// the accept step's workgroups in ONE launch)
template <class S, bool DEEP>
__device__ __forceinline__ void riccati_quad_body(const KArgs& a, const unsigned bid)
{
    constexpr int n = S::n, m = S::m, NL = Lin<S>::n_lin;
    static_assert(n <= 4, "k_riccati_quad: one lane per column, four lanes per problem");
    const long long tile = a.tile0 + (bid >> 2);
    const int lane = threadIdx.x;
    const int c = (lane & 3) < n ? (lane & 3) : n - 1;                                 // lanes beyond n shadow the last column
    const bool owner = (lane & 3) < n;
    const int bl = ((bid & 3) << 4) + (lane >> 2);                                     // problem within the tile
    const long long b = tile * TILE + bl;
    // overlapped with the accept step of the previous iteration (a.piped): that step's seg-0 block has not yet retired the
    // problems whose search ended the solve (ilqr.py:204), so they are masked here the way their cleared `active` would
    const bool act = a.active[b] != 0 && (!a.piped || !ls_ends_solve(a, a.acc_j[b]));
    if (__ballot(act) == 0ull) return;
    const bool first_chunk = a.t_hi >= a.N, last_chunk = a.t_lo <= 0;
#if RICCATI_QUAD_PRIO
    // one wave per SIMD on the critical path, sharing the SIMD with the accept step's memory-bound waves: issue first
    if (a.piped) __builtin_amdgcn_s_setprio(3);
#endif
    double* vst = a.VST + (size_t)tile * (n * n + n + 4) * TILE + bl;
    double target[n], Qk[n * n], Rk[m * m], QT[n * n], Vx[n], Vc[n], Qkc[n];
    load_col<n>(a.TARGET + (size_t)tile * n * TILE + bl, target);
    load_uniform<n * n>(a.WTS, Qk);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);                                    // :67
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + bl;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + bl;
    const double* Lp = a.LIN + (size_t)tile * a.N * NL * TILE + bl;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + bl;
    double* kp = a.KFF + (size_t)tile * a.N * m * TILE + bl;
    if (first_chunk) terminal_value<S>(Xp + (size_t)a.N * n * TILE, target, QT, Vx);
    // this lane's column of Q_T (= V_xx), of Q_k, and where its column of A_t sits in a LIN record
    int aoff[n];
    bool aisx[n];
    double acst[n];
#pragma unroll
    for (int r = 0; r < n; ++r) {
        Vc[r] = 0.0; Qkc[r] = 0.0; aoff[r] = 0; aisx[r] = false; acst[r] = 0.0;
#pragma unroll
        for (int q = 0; q < n; ++q)
            if (c == q) {
                Vc[r] = QT[r * n + q];
                Qkc[r] = Qk[r * n + q];
                const int li = lin_index<S, false>(r * n + q);
                aisx[r] = li >= 0;
                aoff[r] = li >= 0 ? li * TILE : 0;
                acst[r] = li == -2 ? 1.0 : (li == -3 ? a.dt : 0.0);
            }
    }
    // rows of A with at least one general entry (only those are loaded per lane) / with only general entries
    constexpr auto row_any_x = [](int r) { bool v = false; for (int q = 0; q < n; ++q) v = v || S::a_stored(r * n + q); return v; };
    constexpr auto row_all_x = [](int r) { bool v = true; for (int q = 0; q < n; ++q) v = v && S::a_stored(r * n + q); return v; };
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    bool ok = true;
    if (!first_chunk) {                                                                // the chunk above left off here
#pragma unroll
        for (int r = 0; r < n; ++r) { Vc[r] = vst[(size_t)(r * n + c) * TILE]; Vx[r] = vst[(size_t)(n * n + r) * TILE]; }
        ecr = vst[(size_t)(n * n + n) * TILE]; ecr_g = vst[(size_t)(n * n + n + 1) * TILE]; ecr_h = vst[(size_t)(n * n + n + 2) * TILE];
        ok = vst[(size_t)(n * n + n + 3) * TILE] != 0.0;
    }
    // (m > 1: the deep ring does not fit the register file next to a 2 x 2 inverse -- 700 B of spills per lane)
    constexpr int D = DEEP && m == 1 ? RICCATI_QUAD_PREFETCH_DEEP : RICCATI_QUAD_PREFETCH;
    auto fetch = [&](const int tl, double (&A)[n * n], double (&Bm)[n * m], double (&u)[m], double (&Acr)[n]) {
        const double* l = Lp + (size_t)tl * NL * TILE;
        load_lin<S>(l, A, Bm, a.dt);
        load_col<m>(Up + (size_t)tl * m * TILE, u);
#pragma unroll
        for (int r = 0; r < n; ++r) Acr[r] = row_any_x(r) ? l[aoff[r]] : 0.0;
    };
    // one time step; refill() is called as soon as the step's inputs are consumed
    auto one_step = [&](const int t, const double (&A)[n * n], const double (&Bm)[n * m], const double (&u)[m],
                        const double (&Acr)[n], auto&& refill) {
        double acol[n];
#pragma unroll
        for (int r = 0; r < n; ++r) acol[r] = !row_any_x(r) ? acst[r] : (row_all_x(r) ? Acr[r] : (aisx[r] ? Acr[r] : acst[r]));
        // :99  Q_x[c] = (A.T V_x)[c]
        const double Qxc = dot_dense<n>(acol, Vx);
        // :93,:100  Q_u = R_k u + B.T V_x
        double Qu[m], BtVc[m], BtV[m * n], Qux[m * n], Quxc[m], Quu[m * m], Quu_inv[m * m];
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
        // (B.T V_xx)[:, c], then the whole of it from the quad
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double s = 0.0; bool first = true;
#pragma unroll
            for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + j), Bm[r * m + j], Vc[r]);
            BtVc[j] = s;
        }
#pragma unroll
        for (int j = 0; j < m; ++j)
            static_for<0, n>([&](auto R) { constexpr int r = decltype(R)::value; BtV[j * n + r] = bcast_quad<r>(BtVc[j]); });
        // :101  Q_ux = (B.T V_xx) A (whole, and this lane's column from its own column of A); :102  Q_uu
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double bt[n];
#pragma unroll
            for (int r = 0; r < n; ++r) bt[r] = BtV[j * n + r];
            Quxc[j] = dot_dense<n>(acol, bt);
#pragma unroll
            for (int q = 0; q < n; ++q) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + q), A[r * n + q], BtV[j * n + r]);
                Qux[j * n + q] = s;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double s = 0.0; bool first = true;
#pragma unroll
                for (int r = 0; r < n; ++r) acc_term(s, first, S::b_kind(r * m + l), Bm[r * m + l], BtV[j * n + r]);
                Quu[j * m + l] = Rk[j * m + l] + s;
            }
            Quu[j * m + j] += a.quu_reg;
        }
        // :103  Q_xx[:, c] = Q_k[:, c] + ((A.T V_xx) A)[:, c]; row i of A.T V_xx gathered from the quad
        double Qxxc[n];
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = 0.0; bool first = true;
#pragma unroll
            for (int r = 0; r < n; ++r) acc_term(s, first, S::a_kind(r * n + i), A[r * n + i], Vc[r]);
            double row[n];
            static_for<0, n>([&](auto R) { constexpr int r = decltype(R)::value; row[r] = bcast_quad<r>(s); });
            Qxxc[i] = Qkc[i] + dot_dense<n>(acol, row);
        }
        refill();                                                                  // the step's inputs are consumed
        // :106-108  gains
        double k[m], K[m * n], Kc[m];
        ok = Riccati<S>::inv_spd(Quu, Quu_inv) && ok;
#pragma unroll
        for (int j = 0; j < m; ++j) {
            double s = 0.0, tc = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) s = __builtin_fma(-Quu_inv[j * m + l], Qu[l], s);
            k[j] = s;
#pragma unroll
            for (int l = 0; l < m; ++l) tc = __builtin_fma(-Quu_inv[j * m + l], Quxc[l], tc);
            Kc[j] = tc;
#pragma unroll
            for (int q = 0; q < n; ++q) {
                double t2 = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) t2 = __builtin_fma(-Quu_inv[j * m + l], Qux[l * n + q], t2);
                K[j * n + q] = t2;
            }
        }
        if (act && owner) {
#pragma unroll
            for (int j = 0; j < m; ++j) Kp[((size_t)t * (m * n) + j * n + c) * TILE] = Kc[j];   // :112
            if (c == 0) store_col<m>(kp + (size_t)t * m * TILE, k);                              // :111
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
        const double cur = g + h;                                                  // :117
        ecr_g += g; ecr_h += h; ecr += cur;                                        // :119-121
        // :124-125  value function update (four-term forms, no symmetrisation)
        double KtQuu[n * m], KtQuuc[m];
#pragma unroll
        for (int l = 0; l < m; ++l) {
            double sc = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) sc = __builtin_fma(Kc[j], Quu[j * m + l], sc);
            KtQuuc[l] = sc;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < m; ++j) s = __builtin_fma(K[j * n + i], Quu[j * m + l], s);
                KtQuu[i * m + l] = s;
            }
        }
        {
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) v1 = __builtin_fma(KtQuuc[l], k[l], v1);
#pragma unroll
            for (int j = 0; j < m; ++j) v2 = __builtin_fma(Kc[j], Qu[j], v2);
#pragma unroll
            for (int j = 0; j < m; ++j) v3 = __builtin_fma(Quxc[j], k[j], v3);
            const double Vxc = ((Qxc + v1) + v2) + v3;
            static_for<0, n>([&](auto R) { constexpr int r = decltype(R)::value; Vx[r] = bcast_quad<r>(Vxc); });
        }
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) t1 = __builtin_fma(Qux[j * n + i], Kc[j], t1);
#pragma unroll
            for (int j = 0; j < m; ++j) t2 = __builtin_fma(K[j * n + i], Quxc[j], t2);
#pragma unroll
            for (int l = 0; l < m; ++l) t3 = __builtin_fma(KtQuu[i * m + l], Kc[l], t3);
            Vc[i] = ((Qxxc[i] + t1) + t2) + t3;
        }
    };
    // the first N mod D steps one by one, then whole groups of D steps from a ring of D slots (the group loop has no
    // exit inside its unrolled body: an early exit there costs a register shuffle per step)
    int t0 = a.t_hi - 1;
    for (const int rem = (a.t_hi - a.t_lo) % D; t0 >= a.t_hi - rem; --t0) {
        double A1[n * n], B1[n * m], u1[m], Ac1[n];
        fetch(t0, A1, B1, u1, Ac1);
        one_step(t0, A1, B1, u1, Ac1, [] {});
    }
    double Ar[D][n * n], Br[D][n * m], ur[D][m], Ac[D][n];
    if (t0 >= a.t_lo) {
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(t0 - d, Ar[d], Br[d], ur[d], Ac[d]);
    }
    for (; t0 >= a.t_lo; t0 -= D) {                                                    // :82
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 - d;
            one_step(t, Ar[d], Br[d], ur[d], Ac[d], [&] { fetch(t - D > a.t_lo ? t - D : a.t_lo, Ar[d], Br[d], ur[d], Ac[d]); });
        }
    }
    if (!last_chunk) {                                                                 // hand over to the chunk below
        if (owner) {
#pragma unroll
            for (int r = 0; r < n; ++r) vst[(size_t)(r * n + c) * TILE] = Vc[r];
        }
        if ((lane & 3) == 0) {
#pragma unroll
            for (int r = 0; r < n; ++r) vst[(size_t)(n * n + r) * TILE] = Vx[r];
            vst[(size_t)(n * n + n) * TILE] = ecr; vst[(size_t)(n * n + n + 1) * TILE] = ecr_g; vst[(size_t)(n * n + n + 2) * TILE] = ecr_h;
            vst[(size_t)(n * n + n + 3) * TILE] = ok ? 1.0 : 0.0;
        }
        return;
    }
    if (act && (lane & 3) == 0) backward_epilogue(a, tile, bl, b, ecr, ecr_g, ecr_h, ok);
}
template <class S, bool DEEP = false>
__global__ void __launch_bounds__(64) k_riccati_quad(const KArgs a)
{
    riccati_quad_body<S, DEEP>(a, blockIdx.x);
}

// LTV = true: the ltvlqr recursion (ltvlqr.py:14-27) on the same lane mapping -- no V_x, k or expected reductions, the
// value update is P = (Q + A'PA) - (A'PB) G^-1 (B'PA) with A'PB formed from the rows of A'P as the reference does.
template <class S, bool LTV = false>
__global__ void __launch_bounds__(64, Grp<S>::BIGM ? 1 : RICCATI_GRP_WAVES) k_riccati_grp(const KArgs a)
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
    const bool act = LTV || a.active[b] != 0;
    if (__ballot(act) == 0ull) return;
    double* s = sh + grp * G::SLOT;
    const bool row = c < n;
    const int cc = row ? c : n - 1;                     // surplus lanes shadow the last row / column and never store
    // where row cc of A_t / B_t sits in a LIN record: byte offsets of the general entries (structural zeros and ones
    // load a harmless valid entry instead; their constants are written to the transpose buffer once, below)
    int aoff[n], boff[m];
    unsigned amask = 0, bmask = 0, adt = 0, bdt = 0;   // adt / bdt: entries of this lane's row that are exactly dt (not stored)
#pragma unroll
    for (int i = 0; i < n; ++i) {
        const int q = lin_index<S, false>(cc * n + i);
        aoff[i] = (q > 0 ? q : 0) * (TILE * (int)sizeof(double));
        amask |= (q >= 0 ? 1u : 0u) << i;
        adt |= (q == -3 ? 1u : 0u) << i;
        if (row && q < 0) s[G::oAT + i * LD + c] = q == -2 ? 1.0 : (q == -3 ? a.dt : 0.0);
    }
#pragma unroll
    for (int l = 0; l < m; ++l) {
        const int q = lin_index<S, true>(cc * m + l);
        boff[l] = (q > 0 ? q : 0) * (TILE * (int)sizeof(double));
        bdt |= (q == -3 ? 1u : 0u) << l;
        bmask |= (q >= 0 ? 1u : 0u) << l;
        if constexpr (G::BIGM) { if (row && q < 0) s[G::oBT + l * LD + c] = q == -2 ? 1.0 : (q == -3 ? a.dt : 0.0); }
    }
    // columns of A / B that hold a dt entry in some row (compile time): only those pay the per-lane patch below
    constexpr auto a_col_dt = [](int i) { bool v = false; for (int r = 0; r < n; ++r) v = v || S::a_is_dt(r * n + i); return v; };
    constexpr auto b_col_dt = [](int l) { bool v = false; for (int r = 0; r < n; ++r) v = v || S::b_is_dt(r * m + l); return v; };
    double Rk[G::BIGM ? 1 : m * m], Vc[n], vx;
    double* sQk = sh + P * G::SLOT;                                                    // Q_k, shared by the rows
    for (int e = lane; e < n * n; e += TILE) sQk[e] = a.WTS[e];
    const int cm = c < m ? c : m - 1;                   // BIGM: the column of Q_uu (and of B_t, R_k) this lane owns
    double Rcol[G::BIGM ? m : 1], Rrow[G::BIGM ? m : 1];
    if constexpr (G::BIGM) {
#pragma unroll
        for (int l = 0; l < m; ++l) { Rcol[l] = a.WTS[n * n + l * m + cm]; Rrow[l] = a.WTS[n * n + cm * m + l]; }
    } else {
        load_uniform<m * m>(a.WTS + n * n, Rk);
    }
    const double* QT = a.WTS + n * n + m * m;
#pragma unroll
    for (int r = 0; r < n; ++r) Vc[r] = QT[r * n + cc];                                // :67 column j of Q_T
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + pl;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + pl;
    const double* Lp = a.LIN + (size_t)tile * a.N * NL * TILE + pl;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + pl;
    double* kp = a.KFF + (size_t)tile * a.N * m * TILE + pl;
    if constexpr (LTV) {
        vx = 0.0;
    } else {
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
            if (a_col_dt(i)) Ar[i] = ((adt >> i) & 1u) ? a.dt : Ar[i];
        }
#pragma unroll
        for (int q = 0; q < m; ++q) {
            int o = boff[q];
            asm volatile("" : "+v"(o));
            Br[q] = *(const double*)(l + o);
            if (b_col_dt(q)) Br[q] = ((bdt >> q) & 1u) ? a.dt : Br[q];
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
            fmac_row_pattern<S, false, r>(AtV, Ar, Vc[r]);                             // general entries: A[r][i] * V[r][j]
            static_for<0, n>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (S::A_pat[r * n + i] == '1') AtV[i] = AtV[i] + Vc[r];
            });
            fmac_row_pattern<S, true, r>(BtV, Br, Vc[r]);                              // B[r][l] * V[r][j]
            static_for<0, m>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                if constexpr (S::B_pat[r * m + l] == '1') BtV[l] = BtV[l] + Vc[r];
            });
        });
        if (row) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
                s[G::oT + i * LD + c] = AtV[i];
                if ((amask >> i) & 1u) s[G::oAT + i * LD + c] = Ar[i];
            }
            if constexpr (G::BIGM) {
#pragma unroll
                for (int l = 0; l < m; ++l) if ((bmask >> l) & 1u) s[G::oBT + l * LD + c] = Br[l];
            }
        }
        __syncthreads();
        // ---- gains (every lane: Q_uu, Q_u, the inverse, k; lane i: column i of Q_ux and K, Q_x[i]).  Done before the
        // row phase so that Q_uu and its inverse (2 m^2 doubles) are dead when the rows of A^T V_xx come into registers.
        double Qux[m], Qu[m], k[m], Kc[m], KtQuu[m], g = 0.0, h = 0.0;
        double Gi[LTV && !G::BIGM ? m * m : 1];                                        // ltvlqr keeps G^-1 for the P update
        double wl[LTV ? m : 1];                                                        // ltvlqr: (A'PB G^-1)[i][:]
        if constexpr (G::BIGM) {
            // ---- m > 4.  Lane l (< m) owns column l of Q_uu; lane i (< n) owns column i of Q_ux and solves for column i of
            // K; lane m, a spare one, solves for k.  Q_uu = L D L' is factorised right-looking: at step j lane j's pivot and
            // its column come to every lane by row broadcasts (the whole of L ends up replicated, 66 doubles for m = 12),
            // every lane updates its own column and -- forward substitution folded in -- its own right-hand side; then
            // D^-1 and the backward substitution with L'.  The reference multiplies by inv(Q_uu) (ilqr.py:106-108); this
            // solves with the factors: the same gains up to rounding, as the Cholesky-based inverse for m <= 4.
            // np.linalg.inv's rule is kept as far as a factorisation without pivoting can: only an exactly zero pivot is an
            // error (status 3).  An indefinite Q_uu is solved all the same (D then has negative entries; as accurate as its
            // leading minors are conditioned), and a NaN Q_uu yields NaN gains, so that a diverged problem ends as it does in
            // the reference -- every Armijo test False, 'low learning rate'.
            double Acol[n], Bcol[n], qc[m], qo[m], Qx = 0.0, quc = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) { Acol[r] = s[G::oAT + cc * LD + r]; Bcol[r] = s[G::oBT + cm * LD + r]; }
#pragma unroll
            for (int l = 0; l < m; ++l) { Qux[l] = 0.0; qc[l] = 0.0; }
            static_for<0, n>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                fmac_row_vec<r, m>(Qux, BtV, Acol[r]);                                 // :101 (B^T V_xx)[l][r] * A[r][i]
                fmac_row_vec<r, m>(qc, BtV, Bcol[r]);                                  // :102 (B^T V_xx)[l][r] * B[r][c]
                fmac_row<r>(Qx, vx, Acol[r]);                                          // :99
                fmac_row<r>(quc, vx, Bcol[r]);                                         // :100 B[r][c] * V_x[r]
            });
            {
                double lu = 0.0;                                                       // :93 l_u = R_k u, entry c
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) lu = __builtin_fma(Rrow[l2], u[l2], lu);
                quc = lu + quc;
            }
#pragma unroll
            for (int l = 0; l < m; ++l) {
                qc[l] = Rcol[l] + qc[l];
                if (l == cm) qc[l] += a.quu_reg;
                qo[l] = qc[l];                                                         // Q_uu[:, c] as it is, for :116 and :124-125
            }
            static_for<0, m>([&](auto lc) { constexpr int l = decltype(lc)::value; Qu[l] = bcast_row<l>(quc); });
            double b1[m], b2[LTV ? m : 1], rinv[m], Lr[m * m];
#pragma unroll
            for (int l = 0; l < m; ++l) b1[l] = c == m ? Qu[l] : Qux[l];
            if constexpr (LTV) {                                                       // second right-hand side: (A'PB)[i][:]
                double Ri0[n];
#pragma unroll
                for (int r = 0; r < n; ++r) Ri0[r] = s[G::oT + cc * LD + r];
#pragma unroll
                for (int l = 0; l < m; ++l) b2[l] = 0.0;
                static_for<0, n>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    fmac_row_pattern<S, true, r>(b2, Br, Ri0[r]);
                    static_for<0, m>([&](auto lc) {
                        constexpr int l = decltype(lc)::value;
                        if constexpr (S::B_pat[r * m + l] == '1') b2[l] = b2[l] + Ri0[r];
                    });
                });
            }
            static_for<0, m>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double d = bcast_row<j>(qc[j]);                                  // pivot D[j]
                ok = ok && d != 0.0;
                const double ri = 1.0 / d;
                rinv[j] = ri;
                double lcj = 0.0;                                                      // L[c][j] for this lane's column c > j
                static_for<j + 1, m>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const double lij = bcast_row<j>(qc[i]) * ri;                       // L[i][j], the same in every lane
                    Lr[i * m + j] = lij;
                    lcj = i == cm ? lij : lcj;
                    b1[i] = __builtin_fma(-lij, b1[j], b1[i]);                          // forward substitution, column-wise
                    if constexpr (LTV) b2[i] = __builtin_fma(-lij, b2[j], b2[i]);
                });
                const double w = lcj * d;
                static_for<j + 1, m>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    qc[i] = __builtin_fma(-Lr[i * m + j], w, qc[i]);                    // Q[i][c] -= L[i][j] D[j] L[c][j]
                });
            });
            static_for<0, m>([&](auto lc) {                                            // D^-1, then L' x = z from the last row up
                constexpr int l = m - 1 - decltype(lc)::value;
                double x1 = b1[l] * rinv[l], x2 = 0.0;
                if constexpr (LTV) x2 = b2[l] * rinv[l];
                static_for<l + 1, m>([&](auto qcx) {
                    constexpr int q = decltype(qcx)::value;
                    x1 = __builtin_fma(-Lr[q * m + l], b1[q], x1);
                    if constexpr (LTV) x2 = __builtin_fma(-Lr[q * m + l], b2[q], x2);
                });
                b1[l] = x1;
                if constexpr (LTV) b2[l] = x2;
            });
            static_for<0, m>([&](auto lc) {                                            // :107-108
                constexpr int l = decltype(lc)::value;
                Kc[l] = -b1[l];
                k[l] = bcast_row<m>(Kc[l]);                                            // lane m solved for Q_u
                if constexpr (LTV) wl[l] = b2[l];
            });
#pragma unroll
            for (int l = 0; l < m; ++l) g = __builtin_fma(-Qu[l], k[l], g);            // :115
            {
                double sc = 0.0;                                                       // :116  (0.5 k' Q_uu)[c], then . k
#pragma unroll
                for (int l2 = 0; l2 < m; ++l2) sc = __builtin_fma(0.5 * k[l2], qo[l2], sc);
                static_for<0, m>([&](auto lc) { constexpr int l = decltype(lc)::value; h = __builtin_fma(bcast_row<l>(sc), k[l], h); });
            }
#pragma unroll
            for (int l = 0; l < m; ++l) KtQuu[l] = 0.0;
            static_for<0, m>([&](auto lc) {                                            // (K' Q_uu)[i][l] = sum_l2 K[l2][i] Q_uu[l2][l]
                constexpr int l = decltype(lc)::value;
                fmac_row_sum<l, m>(KtQuu[l], qo, Kc);
            });
            double v1 = 0.0, v2 = 0.0, v3 = 0.0;                                       // :124
#pragma unroll
            for (int l = 0; l < m; ++l) v1 = __builtin_fma(KtQuu[l], k[l], v1);
#pragma unroll
            for (int l = 0; l < m; ++l) v2 = __builtin_fma(Kc[l], Qu[l], v2);
#pragma unroll
            for (int l = 0; l < m; ++l) v3 = __builtin_fma(Qux[l], k[l], v3);
            vx = ((Qx + v1) + v2) + v3;
        } else {
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
                        fmac_row_vec<r, m, m, l2>(Quu, BtV, Brl);                      // :102  Quu[l][l2] += (B^T V_xx)[l][r] * B[r][l2]
                        fmac_row<r>(Qu[l2], vx, Brl);                                  // :100 B[r][l2] * V_x[r]
                    } else if constexpr (kd == '1') {
#pragma unroll
                        for (int l = 0; l < m; ++l) Quu[l * m + l2] = Quu[l * m + l2] + bcast_row<r>(BtV[l]);
                        Qu[l2] = Qu[l2] + bcast_row<r>(vx);
                    }
                });
                fmac_row_vec<r, m>(Qux, BtV, Acol[r]);                                 // :101 (B^T V_xx)[l][r] * A[r][i]
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
            if constexpr (LTV) {
#pragma unroll
                for (int q = 0; q < m * m; ++q) Gi[q] = Quu_inv[q];
            }
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
            if constexpr (!LTV) {
#pragma unroll
                for (int l = 0; l < m; ++l) if (c == l) kp[((size_t)t * m + l) * TILE] = k[l];      // :111
            }
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
        if constexpr (LTV && !G::BIGM) {
            double apb[m];
#pragma unroll
            for (int l = 0; l < m; ++l) apb[l] = 0.0;
            static_for<0, n>([&](auto rc) {                                            // (A'PB)[i][l] = sum_r (A'P)[i][r] B[r][l]
                constexpr int r = decltype(rc)::value;
                fmac_row_pattern<S, true, r>(apb, Br, Ri[r]);
                static_for<0, m>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    if constexpr (S::B_pat[r * m + l] == '1') apb[l] = apb[l] + Ri[r];
                });
            });
#pragma unroll
            for (int l = 0; l < m; ++l) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < m; ++q) acc = __builtin_fma(apb[q], Gi[q * m + l], acc);
                wl[l] = acc;
            }
        }
        static_for<0, n>([&](auto jc) {                                                // registers ended with the column phase)
            constexpr int j = decltype(jc)::value;
            double acc = 0.0;
            fmac_col_pattern<S, j>(acc, Aw[j], Ri);                                    // sum over r of (A^T V_xx)[i][r] * A[r][j]
            const double Qxx = sQk[cc * n + j] + acc;
            if constexpr (LTV) {                                                       // ltvlqr.py:27
                double corr = 0.0;
                fmac_row_sum<j, m>(corr, Qux, wl);                                     // w[i][l] * (B'PA)[l][j]
                const double Pij = Qxx - corr;
                if (row) s[G::oT + c * LD + j] = Pij;
                return;
            }
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
            fmac_row_sum<j, m>(t1, Kc, Qux);                                           // Q_ux[l][i] * K[l][j]
            fmac_row_sum<j, m>(t2, Qux, Kc);                                           // K[l][i] * Q_ux[l][j]
            fmac_row_sum<j, m>(t3, Kc, KtQuu);                                         // (K^T Q_uu)[i][l] * K[l][j]
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
    if constexpr (LTV) {
        if (!ok && c == 0) a.status[b] = ILQR_STATUS_NUMERICAL;
    } else {
        if (act && c == 0) backward_epilogue(a, tile, pl, b, ecr, ecr_g, ecr_h, ok);
    }
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
    // The trajectory is walked row by row: the sin/cos pairs of its angle states are carried from row t + 1 to row t by
    // angle addition on the difference of the two rows (TrigCarry: evaluated afresh at the last row, at every multiple of
    // the checkpoint interval and whenever two rows are further apart than 3/32 rad), instead of two range reductions and
    // polynomial pairs per row.  A_t and B_t are I + dt * (...): a trig value's last bits reach them scaled by dt.
    TrigCarry<S> trig;
    double xnext[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xnext[i] = 0.0;
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
        if constexpr (S::n_ang > 0) {
            trig.advance(xnext, x, t == a.N - 1 || (t + 1) % a.ck == 0);
#pragma unroll
            for (int i = 0; i < n; ++i) xnext[i] = x[i];
            S::A_sc(x, u, a.dt, p, trig.sc, A);                                        // :21
            S::B_sc(x, u, a.dt, p, trig.sc, Bm);                                       // :22
        } else {
            S::A(x, u, a.dt, p, A);                                                    // :21
            S::B(x, u, a.dt, p, Bm);                                                   // :22
        }
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

