// kernels_forward.h -- forward side: k_rollout, k_cost, the closed-loop rollouts, k_linesearch, k_forward, k_accept_par (ilqr.py:30-62, 137-207)
// Part of ilqr_hip.hip (included inside its anonymous namespace, in this order: kernels_common.h,
// kernels_backward.h, kernels_forward.h, kernels_util.h); not a stand-alone header.
// ------------------------------------------------------------------------------------------------
// k_rollout: ilqr.py:30-47 with compute_cost (:49-62) fused.  One lane per problem, one wave per tile.
// ------------------------------------------------------------------------------------------------
#ifndef ROLLOUT_PREFETCH
#define ROLLOUT_PREFETCH 8
#endif
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
    TrigCarry<S> trig;
    trig.seed(x);
    int next_seed = a.ck;
    auto one_step = [&](const int t, const double (&u)[m]) {
        double xn[n];
        cost = cost + stage_input_cost<m>(a, u, Rk);
        if constexpr (S::n_ang > 0) {
            S::f_sc(x, u, a.dt, p, trig.sc, xn);
            const bool reseed = t + 1 == next_seed;
            if (reseed) next_seed += a.ck;
            trig.advance(x, xn, reseed);
        } else {
            S::f(x, u, a.dt, p, xn);
        }
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        store_col<n>(Xp + (size_t)(t + 1) * n * TILE, x);                              // :39
        store_col<m>(Up + (size_t)t * m * TILE, u);                                    // :40
    };
    // the guess of the next D steps is in flight (one step of f is shorter than an HBM round trip): the first N mod D
    // steps one by one, then whole groups of D from a ring
    constexpr int D = ROLLOUT_PREFETCH;
    int t0 = 0;
    for (const int rem = a.N % D; t0 < rem; ++t0) {
        double u[m];
        load_col<m>(Gp + (size_t)t0 * m * TILE, u);
        one_step(t0, u);
    }
    double ur[D][m];
    if (t0 < a.N) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_col<m>(Gp + (size_t)(t0 + d) * m * TILE, ur[d]);
    }
    for (; t0 < a.N; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 + d;
            double u[m];
#pragma unroll
            for (int j = 0; j < m; ++j) u[j] = ur[d][j];
            const int tl = t + D < a.N ? t + D : a.N - 1;
            load_col<m>(Gp + (size_t)tl * m * TILE, ur[d]);
            one_step(t, u);
        }
    }
    cost = cost + terminal_cost<n>(x, target, QT);
    a.cost[b] = cost;
}

// k_cost: compute_cost (ilqr.py:49-62) of the stored trajectory.
// TRACK: the tracking cost of matlab/smooth-ilqr/ilqr_mpc.m:193-216 (per-step targets XD, UD; stage cost over rows 0..N-1
// of the states, terminal cost against the last row of XD).
template <class S, bool TRACK = false>
__global__ void __launch_bounds__(64) k_cost(const KArgs a)
{
    constexpr int n = S::n, m = S::m;
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    double x[n], target[n], Rk[m * m], QT[n * n];
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    double cost = 0.0;
    if constexpr (TRACK) {
        double Qk[n * n];
        load_uniform<n * n>(a.WTS, Qk);
        const double* XDp = a.XD + (size_t)tile * a.state_rows * n * TILE + lane;
        const double* UDp = a.UD + (size_t)tile * a.N * m * TILE + lane;
        for (int t = 0; t < a.N; ++t) {
            double u[m], xd[n], ud[m];
            load_col<m>(Up + (size_t)t * m * TILE, u);
            load_col<n>(Xp + (size_t)t * n * TILE, x);
            load_col<n>(XDp + (size_t)t * n * TILE, xd);
            load_col<m>(UDp + (size_t)t * m * TILE, ud);
            cost = cost + tracking_stage_cost<n, m>(x, u, xd, ud, Qk, Rk);
        }
        load_col<n>(XDp + (size_t)a.N * n * TILE, target);
    } else {
        load_col<n>(a.TARGET + (size_t)tile * n * TILE + lane, target);
        for (int t = 0; t < a.N; ++t) {
            double u[m];
            load_col<m>(Up + (size_t)t * m * TILE, u);
            cost = cost + stage_input_cost<m>(a, u, Rk);
        }
    }
    load_col<n>(Xp + (size_t)a.N * n * TILE, x);
    cost = cost + terminal_cost<n>(x, target, QT);
    a.cost[tile * TILE + lane] = cost;
}

// terminal condition of backwards_pass (ilqr.py:67-70): V_x = Q_T (x_N - target); Vxx already holds Q_T
// ------------------------------------------------------------------------------------------------
// closed-loop rollout of one lane over steps [t0, t1) (ilqr.py:144-157), accumulating the input cost of
// compute_cost; optionally storing the new trajectory (STORE) and the state at every segment boundary (CKPT).
// ------------------------------------------------------------------------------------------------
// 1: k_accept_lin also evaluates A_t, B_t (one pass over the trajectory); 0: k_linearize does it afterwards.  Measured on
// the headline config: fused 0.81 ms (200 B per problem and step), separate 0.32 + 0.53 ms (240 B): either way the pair
// moves ~3.4 TB/s of mixed reads and writes; the old X2/U2 copy-back schedule took 0.39 + 0.70 ms.
#ifndef ACCEPT_FUSES_LINEARIZE
#define ACCEPT_FUSES_LINEARIZE 1
#endif
#ifndef ACCEPT_PREFETCH
#define ACCEPT_PREFETCH 1
#endif
#ifndef CLOSED_LOOP_PREFETCH
#define CLOSED_LOOP_PREFETCH 3
#endif
#ifndef CLOSED_LOOP_SCHED_BARRIER
#define CLOSED_LOOP_SCHED_BARRIER 1
#endif
template <class S, bool STORE, bool CKPT, bool LINOUT = false, bool TRACK = false>
__device__ __forceinline__ double closed_loop_range(const KArgs& a, const double* Xp, const double* Up, const double* Kp,
                                                    const double* kp, double* XOp, double* UOp, double* ckp,
                                                    double (&x)[S::n], const double (&p)[S::np],
                                                    const double (&Rk)[S::m * S::m], const double alpha, const int t0,
                                                    const int t1, const double* Xfirst = nullptr, double* Lp = nullptr,
                                                    const double kill_at = __builtin_inf(), const double* XDp = nullptr,
                                                    const double* UDp = nullptr)
{
    // TRACK: the stage cost is the tracking cost of ilqr_mpc.m:207-209 against the per-step targets XDp, UDp (read one
    // step ahead; these rollouts keep the plain one-step prefetch of the records, the ring is for the headline path)
    // kill_at (line-search trials only, STORE = false): once the running cost exceeds it the trial is known to fail the
    // Armijo test (k_linesearch sets it where that is provable) and the lane's remaining steps are don't-cares.
    // Xfirst: where the reference-state row of step t0 is kept when X itself is being overwritten by a neighbour
    // (k_accept_lin).  LINOUT: also evaluate A_t, B_t at the new (x_t, u_t) and store them for the next backward pass.
    constexpr int n = S::n, m = S::m;
    // Rows of the reference trajectory and gains for the next D steps are in flight while step t computes -- in a ring
    // of registers when the per-step record is small (n <= 4: 10 doubles per slot), else (quadcopter: 68 doubles) only
    // the trajectory row is one step ahead and the gains are left to the scheduler.  A slot is refilled as soon as
    // its step's input is formed, i.e. row t+D of X is read before the STORE of row t+1 (in-place accept, D >= 1).
    constexpr bool PREFETCH = (n + 2 * m + m * n) <= 16;
    constexpr int D = PREFETCH ? (LINOUT || TRACK ? ACCEPT_PREFETCH : CLOSED_LOOP_PREFETCH) : 1;   // the accept streams: deeper
    double Qk[TRACK ? n * n : 1], xdn[TRACK ? n : 1], udn[TRACK ? m : 1];
    if constexpr (TRACK) {
        load_uniform<n * n>(a.WTS, Qk);
        load_col<n>(XDp + (size_t)t0 * n * TILE, xdn);
        load_col<m>(UDp + (size_t)t0 * m * TILE, udn);
    }
    double cost = 0.0;
    double xb[D][n], ub[D][m], K[D][m * n], kf[D][m];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int tl = t0 + d < t1 ? t0 + d : t1 - 1;
        load_col<n>(Xfirst != nullptr && tl == t0 ? Xfirst : Xp + (size_t)tl * n * TILE, xb[d]);
        load_col<m>(Up + (size_t)tl * m * TILE, ub[d]);
        if constexpr (PREFETCH) { load_col<m * n>(Kp + (size_t)tl * (m * n) * TILE, K[d]); load_col<m>(kp + (size_t)tl * m * TILE, kf[d]); }
    }
    int next_ck = CKPT ? a.ck : 0;
    // sin/cos pairs of the angle states, carried from step to step and evaluated afresh at every multiple of the
    // checkpoint interval (t0 is one when this range resumes from a checkpoint): see TrigCarry
    TrigCarry<S> trig;
    trig.seed(x);
    int next_seed = (t0 / a.ck + 1) * a.ck;
    auto one_step = [&](const int t, const int d, const bool refill) __attribute__((always_inline)) {
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
        if constexpr (LINOUT) {                                                    // :96-97 for the next backward pass
            double xl[n], A[n * n], Bm[n * m];
#pragma unroll
            for (int i = 0; i < n; ++i) xl[i] = t == 0 ? xb[d][i] : x[i];          // row 0 as stored (quirk Q1), else x_t
            S::A(xl, u, a.dt, p, A);
            S::B(xl, u, a.dt, p, Bm);
            store_lin<S>(Lp + (size_t)t * Lin<S>::n_lin * TILE, A, Bm);
        }
        {
#if CLOSED_LOOP_SCHED_BARRIER
            // refill AFTER the slot's last use (u pinned in place: the compiler otherwise sinks its computation below
            // the loads and pays one register copy per loaded value and step): same registers, no copies
            if constexpr (PREFETCH && !LINOUT) {
#pragma unroll
                for (int j = 0; j < m; ++j) asm volatile("" : "+v"(u[j]));
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if (refill) {
                const int tl = t + D < t1 ? t + D : t1 - 1;                        // slot d next holds step t + D
                load_col<n>(Xp + (size_t)tl * n * TILE, xb[d]); load_col<m>(Up + (size_t)tl * m * TILE, ub[d]);
                if constexpr (PREFETCH) { load_col<m * n>(Kp + (size_t)tl * (m * n) * TILE, K[d]); load_col<m>(kp + (size_t)tl * m * TILE, kf[d]); }
            }
        }
        if constexpr (TRACK) {
            double xd[n], ud[m];
#pragma unroll
            for (int i = 0; i < n; ++i) xd[i] = xdn[i];
#pragma unroll
            for (int j = 0; j < m; ++j) ud[j] = udn[j];
            const int tn = t + 1 < t1 ? t + 1 : t1 - 1;
            load_col<n>(XDp + (size_t)tn * n * TILE, xdn);
            load_col<m>(UDp + (size_t)tn * m * TILE, udn);
            cost = cost + tracking_stage_cost<n, m>(x, u, xd, ud, Qk, Rk);         // ilqr_mpc.m:207-210
        } else {
            cost = cost + stage_input_cost<m>(a, u, Rk);
        }
        if constexpr (S::n_ang > 0) {
            S::f_sc(x, u, a.dt, p, trig.sc, xn);                                   // :151
            const bool reseed = t + 1 == next_seed;
            if (reseed) next_seed += a.ck;
            trig.advance(x, xn, reseed, !STORE && cost > kill_at);
        } else {
            S::f(x, u, a.dt, p, xn);                                               // :151
        }
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        if constexpr (STORE) {
            store_col<m>(UOp + (size_t)t * m * TILE, u);                           // :154
            store_col<n>(XOp + (size_t)(t + 1) * n * TILE, x);                     // :153
        }
    };
    // whole groups of D steps with no exit inside the unrolled group (an early exit there costs a register shuffle per
    // step), then the last (t1 - t0) mod D steps from the slots the clamped refills left in place
    int tb = t0;
    for (; tb + D <= t1; tb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) one_step(tb + d, d, true);
    }
#pragma unroll
    for (int d = 0; d + 1 < D; ++d)
        if (tb + d < t1) one_step(tb + d, d, false);
    return cost;
}

template <class S, bool STORE, bool CKPT = false, bool TRACK = false>
__device__ __forceinline__ double closed_loop(const KArgs& a, const double* Xp, const double* Up, const double* Kp,
                                              const double* kp, double* XOp, double* UOp, double (&x)[S::n],
                                              const double (&p)[S::np], const double (&target)[S::n],
                                              const double (&Rk)[S::m * S::m], const double (&QT)[S::n * S::n],
                                              const double alpha, double* ckp = nullptr, const double kill_at = __builtin_inf(),
                                              const double* XDp = nullptr, const double* UDp = nullptr)
{
    double cost = closed_loop_range<S, STORE, CKPT, false, TRACK>(a, Xp, Up, Kp, kp, XOp, UOp, ckp, x, p, Rk, alpha, 0, a.N, nullptr,
                                                                  nullptr, kill_at, XDp, UDp);
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
    // smaller records are prefetched in registers (or gathered, below).  Narrow rounds (W = 2, 4: 32 or 16 records per wave,
    // 35 / 18 KB of LDS) run one wave per workgroup; beyond 4 inputs only the widths from 16 up are compiled (launch_ls)
    static constexpr bool use = (n + 2 * m + m * n) > 16 && W >= 2 && (m <= 4 || W >= 16);
    static constexpr int WAVES_PER_BLOCK = W == 64 ? 8 : (W >= 8 ? 4 : 1);
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
    TrigCarry<S> trig;                                                                 // as in closed_loop_range: same pairs, same bits
    trig.seed(x);
    int next_seed = a.ck;
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
        cost = cost + stage_input_cost<m>(a, u, Rk);
        if constexpr (S::n_ang > 0) {
            S::f_sc(x, u, a.dt, p, trig.sc, xn);                                       // :151
            const bool reseed = t + 1 == next_seed;
            if (reseed) next_seed += a.ck;
            trig.advance(x, xn, reseed);
        } else {
            S::f(x, u, a.dt, p, xn);                                                   // :151
        }
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        __builtin_amdgcn_wave_barrier();                                               // reads of this parity's other buffer are done
        stash(t + 1);
        __builtin_amdgcn_wave_barrier();
    }
    return cost;
}

// ------------------------------------------------------------------------------------------------
// The same idea for SMALL records (n <= 4: 6 to 16 doubles) at the narrow widths the large batches start with (W = 4, 8,
// 16: 16, 8, 4 problems per wave).  There the register path issues one load per component and step whose 64 lanes hold
// only 64 / W distinct doubles, and rocprofv3 (profiles/r3_*) showed all three large-batch configurations pinned at the same
// 2.3e10 such instructions per second -- 2.95 TB/s, VALU 35-58 % busy, whatever the prefetch depth or occupancy
// (tools/probe/ls_load_probe.hip: that pattern caps at 3.1 TB/s, the same lines fetched by fully distinct lanes run at 6.3).
// Here lane l fetches component l/G (+W, +2W) of problem l%G -- ceil(REC / W) loads per step instead of REC, every lane a
// different double -- D steps ahead into a small register ring (3 doubles per slot at W = 4 instead of 10), hands the
// record over through LDS two steps ahead of its use (double buffered) and reads its problem's record back one step ahead.
// The rows a lane reads beyond step N - 1 exist (LS_GATHER_PAD_STEPS rows of padding behind every trajectory array) and
// are never consumed.  Same arithmetic in the same order as closed_loop_range: bit-identical trials.
// ------------------------------------------------------------------------------------------------
#ifndef LS_GATHER
#define LS_GATHER 1
#endif
#ifndef LS_GATHER_DEPTH
#define LS_GATHER_DEPTH 4
#endif
#ifndef LS_GATHER_MAXW
#define LS_GATHER_MAXW 64
#endif
constexpr int LS_GATHER_PAD_STEPS = 16;      // >= LS_GATHER_DEPTH + 3: rows readable behind the last time step of X, U, KFB, KFF
template <class S, int W>
struct Gather {
    static constexpr int n = S::n, m = S::m, G = TILE / W;
    static constexpr int REC = n + 2 * m + m * n, RECP = (REC + 1) | 1;  // odd stride >= REC + 1: slot REC takes the idle lanes' stores
    static constexpr int oX = 0, oU = n, oK = n + m, ok = n + m + m * n;
    static constexpr int Q = (REC + W - 1) / W;                          // loads per lane and step
    static constexpr int D = LS_GATHER_DEPTH;                            // steps in flight from HBM
    static constexpr int BUF = G * RECP, WAVE_DOUBLES = 2 * BUF;
    static constexpr bool use = LS_GATHER && REC <= 16 && W >= 4 && W <= LS_GATHER_MAXW;
    static_assert(D % 2 == 0 && D + 3 <= LS_GATHER_PAD_STEPS, "ring depth");
};

template <class S, int W, bool CKPT>
__device__ __forceinline__ double closed_loop_gather(const KArgs& a, double* sw, const int lane, const int grp,
                                                     const long long b_duty, double* ckp, const bool valid,
                                                     double (&x)[S::n], const double (&p)[S::np],
                                                     const double (&Rk)[S::m * S::m], const double alpha,
                                                     const double kill_at)
{
    using T = Gather<S, W>;
    constexpr int n = S::n, m = S::m, G = T::G, RECP = T::RECP, Q = T::Q, D = T::D, REC = T::REC;
    const int gd = lane % G, cd = lane / G;                                            // load duty: problem gd, components cd + W q
    const long long tile_d = b_duty >> 6;
    const int pl_d = (int)(b_duty & 63);
    // per load q: where this lane's component lives, how far it moves per time step, and its slot in the LDS record
    const double* src[Q];
    long long stride[Q];
    int slot[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c0 = cd + W * q, c = c0 < REC ? c0 : REC - 1;
        slot[q] = gd * RECP + (c0 < REC ? c0 : REC);
        if (c < T::oU) {
            src[q] = a.X + (size_t)tile_d * a.state_rows * n * TILE + pl_d + (size_t)c * TILE; stride[q] = n * TILE;
        } else if (c < T::oK) {
            src[q] = a.U + (size_t)tile_d * a.N * m * TILE + pl_d + (size_t)(c - T::oU) * TILE; stride[q] = m * TILE;
        } else if (c < T::ok) {
            src[q] = a.KFB + (size_t)tile_d * a.N * (m * n) * TILE + pl_d + (size_t)(c - T::oK) * TILE; stride[q] = m * n * TILE;
        } else {
            src[q] = a.KFF + (size_t)tile_d * a.N * m * TILE + pl_d + (size_t)(c - T::ok) * TILE; stride[q] = m * TILE;
        }
    }
    double ring[D][Q];                                                                 // slot s holds a step congruent to s mod D
    auto fetch = [&](const int s) __attribute__((always_inline)) {                     // the next step of the stream into slot s
#pragma unroll
        for (int q = 0; q < Q; ++q) { ring[s][q] = *src[q]; src[q] += stride[q]; }
    };
    auto stash = [&](const int s, const int par) __attribute__((always_inline)) {      // slot s -> LDS buffer `par`
#pragma unroll
        for (int q = 0; q < Q; ++q) sw[par * T::BUF + slot[q]] = ring[s][q];
    };
    double rec[2][REC];
    auto readback = [&](const int par, double (&o)[REC]) __attribute__((always_inline)) {
        const double* r = sw + par * T::BUF + grp * RECP;
#pragma unroll
        for (int c = 0; c < REC; ++c) o[c] = r[c];
    };
    // prologue: steps 0 .. D+1 fetched; 0 and 1 in LDS; record 0 read back
#pragma unroll
    for (int s = 0; s < D; ++s) fetch(s);
    stash(0, 0); fetch(0);                                                             // slot 0 now holds step D
    stash(1, 1); fetch(1);                                                             // slot 1 now holds step D + 1
    __builtin_amdgcn_wave_barrier();
    readback(0, rec[0]);
    double cost = 0.0;
    int next_ck = CKPT ? a.ck : 0;
    TrigCarry<S> trig;                                                                 // as in closed_loop_range: same pairs, same bits
    trig.seed(x);
    int next_seed = a.ck;
    // step t: record t is in rec[t & 1]; LDS holds t + 1 (buffer (t + 1) & 1) and, once this step has stashed it, t + 2
    // (buffer t & 1, whose record t was read back during step t - 1); the ring holds t + 2 .. t + D + 1
    auto one_step = [&](const int t, const int e) __attribute__((always_inline)) {     // e = t mod D (compile time)
        const int par = e & 1;
        readback(par ^ 1, rec[par ^ 1]);                                               // record t + 1, for the next step
        __builtin_amdgcn_wave_barrier();
        stash((e + 2) % D, par);                                                       // record t + 2 over record t
        fetch((e + 2) % D);                                                            // step t + 2 + D
        __builtin_amdgcn_wave_barrier();
        const double (&r)[REC] = rec[par];
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
        double u[m], xn[n];
#pragma unroll
        for (int j = 0; j < m; ++j) {
            const double ff = alpha * r[T::ok + j];                                    // :146
            double fb = 0.0;                                                           // :147
#pragma unroll
            for (int i = 0; i < n; ++i) fb = __builtin_fma(r[T::oK + j * n + i], x[i] - r[T::oX + i], fb);
            u[j] = (r[T::oU + j] + fb) + ff;                                           // :148
        }
        cost = cost + stage_input_cost<m>(a, u, Rk);
        if constexpr (S::n_ang > 0) {
            S::f_sc(x, u, a.dt, p, trig.sc, xn);                                       // :151
            const bool reseed = t + 1 == next_seed;
            if (reseed) next_seed += a.ck;
            trig.advance(x, xn, reseed, !valid || cost > kill_at);
        } else {
            S::f(x, u, a.dt, p, xn);                                                   // :151
        }
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
    };
    int t = 0;
    for (; t + D <= a.N; t += D) {
#pragma unroll
        for (int e = 0; e < D; ++e) one_step(t + e, e);
    }
#pragma unroll
    for (int e = 0; e + 1 < D; ++e)
        if (t + e < a.N) one_step(t + e, e);
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
#ifndef LS_WAVES_ACROBOT
#define LS_WAVES_ACROBOT 2
#endif
constexpr int LS_STILL_SEARCHING = -2;   // acc_j between rounds: no trial accepted yet, more rounds to come
template <class S>
constexpr int ls_waves_per_simd()
{
    return S::n > 4 ? 1 : (S::id == ILQR_DYN_ACROBOT_ID ? LS_WAVES_ACROBOT : LS_WAVES_SMALL);   // (4 for cartpole spills 14 VGPRs and gains nothing)
}
// waves per workgroup of k_linesearch<S, W>: 4 (8 at W = 64), one where a narrow round stages large records through LDS
template <class S, int W, bool TRACK = false>
constexpr int ls_waves_per_block()
{
    return Staged<S, W>::use && !TRACK ? Staged<S, W>::WAVES_PER_BLOCK : (W == 64 ? 8 : 4);
}
template <class S, int W, bool CKPT, bool TRACK = false>
__global__ void __launch_bounds__((64 * ls_waves_per_block<S, W, TRACK>()), ls_waves_per_simd<S>()) k_linesearch(const KArgs a)
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
    const double* XDp = TRACK ? a.XD + (size_t)tile * a.state_rows * n * TILE + pl : nullptr;
    const double* UDp = TRACK ? a.UD + (size_t)tile * a.N * m * TILE + pl : nullptr;
    load_col<n>(TRACK ? XDp + (size_t)a.N * n * TILE : a.TARGET + (size_t)tile * n * TILE + pl, target);   // ilqr_mpc.m:213
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double cur_cost = a.cost[b];
    const double dg = a.dJ[((size_t)tile * 3 + 1) * TILE + pl], dh = a.dJ[((size_t)tile * 3 + 2) * TILE + pl];
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + pl;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + pl;
    const double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + pl;
    const double* kp = a.KFF + (size_t)tile * a.N * m * TILE + pl;
    const int j = a.ls_j0 + al;
    const bool valid = act && j < a.n_trials;
    double c = __builtin_nan("");
    bool okj = false;
    if constexpr (Staged<S, W>::use && !TRACK) {
        // whole wave together (the record loads are shared); lanes without a trial compute on and are ignored
        __shared__ double sh[Staged<S, W>::WAVES_PER_BLOCK * Staged<S, W>::WAVE_DOUBLES];
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
    } else if constexpr (Gather<S, W>::use && !TRACK) {
        // small records at narrow widths: the wave fetches its 64 / W records cooperatively (closed_loop_gather); lanes
        // without a trial compute on and are ignored
        __shared__ double sh[(W == 64 ? 8 : 4) * Gather<S, W>::WAVE_DOUBLES];
        double* sw = sh + (threadIdx.x >> 6) * Gather<S, W>::WAVE_DOUBLES;
        const long long b_duty = __shfl(b, (lane % G) * W);
        const double alpha = a.alphas[j < a.n_trials ? j : 0];
        double x[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        double* ckp = CKPT ? a.CKPT + (size_t)b * a.n_seg * n * TILE + j : nullptr;
        const double expected = alpha * dg + (alpha * alpha) * dh;                 // :192
        const bool provable = a.cost_monotone != 0 && expected > 0.0 && a.armijo_thr >= 0.0;   // see the register path below
        c = closed_loop_gather<S, W, CKPT>(a, sw, lane, grp, b_duty, ckp, valid, x, p, Rk, alpha,
                                           provable ? cur_cost : __builtin_inf());
        c = c + terminal_cost<n>(x, target, QT);
        const double cost_difference = cur_cost - c;                               // :191
        okj = valid && (cost_difference / expected) > a.armijo_thr;                // :193 (NaN -> false)
        if (!valid) c = __builtin_nan("");
    } else if (valid) {
        const double alpha = a.alphas[j];
        double x[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        double* ckp = CKPT ? a.CKPT + (size_t)b * a.n_seg * n * TILE + j : nullptr;
        const double expected = alpha * dg + (alpha * alpha) * dh;                 // :192
        // A trial whose cost exceeds the current cost fails :193 whenever the expected reduction is positive and the
        // threshold is not negative: (J - J_new) / expected < 0.  With R_k, Q_T diagonal and non-negative the running cost
        // never decreases along the rollout (every stage term is a rounded R_jj u_j^2 >= 0), so once it has passed the
        // current cost the outcome is known -- exactly, in floating point -- and the rest of the lane's rollout is a
        // don't-care (diverging trials stop forcing the full trig evaluation on their wave; see TrigCarry::advance).
        const bool provable = !TRACK && a.cost_monotone != 0 && expected > 0.0 && a.armijo_thr >= 0.0;
        c = closed_loop<S, false, CKPT, TRACK>(a, Xp, Up, Kp, kp, nullptr, nullptr, x, p, target, Rk, QT, alpha, ckp,
                                               provable ? cur_cost : __builtin_inf(), XDp, UDp);
        const double cost_difference = cur_cost - c;                               // :191
        okj = (cost_difference / expected) > a.armijo_thr;                         // :193 (NaN -> false)
    }
    const unsigned long long mask = __ballot(okj);
    const unsigned long long gm = (W == 64) ? mask : ((mask >> (grp * W)) & ((1ull << (W % 64)) - 1ull));
    const int first = gm != 0ull ? __ffsll((long long)gm) - 1 : 0;
    const double cwin = __shfl(c, grp * W + first);
    if (al == 0 && act) {
        a.ls_rounds[b] += 1;
        if (gm != 0ull) {
            a.acc_j[b] = a.ls_j0 + first;
            a.new_cost[b] = cwin;
        } else if (!a.ls_last) {                                                   // :200-202: k_ls_compact lists it for the next round
            a.acc_j[b] = LS_STILL_SEARCHING;
        } else {                                                                   // :204 learning rate ran out
            a.acc_j[b] = -1;
            a.new_cost[b] = __builtin_nan("");
        }
    }
}

// The list of the next round: the problems the round just run left searching, in ascending order inside each chunk of
// 1024 problems (one atomic per chunk reserves its range).  Order matters for memory only: neighbours in the list are
// neighbours in a tile, so the lanes of the next round's workgroups share 128-byte lines the way round 0's do
// (appending from the search kernel itself, in completion order, made round 1 fetch 4-8x its algorithmic bytes).
__global__ void __launch_bounds__(1024) k_ls_compact(const KArgs a)
{
    __shared__ int wsum[16];
    __shared__ int base;
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    const long long b = a.tile0 * TILE + i;
    const bool f = i < a.n_tiles * TILE && a.active[b] != 0 && a.acc_j[b] == LS_STILL_SEARCHING;
    const unsigned long long mask = __ballot(f);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) { const int c = wsum[w]; wsum[w] = tot; tot += c; }
        base = tot ? atomicAdd(a.ls_count + a.ls_round, tot) : 0;
    }
    __syncthreads();
    if (f) {
        const int slot = base + wsum[wave] + __popcll(mask & ((1ull << lane) - 1ull));
        a.ls_list[(size_t)(a.ls_round & 1) * a.Bpad + a.tile0 * TILE + slot] = (int)b;
    }
}

// ------------------------------------------------------------------------------------------------
// k_forward: ilqr.forwards_pass (ilqr.py:137-158), one lane per problem.
//   ACCEPT = true : the accept step of solve (:194-207): problems with acc_j >= 0 re-roll the winning alpha
//                   in place into X/U, take new_cost as their cost; the others stop with "low learning rate".
//   ACCEPT = false: plain forwards_pass with a per-problem alpha into XO/UO (+ its cost into new_cost).
// ------------------------------------------------------------------------------------------------
template <class S, bool ACCEPT, bool TRACK = false>
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
        if (act && ls_ends_solve(a, accj)) { a.status[b] = ILQR_STATUS_LOW_LR; a.active[b] = 0; }   // :204-207
        if (__ballot(run) == 0ull) return;
    } else {
        alpha = a.alpha_pp[b];
    }
    double x[n], p[np], target[n], Rk[m * m], QT[n * n];
    load_col<n>(a.INIT + (size_t)tile * n * TILE + lane, x);
    load_col<np>(a.PARAMS + (size_t)tile * np * TILE + lane, p);
    const double* XDp = TRACK ? a.XD + (size_t)tile * a.state_rows * n * TILE + lane : nullptr;
    const double* UDp = TRACK ? a.UD + (size_t)tile * a.N * m * TILE + lane : nullptr;
    load_col<n>(TRACK ? XDp + (size_t)a.N * n * TILE : a.TARGET + (size_t)tile * n * TILE + lane, target);
    load_uniform<m * m>(a.WTS + n * n, Rk);
    load_uniform<n * n>(a.WTS + n * n + m * m, QT);
    const double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    const double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    const double* kp = a.KFF + (size_t)tile * a.N * m * TILE + lane;
    double* XOp = a.XO + (size_t)tile * a.state_rows * n * TILE + lane;
    double* UOp = a.UO + (size_t)tile * a.N * m * TILE + lane;
    if (run) {
        double x0[n];
#pragma unroll
        for (int i = 0; i < n; ++i) x0[i] = a.zero_first ? 0.0 : x[i];                 // :138,:143 (row 0 stays zero)
        const double c = closed_loop<S, true, false, TRACK>(a, Xp, Up, Kp, kp, XOp, UOp, x, p, target, Rk, QT, alpha, nullptr,
                                                            __builtin_inf(), XDp, UDp);
        // row 0 last: XO may be X itself (ilqr_ws_mpc_solve rolls out in place), whose old row 0 is step 0's reference state
        if constexpr (!ACCEPT) store_col<n>(XOp, x0);
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
            if (a.fwd_to_cost) a.cost[b] = c;                                          // ilqr_mpc.m:132
        }
    }
}

// Warm start of a receding-horizon solve (matlab/smooth-ilqr/ilqr_mpc.m:100-121): the previous solution moved `span` steps
// along the horizon, its end padded from the new target window (XD, UD -- already holding the window of the new index), the
// gains K moved span - 1 steps (the reference keeps rows span..end, 1-based) with the last gain repeated.  k_feedforward
// is left alone (the next pass uses learning rate 0, :127).  One lane per problem, ascending t: row t + span is read before
// row t is written.
__global__ void __launch_bounds__(64) k_mpc_shift(const KArgs a, const int n, const int m, const int span)
{
    const long long tile = a.tile0 + blockIdx.x;
    const int lane = threadIdx.x;
    double* Xp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
    double* Up = a.U + (size_t)tile * a.N * m * TILE + lane;
    double* Kp = a.KFB + (size_t)tile * a.N * (m * n) * TILE + lane;
    const double* XDp = a.XD + (size_t)tile * a.state_rows * n * TILE + lane;
    const double* UDp = a.UD + (size_t)tile * a.N * m * TILE + lane;
    for (int t = 0; t <= a.N; ++t)
        for (int i = 0; i < n; ++i)
            Xp[((size_t)t * n + i) * TILE] = t + span <= a.N ? Xp[((size_t)(t + span) * n + i) * TILE] : XDp[((size_t)t * n + i) * TILE];
    for (int t = 0; t < a.N; ++t)
        for (int j = 0; j < m; ++j)
            Up[((size_t)t * m + j) * TILE] = t + span < a.N ? Up[((size_t)(t + span) * m + j) * TILE] : UDp[((size_t)t * m + j) * TILE];
    if (span > 1)
        for (int t = 0; t < a.N; ++t) {
            const int ts = t + span - 1 < a.N ? t + span - 1 : a.N - 1;
            for (int e = 0; e < m * n; ++e) Kp[((size_t)t * (m * n) + e) * TILE] = Kp[((size_t)ts * (m * n) + e) * TILE];
        }
}

// ------------------------------------------------------------------------------------------------
// Split pipeline (small batches: the horizon-sequential kernels would leave most SIMDs idle).
//   k_accept_lin : the accept step (:194-207) in parallel over (tile, segment): each lane restarts the winning
//                  trial from the checkpoint the line search stored at its segment boundary and re-rolls ck steps
//                  IN PLACE into X/U (same arithmetic, same order => same bits as the trial itself), and evaluates
//                  A_t, B_t (ilqr.py:96-97) at the new (x_t, u_t) into LIN on the way.  The only row two segments
//                  share is the boundary row (last state of one = reference state of the next one's first step):
//                  k_save_boundaries copies those rows to XB beforehand.
//   k_linearize  : parallel over (tile, block of time steps): A_t, B_t of the initial rollout into LIN.
//   k_riccati    : the horizon-sequential part of backwards_pass (:99-125) reading LIN instead of calling A(), B().
// ------------------------------------------------------------------------------------------------
template <class S>
__device__ __forceinline__ void accept_lin_body(const KArgs& a, const unsigned bid)
{
    constexpr int n = S::n, m = S::m, np = S::np;
    const long long tile = a.tile0 + bid / a.seg_cnt;                                  // grid = tiles x seg_cnt
    const int seg = a.seg_lo + (int)(bid % a.seg_cnt);
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
        double* XOp = a.X + (size_t)tile * a.state_rows * n * TILE + lane;
        double* UOp = a.U + (size_t)tile * a.N * m * TILE + lane;
        double* Lp = a.LIN + (size_t)tile * a.N * Lin<S>::n_lin * TILE + lane;
        const double* Xfirst = seg > 0 ? a.XB + ((size_t)(tile * a.n_seg + seg) * n) * TILE + lane : nullptr;
        const int t0 = seg * a.ck;
        const int t1 = t0 + a.ck < a.N ? t0 + a.ck : a.N;
        (void)closed_loop_range<S, true, false, ACCEPT_FUSES_LINEARIZE != 0>(a, Xp, Up, Kp, kp, XOp, UOp, nullptr, x, p, Rk, alpha, t0, t1, Xfirst, Lp);
    }
    if (seg == 0 && act) {                                                             // per-problem bookkeeping, once
        if (ls_ends_solve(a, accj)) { a.status[b] = ILQR_STATUS_LOW_LR; a.active[b] = 0; }   // :204-207
        if (accj >= 0) {
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
__global__ void __launch_bounds__(64) k_accept_lin(const KArgs a)
{
    accept_lin_body<S>(a, blockIdx.x);
}
// The accept step of iteration i over the segments of one chunk of the horizon (workgroups n_ric ...) and the Riccati
// recursion of iteration i + 1 over the steps of the chunk ABOVE it (workgroups 0 .. n_ric - 1: dispatched first, they are
// the critical path), in ONE launch.  The recursion's chunk needs the accept step's result for ITS steps -- the previous
// launch on the same stream -- and the value function of the chunk above it -- the previous launch as well; the two
// halves of this launch touch disjoint time steps.  Stream order is the only synchronisation: no second stream, no events.
template <class S>
__global__ void __launch_bounds__(64) k_accept_riccati(const KArgs aa, const KArgs ar, const unsigned n_ric)
{
    if constexpr (S::n <= 4) {
        if (blockIdx.x < n_ric) riccati_quad_body<S, true>(ar, blockIdx.x);
        else accept_lin_body<S>(aa, blockIdx.x - n_ric);
    }
}

// reference-state rows at the segment boundaries, before k_accept_lin overwrites X in place (system independent)
__global__ void __launch_bounds__(64) k_save_boundaries(const KArgs a, const int n)
{
    const long long tile = a.tile0 + blockIdx.x / a.n_seg;
    const int seg = (int)(blockIdx.x % a.n_seg);
    const int lane = threadIdx.x;
    if (seg == 0) return;
    const double* src = a.X + ((size_t)tile * a.state_rows + (size_t)seg * a.ck) * n * TILE + lane;
    double* dst = a.XB + ((size_t)(tile * a.n_seg + seg) * n) * TILE + lane;
    for (int i = 0; i < n; ++i) dst[(size_t)i * TILE] = src[(size_t)i * TILE];
}

