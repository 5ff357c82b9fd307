/* oracle/ilqr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference algorithm for the hot path of Kongx231/ilqr
 * (python/ilqr.py and python/ltvlqr.py), one function per reference function, every
 * expression evaluated in the reference's order (python `@` chains are left-associative),
 * no FMA contraction (build with -ffp-contract=off), float64 throughout.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the
 * checker / the CPU baseline only.  ilqr_amd/ (the product) never loads this library.
 *
 * Pinned against the reference itself: tools/make_golden.py imports the reference python in the
 * build container and commits its outputs under tests/golden/; tests/test_cpu.py
 * checks every function below against those fixtures (and against the reference's own committed
 * acrobot_swingup_{states,inputs}.csv rows kept in the fixtures).
 *
 * Dynamics: the generated headers ilqr_amd/csrc/dynamics/<system>.h (same expression tree and
 * literals as the reference's lambdified callables, symbolic_dynamics.py:23-25,88-90; cartpole,
 * quadcopter, car, double_integrator and double_pendulum are re-derived from the MATLAB specs), compiled here by gcc against glibc libm.
 *
 * Layout of every array: the reference's numpy layout, C-contiguous float64, with an optional
 * leading batch axis handled by the *_batched entry points (OpenMP over problems).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ILQR_DYN_FN static inline
#define ILQR_DYN_CONST static const
/* numpy's sin/cos are glibc's: one libm call each, as the lambdified callables make */
#define ILQR_DYN_SINCOS(x, s, c) do { (s) = sin(x); (c) = cos(x); } while (0)
#define ILQR_DYN_SINCOS_SLOW(x, s, c) do { } while (0)
#define ILQR_DYN_F_PROLOGUE              /* the reference: every operation rounded separately */
#define ILQR_DYN_TAN_SC(x, s, c) tan(x)   /* the reference: numpy tan */
#define ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c) do { (s) = sin((a) + (b)); (c) = cos((a) + (b)); } while (0)   /* the reference: numpy sin/cos of the rounded sum */
#define ILQR_DYN_SINCOS_SUM_SC(a, b, sa, ca, sb, cb, s, c) ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c)   /* (f_sc is not used here) */
#define ILQR_DYN_TRIG_BIG(x) 0
#define ILQR_DYN_TAN(x) tan(x)
/* plain IEEE division, as numpy does; the hoisted reciprocal is a device-side device and unused here */
#define ILQR_DYN_RCP(d) 0.0
#define ILQR_DYN_DIV(a, d, rd) ((a) / (d))
#include "../ilqr_amd/csrc/dynamics/pendulum.h"
#include "../ilqr_amd/csrc/dynamics/acrobot.h"
#include "../ilqr_amd/csrc/dynamics/cartpole.h"
#include "../ilqr_amd/csrc/dynamics/quadcopter.h"
#include "../ilqr_amd/csrc/dynamics/car.h"
#include "../ilqr_amd/csrc/dynamics/double_integrator.h"
#include "../ilqr_amd/csrc/dynamics/double_pendulum.h"
#include "../ilqr_amd/csrc/dynamics/floating_base.h"

#define MAXN 12
#define MAXM 12

typedef void (*dyn_fn)(const double*, const double*, const double, const double*, double*);
typedef struct { int n, m, np; dyn_fn f, A, B; } oracle_system;

#define N_SYSTEMS 8
static const oracle_system SYSTEMS[N_SYSTEMS] = {
    {ILQR_DYN_PENDULUM_N, ILQR_DYN_PENDULUM_M, ILQR_DYN_PENDULUM_NP, ilqr_dyn_pendulum_f, ilqr_dyn_pendulum_A, ilqr_dyn_pendulum_B},
    {ILQR_DYN_ACROBOT_N, ILQR_DYN_ACROBOT_M, ILQR_DYN_ACROBOT_NP, ilqr_dyn_acrobot_f, ilqr_dyn_acrobot_A, ilqr_dyn_acrobot_B},
    {ILQR_DYN_CARTPOLE_N, ILQR_DYN_CARTPOLE_M, ILQR_DYN_CARTPOLE_NP, ilqr_dyn_cartpole_f, ilqr_dyn_cartpole_A, ilqr_dyn_cartpole_B},
    {ILQR_DYN_QUADCOPTER_N, ILQR_DYN_QUADCOPTER_M, ILQR_DYN_QUADCOPTER_NP, ilqr_dyn_quadcopter_f, ilqr_dyn_quadcopter_A, ilqr_dyn_quadcopter_B},
    {ILQR_DYN_CAR_N, ILQR_DYN_CAR_M, ILQR_DYN_CAR_NP, ilqr_dyn_car_f, ilqr_dyn_car_A, ilqr_dyn_car_B},
    {ILQR_DYN_DOUBLE_INTEGRATOR_N, ILQR_DYN_DOUBLE_INTEGRATOR_M, ILQR_DYN_DOUBLE_INTEGRATOR_NP, ilqr_dyn_double_integrator_f, ilqr_dyn_double_integrator_A, ilqr_dyn_double_integrator_B},
    {ILQR_DYN_DOUBLE_PENDULUM_N, ILQR_DYN_DOUBLE_PENDULUM_M, ILQR_DYN_DOUBLE_PENDULUM_NP, ilqr_dyn_double_pendulum_f, ilqr_dyn_double_pendulum_A, ilqr_dyn_double_pendulum_B},
    {ILQR_DYN_FLOATING_BASE_N, ILQR_DYN_FLOATING_BASE_M, ILQR_DYN_FLOATING_BASE_NP, ilqr_dyn_floating_base_f, ilqr_dyn_floating_base_A, ilqr_dyn_floating_base_B},
};

int oracle_system_dims(int sys, int* n, int* m, int* np)
{
    if (sys < 0 || sys >= N_SYSTEMS) return -1;
    *n = SYSTEMS[sys].n; *m = SYSTEMS[sys].m; *np = SYSTEMS[sys].np;
    return 0;
}

/* ---- tiny dense helpers (row-major); sums run k = 0..K-1 starting from 0.0 ---- */
static void mm(int M, int K, int N, const double* A, const double* B, double* C)      /* C = A @ B */
{
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j];
            C[i * N + j] = s;
        }
}
static void mtm(int K, int M, int N, const double* A, const double* B, double* C)     /* C = A.T @ B, A is K x M */
{
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += A[k * M + i] * B[k * N + j];
            C[i * N + j] = s;
        }
}
/* np.linalg.inv: LU with partial (row) pivoting, then solve for the identity columns.
 * Returns 0, or -1 on an exactly zero pivot (numpy raises LinAlgError there). */
static int inv_small(int m, const double* Ain, double* Ainv)
{
    double a[MAXM * MAXM]; int piv[MAXM];
    memcpy(a, Ain, sizeof(double) * m * m);
    if (m == 1) { if (a[0] == 0.0) return -1; Ainv[0] = 1.0 / a[0]; return 0; }
    for (int c = 0; c < m; ++c) {
        int pr = c; double best = fabs(a[c * m + c]);
        for (int r = c + 1; r < m; ++r) if (fabs(a[r * m + c]) > best) { best = fabs(a[r * m + c]); pr = r; }
        piv[c] = pr;
        if (best == 0.0) return -1;
        if (pr != c) for (int j = 0; j < m; ++j) { double t = a[c * m + j]; a[c * m + j] = a[pr * m + j]; a[pr * m + j] = t; }
        for (int r = c + 1; r < m; ++r) {
            a[r * m + c] = a[r * m + c] / a[c * m + c];
            for (int j = c + 1; j < m; ++j) a[r * m + j] -= a[r * m + c] * a[c * m + j];
        }
    }
    for (int col = 0; col < m; ++col) {
        double b[MAXM];
        for (int i = 0; i < m; ++i) b[i] = (i == col) ? 1.0 : 0.0;
        for (int c = 0; c < m; ++c) if (piv[c] != c) { double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
        for (int i = 0; i < m; ++i) for (int j = 0; j < i; ++j) b[i] -= a[i * m + j] * b[j];
        for (int i = m - 1; i >= 0; --i) {
            for (int j = i + 1; j < m; ++j) b[i] -= a[i * m + j] * b[j];
            b[i] = b[i] / a[i * m + i];
        }
        for (int i = 0; i < m; ++i) Ainv[i * m + col] = b[i];
    }
    return 0;
}

/* ---- lambdified callables: symbolic_dynamics.py:23-25, 88-90 ---- */
/* which: 0 f (n), 1 A (n*n), 2 B (n*m); x (count,n), u (count,m), p (np) shared */
int oracle_dynamics(int sys, int which, int64_t count, const double* x, const double* u, double dt,
                    const double* p, double* out)
{
    if (sys < 0 || sys >= N_SYSTEMS) return -1;
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    const int sz = which == 0 ? n : (which == 1 ? n * n : n * m);
    dyn_fn fn = which == 0 ? S->f : (which == 1 ? S->A : S->B);
    for (int64_t i = 0; i < count; ++i) fn(x + i * n, u + i * m, dt, p, out + i * sz);
    return 0;
}

/* ---- ilqr.rollout: ilqr.py:30-47.  states (N+1,n); row 0 is left at zero when zero_first (quirk:
 * the reference never stores init_state, ilqr.py:31-42) ---- */
void oracle_rollout(int sys, int N, const double* init_state, const double* inputs_in, double dt, const double* p,
                    int zero_first, double* states, double* inputs)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    double cur[MAXN], nxt[MAXN];
    for (int i = 0; i < n; ++i) { cur[i] = init_state[i]; states[i] = zero_first ? 0.0 : init_state[i]; }
    for (int t = 0; t < N; ++t) {
        S->f(cur, inputs_in + t * m, dt, p, nxt);
        for (int i = 0; i < n; ++i) { states[(t + 1) * n + i] = nxt[i]; cur[i] = nxt[i]; }
        for (int j = 0; j < m; ++j) inputs[t * m + j] = inputs_in[t * m + j];
    }
}

/* ---- ilqr.compute_cost: ilqr.py:49-62.  J = sum_t (u^T R) u + (d^T Q_T) d, d = target - x_N ---- */
double oracle_cost(int n, int m, int N, const double* states, const double* inputs, const double* Rk, const double* QT,
                   const double* target)
{
    double total = 0.0;
    for (int t = 0; t < N; ++t) {
        const double* u = inputs + t * m;
        double uR[MAXM];
        for (int j = 0; j < m; ++j) { double s = 0.0; for (int i = 0; i < m; ++i) s += u[i] * Rk[i * m + j]; uR[j] = s; }
        double c = 0.0;
        for (int j = 0; j < m; ++j) c += uR[j] * u[j];
        total = total + c;
    }
    double d[MAXN], dQ[MAXN];
    for (int i = 0; i < n; ++i) d[i] = target[i] - states[N * n + i];
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += d[i] * QT[i * n + j]; dQ[j] = s; }
    double tc = 0.0;
    for (int j = 0; j < n; ++j) tc += dQ[j] * d[j];
    total = total + tc;
    return total;
}

/* ---- ilqr.backwards_pass: ilqr.py:64-135.  dJ[3] = {expected_cost_reduction, _grad, _hess}.
 * Returns 0, or -(t+1) if inv(Q_uu) hit an exactly singular matrix at step t. ---- */
int oracle_backward(int sys, int N, const double* states, const double* inputs, const double* target, const double* Qk,
                    const double* Rk, const double* QT, double dt, const double* p, double* k_trj, double* K_trj, double* dJ)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    double Vxx[MAXN * MAXN], Vx[MAXN], ed[MAXN];
    memcpy(Vxx, QT, sizeof(double) * n * n);                                   /* :67 */
    for (int i = 0; i < n; ++i) ed[i] = states[N * n + i] - target[i];         /* :68 */
    mm(n, n, 1, QT, ed, Vx);                                                   /* :70 */
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    for (int t = N - 1; t >= 0; --t) {                                         /* :82 */
        const double* x = states + t * n;
        const double* u = inputs + t * m;
        double A[MAXN * MAXN], B[MAXN * MAXM];
        double lu[MAXM], Qx[MAXN], Qu[MAXM], BtV[MAXM * MAXN], AtV[MAXN * MAXN];
        double Qux[MAXM * MAXN], Quu[MAXM * MAXM], Qxx[MAXN * MAXN], Quu_inv[MAXM * MAXM];
        double k[MAXM], K[MAXM * MAXN], tmp[MAXN * MAXN], tmp2[MAXN * MAXN];
        mm(m, m, 1, Rk, u, lu);                                                /* :93  l_u = R_k @ u */
        S->A(x, u, dt, p, A);                                                  /* :96 */
        S->B(x, u, dt, p, B);                                                  /* :97 */
        mtm(n, n, 1, A, Vx, Qx);                                               /* :99  (l_x = 0) */
        mtm(n, m, 1, B, Vx, Qu);                                               /* :100 */
        for (int j = 0; j < m; ++j) Qu[j] = lu[j] + Qu[j];
        mtm(n, m, n, B, Vxx, BtV);                                             /* B.T @ V_xx */
        mm(m, n, n, BtV, A, Qux);                                              /* :101 */
        mm(m, n, m, BtV, B, Quu);                                              /* :102 */
        for (int i = 0; i < m * m; ++i) Quu[i] = Rk[i] + Quu[i];
        mtm(n, n, n, A, Vxx, AtV);
        mm(n, n, n, AtV, A, Qxx);                                              /* :103 */
        for (int i = 0; i < n * n; ++i) Qxx[i] = Qk[i] + Qxx[i];
        if (inv_small(m, Quu, Quu_inv) != 0) return -(t + 1);                  /* :106 */
        for (int i = 0; i < m * m; ++i) Quu_inv[i] = -Quu_inv[i];              /* unary minus binds first */
        mm(m, m, 1, Quu_inv, Qu, k);                                           /* :107 */
        mm(m, m, n, Quu_inv, Qux, K);                                          /* :108 */
        for (int j = 0; j < m; ++j) k_trj[t * m + j] = k[j];
        for (int i = 0; i < m * n; ++i) K_trj[t * m * n + i] = K[i];
        double g = 0.0;                                                        /* :115  -Q_u.T @ k */
        for (int j = 0; j < m; ++j) g += (-Qu[j]) * k[j];
        double hk[MAXM], hkQ[MAXM], h = 0.0;                                   /* :116  ((0.5*k.T) @ Q_uu) @ k */
        for (int j = 0; j < m; ++j) hk[j] = 0.5 * k[j];
        for (int j = 0; j < m; ++j) { double s = 0.0; for (int i = 0; i < m; ++i) s += hk[i] * Quu[i * m + j]; hkQ[j] = s; }
        for (int j = 0; j < m; ++j) h += hkQ[j] * k[j];
        double cur = g + h;                                                    /* :117 */
        ecr_g += g; ecr_h += h; ecr += cur;                                    /* :119-121 */
        /* :124  V_x = Q_x + K.T@Q_uu@k + K.T@Q_u + Q_ux.T@k */
        double KtQuu[MAXN * MAXM], v1[MAXN], v2[MAXN], v3[MAXN];
        mtm(m, n, m, K, Quu, KtQuu);
        mm(n, m, 1, KtQuu, k, v1);
        mtm(m, n, 1, K, Qu, v2);
        mtm(m, n, 1, Qux, k, v3);
        for (int i = 0; i < n; ++i) Vx[i] = ((Qx[i] + v1[i]) + v2[i]) + v3[i];
        /* :125  V_xx = Q_xx + Q_ux.T@K + K.T@Q_ux + K.T@Q_uu@K */
        double t3[MAXN * MAXN];
        mtm(m, n, n, Qux, K, tmp);
        mtm(m, n, n, K, Qux, tmp2);
        mm(n, m, n, KtQuu, K, t3);
        for (int i = 0; i < n * n; ++i) Vxx[i] = ((Qxx[i] + tmp[i]) + tmp2[i]) + t3[i];
    }
    dJ[0] = ecr; dJ[1] = ecr_g; dJ[2] = ecr_h;                                 /* :128-130 */
    return 0;
}

/* ---- ilqr.forwards_pass: ilqr.py:137-158 ---- */
void oracle_forward(int sys, int N, const double* init_state, const double* ref_states, const double* ref_inputs,
                    const double* k_ff, const double* K_fb, double learning_rate, double dt, const double* p,
                    int zero_first, double* states, double* inputs)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    double cur[MAXN], nxt[MAXN], u[MAXM];
    for (int i = 0; i < n; ++i) { cur[i] = init_state[i]; states[i] = zero_first ? 0.0 : init_state[i]; }
    for (int t = 0; t < N; ++t) {
        for (int j = 0; j < m; ++j) {
            const double ff = learning_rate * k_ff[t * m + j];                 /* :146 */
            double fb = 0.0;                                                   /* :147 */
            for (int i = 0; i < n; ++i) fb += K_fb[(t * m + j) * n + i] * (cur[i] - ref_states[t * n + i]);
            u[j] = (ref_inputs[t * m + j] + fb) + ff;                          /* :148 */
        }
        S->f(cur, u, dt, p, nxt);                                              /* :151 */
        for (int i = 0; i < n; ++i) { states[(t + 1) * n + i] = nxt[i]; cur[i] = nxt[i]; }
        for (int j = 0; j < m; ++j) inputs[t * m + j] = u[j];
    }
}

/* ---- ilqr.solve: ilqr.py:159-211.
 * opts[4] = {learning_speed 0.95, low_learning_rate 0.05, low_expected_reduction 1e-3, armijo_threshold 0.1}.
 * trace (optional, n_iterations x 6): {cost at start, expected reduction, _grad, _hess, accepted trial j or -1, cost after}.
 * status: 0 ran all iterations, 1 "optimal" break (:175-179), 2 "low learning rate" break (:204-207),
 *         3 exactly singular Q_uu (numpy would raise).  iters = backward passes executed. ---- */
int oracle_solve(int sys, int N, const double* init_state, const double* target, const double* guess, const double* Qk,
                 const double* Rk, const double* QT, double dt, const double* p, int n_iterations, const double* opts,
                 int zero_first, double* states, double* inputs, double* k_ff, double* K_fb, double* cost_out,
                 int32_t* iters_out, int32_t* status_out, double* trace)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    const double learning_speed = opts[0], low_learning_rate = opts[1], low_expected_reduction = opts[2],
                 armijo_threshold = opts[3];
    double* new_states = (double*)malloc(sizeof(double) * (size_t)(N + 1) * n);
    double* new_inputs = (double*)malloc(sizeof(double) * (size_t)N * m);
    oracle_rollout(sys, N, init_state, guess, dt, p, zero_first, states, inputs);       /* :162 */
    double current_cost = oracle_cost(n, m, N, states, inputs, Rk, QT, target);         /* :164 */
    int status = 0, iters = 0;
    if (trace) for (int i = 0; i < n_iterations * 6; ++i) trace[i] = NAN;
    for (int ii = 0; ii < n_iterations; ++ii) {                                         /* :170 */
        double dJ[3];
        ++iters;
        if (trace) trace[ii * 6 + 0] = current_cost;
        if (oracle_backward(sys, N, states, inputs, target, Qk, Rk, QT, dt, p, k_ff, K_fb, dJ) != 0) { status = 3; break; }
        if (trace) { trace[ii * 6 + 1] = dJ[0]; trace[ii * 6 + 2] = dJ[1]; trace[ii * 6 + 3] = dJ[2]; trace[ii * 6 + 4] = -1.0; }
        if (fabs(dJ[0]) < low_expected_reduction) { status = 1; break; }                /* :175 */
        double learning_rate = 1.0;
        int armijo_flag = 0, j = 0;
        while (learning_rate > 0.05 && armijo_flag == 0) {                              /* :185 (literal 0.05) */
            oracle_forward(sys, N, init_state, states, inputs, k_ff, K_fb, learning_rate, dt, p, zero_first,
                           new_states, new_inputs);
            const double new_cost = oracle_cost(n, m, N, new_states, new_inputs, Rk, QT, target);
            const double cost_difference = current_cost - new_cost;                     /* :191 */
            const double expected = learning_rate * dJ[1] + learning_rate * learning_rate * dJ[2];   /* :192 */
            armijo_flag = (cost_difference / expected > armijo_threshold);              /* :193 */
            if (armijo_flag) {
                current_cost = new_cost;
                memcpy(states, new_states, sizeof(double) * (size_t)(N + 1) * n);
                memcpy(inputs, new_inputs, sizeof(double) * (size_t)N * m);
                if (trace) { trace[ii * 6 + 4] = (double)j; trace[ii * 6 + 5] = new_cost; }
            } else {
                learning_rate = learning_speed * learning_rate;                         /* :202 */
                ++j;
            }
        }
        if (learning_rate < low_learning_rate) { status = 2; break; }                   /* :204 */
    }
    free(new_states); free(new_inputs);
    *cost_out = current_cost; *iters_out = iters; *status_out = status;
    return 0;
}

/* ---- ltvlqr: ltvlqr.py:3-28.  states (n_rows >= N, n), only rows 0..N-1 are read ---- */
int oracle_ltvlqr(int sys, int N, const double* states, const double* inputs, const double* Qk, const double* Rk,
                  const double* QT, double dt, const double* p, double* K_out)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    double P[MAXN * MAXN];
    memcpy(P, QT, sizeof(double) * n * n);                                              /* :14 */
    for (int t = N - 1; t >= 0; --t) {
        double A[MAXN * MAXN], B[MAXN * MAXM], BtP[MAXM * MAXN], G[MAXM * MAXM], Ginv[MAXM * MAXM], nGinv[MAXM * MAXM];
        double t1[MAXM * MAXN], t2[MAXM * MAXN], K[MAXM * MAXN];
        double AtP[MAXN * MAXN], AtPA[MAXN * MAXN], AtPB[MAXN * MAXM], c1[MAXN * MAXM], c2[MAXN * MAXN], c3[MAXN * MAXN], c4[MAXN * MAXN];
        S->A(states + t * n, inputs + t * m, dt, p, A);                                 /* :21 */
        S->B(states + t * n, inputs + t * m, dt, p, B);                                 /* :22 */
        mtm(n, m, n, B, P, BtP);
        mm(m, n, m, BtP, B, G);
        for (int i = 0; i < m * m; ++i) G[i] = Rk[i] + G[i];
        if (inv_small(m, G, Ginv) != 0) return -(t + 1);
        /* :24  K = (((-inv) @ B.T) @ P) @ A */
        for (int i = 0; i < m * m; ++i) nGinv[i] = -Ginv[i];
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int q = 0; q < m; ++q) s += nGinv[i * m + q] * B[j * m + q]; t1[i * n + j] = s; }
        mm(m, n, n, t1, P, t2);
        mm(m, n, n, t2, A, K);
        for (int i = 0; i < m * n; ++i) K_out[t * m * n + i] = K[i];
        /* :27  P = (Q + (A.T@P)@A) - (((((A.T@P)@B)@inv)@B.T)@P)@A */
        mtm(n, n, n, A, P, AtP);
        mm(n, n, n, AtP, A, AtPA);
        mm(n, n, m, AtP, B, AtPB);
        mm(n, m, m, AtPB, Ginv, c1);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int q = 0; q < m; ++q) s += c1[i * m + q] * B[j * m + q]; c2[i * n + j] = s; }
        mm(n, n, n, c2, P, c3);
        mm(n, n, n, c3, A, c4);
        for (int i = 0; i < n * n; ++i) P[i] = (Qk[i] + AtPA[i]) - c4[i];
    }
    return 0;
}

/* ---- batched drivers (OpenMP over problems).  target/params are shared when their stride is 0. ---- */
int oracle_solve_batched(int sys, int64_t B, int N, const double* init_state, const double* target, int64_t target_stride,
                         const double* guess, const double* Qk, const double* Rk, const double* QT, double dt,
                         const double* params, int64_t param_stride, int n_iterations, const double* opts, int zero_first,
                         double* states, double* inputs, double* k_ff, double* K_fb, double* cost, int32_t* iters,
                         int32_t* status, double* trace, int n_threads)
{
    if (sys < 0 || sys >= N_SYSTEMS) return -1;
    const int n = SYSTEMS[sys].n, m = SYSTEMS[sys].m;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        oracle_solve(sys, N, init_state + b * n, target + b * target_stride, guess + b * (int64_t)N * m, Qk, Rk, QT, dt,
                     params + b * param_stride, n_iterations, opts, zero_first,
                     states + b * (int64_t)(N + 1) * n, inputs + b * (int64_t)N * m, k_ff + b * (int64_t)N * m,
                     K_fb + b * (int64_t)N * m * n, cost + b, iters + b, status + b,
                     trace ? trace + b * (int64_t)n_iterations * 6 : NULL);
    }
    return 0;
}

int oracle_ltvlqr_batched(int sys, int64_t B, int N, int n_rows, const double* states, const double* inputs,
                          const double* Qk, const double* Rk, const double* QT, double dt, const double* params,
                          int64_t param_stride, double* K_out, int n_threads)
{
    if (sys < 0 || sys >= N_SYSTEMS) return -1;
    const int n = SYSTEMS[sys].n, m = SYSTEMS[sys].m;
    int rc = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(min : rc)
    for (int64_t b = 0; b < B; ++b) {
        int r = oracle_ltvlqr(sys, N, states + b * (int64_t)n_rows * n, inputs + b * (int64_t)N * m, Qk, Rk, QT, dt,
                              params + b * param_stride, K_out + b * (int64_t)N * m * n);
        if (r < rc) rc = r;                         /* error codes are negative */
    }
    return rc;
}

/* =====================================================================================================
 * Receding-horizon trajectory-tracking iLQR: matlab/smooth-ilqr/ilqr_mpc.m (MATLAB only: the reference has no python
 * twin and MATLAB cannot run here, so this restatement is NOT pinned by reference outputs -- "parity unpinned"; it
 * follows the .m file line by line, and the GPU path is tested against it and, in the degenerate case x_des = target,
 * u_des = 0, Q_k = 0, against the pinned ilqr.py path).  Layout: states (N+1,n) with row 0 = the current state (:314),
 * x_des (N+1,n) = target_states(current_idx : current_idx+N), u_des (N,m) = target_inputs(current_idx : current_idx+N-1).
 * ===================================================================================================== */

/* x = A \ b for the small square A (MATLAB mldivide: LU with partial pivoting); nrhs right-hand sides in the columns of
 * the row-major b (m x nrhs).  Returns -1 on an exactly zero pivot (MATLAB warns and returns Inf/NaN). */
static int solve_small(int m, const double* Ain, const double* b, int nrhs, double* x)
{
    double a[MAXM * MAXM], y[MAXM]; int piv[MAXM];
    memcpy(a, Ain, sizeof(double) * m * m);
    for (int c = 0; c < m; ++c) {
        int pr = c; double best = fabs(a[c * m + c]);
        for (int r = c + 1; r < m; ++r) if (fabs(a[r * m + c]) > best) { best = fabs(a[r * m + c]); pr = r; }
        piv[c] = pr;
        if (best == 0.0) return -1;
        if (pr != c) for (int j = 0; j < m; ++j) { double t = a[c * m + j]; a[c * m + j] = a[pr * m + j]; a[pr * m + j] = t; }
        for (int r = c + 1; r < m; ++r) {
            a[r * m + c] = a[r * m + c] / a[c * m + c];
            for (int j = c + 1; j < m; ++j) a[r * m + j] -= a[r * m + c] * a[c * m + j];
        }
    }
    for (int col = 0; col < nrhs; ++col) {
        for (int i = 0; i < m; ++i) y[i] = b[i * nrhs + col];
        for (int c = 0; c < m; ++c) if (piv[c] != c) { double t = y[c]; y[c] = y[piv[c]]; y[piv[c]] = t; }
        for (int i = 0; i < m; ++i) for (int j = 0; j < i; ++j) y[i] -= a[i * m + j] * y[j];
        for (int i = m - 1; i >= 0; --i) {
            for (int j = i + 1; j < m; ++j) y[i] -= a[i * m + j] * y[j];
            y[i] = y[i] / a[i * m + i];
        }
        for (int i = 0; i < m; ++i) x[i * nrhs + col] = y[i];
    }
    return 0;
}

/* ilqr_mpc.compute_cost: ilqr_mpc.m:193-216 */
double oracle_mpc_cost(int n, int m, int N, const double* states, const double* inputs, const double* x_des, const double* u_des,
                       const double* Qk, const double* Rk, const double* QT)
{
    double total = 0.0;
    for (int t = 0; t < N; ++t) {                                              /* :196 */
        double xd[MAXN], ud[MAXM], tmp[MAXN];
        for (int i = 0; i < n; ++i) xd[i] = x_des[t * n + i] - states[t * n + i];   /* :207 */
        for (int j = 0; j < m; ++j) ud[j] = u_des[t * m + j] - inputs[t * m + j];   /* :208 */
        double cu = 0.0, cx = 0.0;                                             /* :209  (u'R)u + (x'Q)x */
        for (int j = 0; j < m; ++j) { double s = 0.0; for (int i = 0; i < m; ++i) s += ud[i] * Rk[i * m + j]; tmp[j] = s; }
        for (int j = 0; j < m; ++j) cu += tmp[j] * ud[j];
        for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += xd[i] * Qk[i * n + j]; tmp[j] = s; }
        for (int j = 0; j < n; ++j) cx += tmp[j] * xd[j];
        total = total + (cu + cx);                                             /* :210 */
    }
    double d[MAXN], dQ[MAXN], tc = 0.0;
    for (int i = 0; i < n; ++i) d[i] = x_des[N * n + i] - states[N * n + i];    /* :213 */
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += d[i] * QT[i * n + j]; dQ[j] = s; }
    for (int j = 0; j < n; ++j) tc += dQ[j] * d[j];
    return total + tc;                                                         /* :215 */
}

/* ilqr_mpc.backwards_pass: ilqr_mpc.m:236-308.  Returns -(t+1) on an exactly singular Q_uu. */
int oracle_mpc_backward(int sys, int N, const double* states, const double* inputs, const double* x_des, const double* u_des,
                        const double* Qk, const double* Rk, const double* QT, double dt, const double* p, double* k_trj,
                        double* K_trj, double* dJ)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    double Vxx[MAXN * MAXN], Vx[MAXN], ed[MAXN];
    for (int i = 0; i < n; ++i) ed[i] = states[N * n + i] - x_des[N * n + i];  /* :247 */
    mm(n, n, 1, QT, ed, Vx);
    memcpy(Vxx, QT, sizeof(double) * n * n);                                   /* :248 */
    double ecr = 0.0, ecr_g = 0.0, ecr_h = 0.0;
    for (int t = N - 1; t >= 0; --t) {                                         /* :250 */
        const double* x = states + t * n;
        const double* u = inputs + t * m;
        double A[MAXN * MAXN], B[MAXN * MAXM], ex[MAXN], eu[MAXM], lx[MAXN], lu[MAXM];
        double Qx[MAXN], Qu[MAXM], BtV[MAXM * MAXN], AtV[MAXN * MAXN], Qux[MAXM * MAXN], Quu[MAXM * MAXM], Qxx[MAXN * MAXN];
        double nQuu[MAXM * MAXM], k[MAXM], K[MAXM * MAXN];
        for (int i = 0; i < n; ++i) ex[i] = x[i] - x_des[t * n + i];
        for (int j = 0; j < m; ++j) eu[j] = u[j] - u_des[t * m + j];
        mm(n, n, 1, Qk, ex, lx);                                               /* :260 */
        mm(m, m, 1, Rk, eu, lu);                                               /* :263 */
        S->A(x, u, dt, p, A);                                                  /* :267 */
        S->B(x, u, dt, p, B);                                                  /* :268 */
        mtm(n, n, 1, A, Vx, Qx);
        for (int i = 0; i < n; ++i) Qx[i] = lx[i] + Qx[i];                     /* :271 */
        mtm(n, m, 1, B, Vx, Qu);
        for (int j = 0; j < m; ++j) Qu[j] = lu[j] + Qu[j];                     /* :272 */
        mtm(n, n, n, A, Vxx, AtV);
        mm(n, n, n, AtV, A, Qxx);
        for (int i = 0; i < n * n; ++i) Qxx[i] = Qk[i] + Qxx[i];               /* :273 */
        mtm(n, m, n, B, Vxx, BtV);
        mm(m, n, n, BtV, A, Qux);                                              /* :274 */
        mm(m, n, m, BtV, B, Quu);
        for (int i = 0; i < m * m; ++i) Quu[i] = Rk[i] + Quu[i];               /* :275 */
        for (int i = 0; i < m * m; ++i) nQuu[i] = -Quu[i];                     /* :278-279  -(Q_uu)\b = (-Q_uu)\b */
        if (solve_small(m, nQuu, Qu, 1, k) != 0) return -(t + 1);
        if (solve_small(m, nQuu, Qux, n, K) != 0) return -(t + 1);
        /* :282  V_x = Q_x + K'*Q_uu*k + K'*Q_u + Q_ux'*k */
        double KtQuu[MAXN * MAXM], v1[MAXN], v2[MAXN], v3[MAXN], t1[MAXN * MAXN], t2[MAXN * MAXN], t3[MAXN * MAXN];
        mtm(m, n, m, K, Quu, KtQuu);
        mm(n, m, 1, KtQuu, k, v1);
        mtm(m, n, 1, K, Qu, v2);
        mtm(m, n, 1, Qux, k, v3);
        for (int i = 0; i < n; ++i) Vx[i] = ((Qx[i] + v1[i]) + v2[i]) + v3[i];
        /* :283  V_xx = Q_xx + Q_ux'*K + K'*Q_ux + K'*Q_uu*K */
        mtm(m, n, n, Qux, K, t1);
        mtm(m, n, n, K, Qux, t2);
        mm(n, m, n, KtQuu, K, t3);
        for (int i = 0; i < n * n; ++i) Vxx[i] = ((Qxx[i] + t1[i]) + t2[i]) + t3[i];
        for (int j = 0; j < m; ++j) k_trj[t * m + j] = k[j];                   /* :286 */
        for (int i = 0; i < m * n; ++i) K_trj[t * m * n + i] = K[i];           /* :287 */
        double g = 0.0, hk[MAXM], hkQ[MAXM], h = 0.0;                          /* :290-291 */
        for (int j = 0; j < m; ++j) g += (-Qu[j]) * k[j];
        for (int j = 0; j < m; ++j) hk[j] = 0.5 * k[j];
        for (int j = 0; j < m; ++j) { double s = 0.0; for (int i = 0; i < m; ++i) s += hk[i] * Quu[i * m + j]; hkQ[j] = s; }
        for (int j = 0; j < m; ++j) h += hkQ[j] * k[j];
        ecr_g += g; ecr_h += h; ecr += g + h;                                  /* :296-298 */
    }
    dJ[0] = ecr; dJ[1] = ecr_g; dJ[2] = ecr_h;
    return 0;
}

/* the warm start of a later call: ilqr_mpc.m:100-121 (states_/inputs_ moved span steps and padded from the new window;
 * K_feedback_ keeps rows span..end (1-based) and repeats its last row) */
void oracle_mpc_shift(int n, int m, int N, int span, const double* x_des, const double* u_des, double* states, double* inputs,
                      double* K_fb)
{
    for (int t = 0; t <= N; ++t) for (int i = 0; i < n; ++i)
        states[t * n + i] = t + span <= N ? states[(t + span) * n + i] : x_des[t * n + i];                  /* :109-110 */
    for (int t = 0; t < N; ++t) for (int j = 0; j < m; ++j)
        inputs[t * m + j] = t + span < N ? inputs[(t + span) * m + j] : u_des[t * m + j];                   /* :107-108 */
    if (span > 1)
        for (int t = 0; t < N; ++t) {                                                                       /* :120-121 */
            const int ts = t + span - 1 < N ? t + span - 1 : N - 1;
            for (int e = 0; e < m * n; ++e) K_fb[t * m * n + e] = K_fb[ts * m * n + e];
        }
}

/* ilqr_mpc.solve_ilqr: ilqr_mpc.m:63-192 for one call.  states/inputs/K_fb in: the warm start (first call: the reference
 * window itself, :73-74; later calls: after oracle_mpc_shift); out: the solution.  opts as oracle_solve; the while
 * condition of the line search uses low_learning_rate (:158), not a literal. */
int oracle_mpc_solve(int sys, int N, int first_call, const double* current_state, const double* x_des, const double* u_des,
                     const double* Qk, const double* Rk, const double* QT, double dt, const double* p, int n_iterations,
                     const double* opts, double* states, double* inputs, double* k_ff, double* K_fb, double* cost_out,
                     int32_t* iters_out, int32_t* status_out, double* trace)
{
    const oracle_system* S = &SYSTEMS[sys];
    const int n = S->n, m = S->m;
    const double learning_speed = opts[0], low_learning_rate = opts[1], low_expected_reduction = opts[2], armijo_threshold = opts[3];
    double* new_states = (double*)malloc(sizeof(double) * (size_t)(N + 1) * n);
    double* new_inputs = (double*)malloc(sizeof(double) * (size_t)N * m);
    double dJ[3];
    int status = 0, iters = 0;
    if (trace) for (int i = 0; i < n_iterations * 6; ++i) trace[i] = NAN;
    if (first_call && oracle_mpc_backward(sys, N, states, inputs, x_des, u_des, Qk, Rk, QT, dt, p, k_ff, K_fb, dJ) != 0) {   /* :75-78 */
        status = 3; goto done;
    }
    oracle_forward(sys, N, current_state, states, inputs, k_ff, K_fb, 0.0, dt, p, 0, new_states, new_inputs);             /* :127 */
    double current_cost = oracle_mpc_cost(n, m, N, new_states, new_inputs, x_des, u_des, Qk, Rk, QT);                      /* :132 */
    memcpy(states, new_states, sizeof(double) * (size_t)(N + 1) * n);                                                      /* :133-134 */
    memcpy(inputs, new_inputs, sizeof(double) * (size_t)N * m);
    for (int ii = 0; ii < n_iterations; ++ii) {                                                                            /* :142 */
        ++iters;
        if (trace) trace[ii * 6 + 0] = current_cost;
        if (oracle_mpc_backward(sys, N, states, inputs, x_des, u_des, Qk, Rk, QT, dt, p, k_ff, K_fb, dJ) != 0) { status = 3; break; }   /* :145 */
        if (trace) { trace[ii * 6 + 1] = dJ[0]; trace[ii * 6 + 2] = dJ[1]; trace[ii * 6 + 3] = dJ[2]; trace[ii * 6 + 4] = -1.0; }
        if (fabs(dJ[0]) < low_expected_reduction) { status = 1; break; }                                                   /* :147 */
        double learning_rate = 1.0;
        int armijo_flag = 0, j = 0;
        while (learning_rate > low_learning_rate && armijo_flag == 0) {                                                    /* :158 */
            oracle_forward(sys, N, current_state, states, inputs, k_ff, K_fb, learning_rate, dt, p, 0, new_states, new_inputs);
            const double new_cost = oracle_mpc_cost(n, m, N, new_states, new_inputs, x_des, u_des, Qk, Rk, QT);
            const double cost_difference = current_cost - new_cost;                                                        /* :163 */
            const double expected = learning_rate * dJ[1] + learning_rate * learning_rate * dJ[2];                         /* :164 */
            armijo_flag = (cost_difference / expected > armijo_threshold);                                                 /* :165 */
            if (armijo_flag) {
                current_cost = new_cost;
                memcpy(states, new_states, sizeof(double) * (size_t)(N + 1) * n);
                memcpy(inputs, new_inputs, sizeof(double) * (size_t)N * m);
                if (trace) { trace[ii * 6 + 4] = (double)j; trace[ii * 6 + 5] = new_cost; }
            } else {
                learning_rate = learning_speed * learning_rate;                                                            /* :175 */
                ++j;
            }
        }
        if (learning_rate < low_learning_rate) { status = 2; break; }                                                      /* :179 */
    }
    *cost_out = current_cost;
done:
    if (status == 3 && iters == 0) *cost_out = NAN;
    free(new_states); free(new_inputs);
    *iters_out = iters; *status_out = status;
    return 0;
}

/* B independent problems (OpenMP); every array carries the leading batch axis, params/weights shared */
int oracle_mpc_solve_batched(int sys, int64_t B, int N, int first_call, const double* current_state, const double* x_des,
                             const double* u_des, const double* Qk, const double* Rk, const double* QT, double dt, const double* p,
                             int n_iterations, const double* opts, double* states, double* inputs, double* k_ff, double* K_fb,
                             double* cost, int32_t* iters, int32_t* status, double* trace, int n_threads)
{
    if (sys < 0 || sys >= N_SYSTEMS) return -1;
    const int n = SYSTEMS[sys].n, m = SYSTEMS[sys].m;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b)
        oracle_mpc_solve(sys, N, first_call, current_state + b * n, x_des + b * (int64_t)(N + 1) * n, u_des + b * (int64_t)N * m, Qk, Rk,
                         QT, dt, p, n_iterations, opts, states + b * (int64_t)(N + 1) * n, inputs + b * (int64_t)N * m,
                         k_ff + b * (int64_t)N * m, K_fb + b * (int64_t)N * m * n, cost + b, iters + b, status + b,
                         trace ? trace + b * (int64_t)n_iterations * 6 : NULL);
    return 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
