/* include/ilqr_hip.h -- C ABI of libilqr_hip.so, the MI355X (gfx950) batched iLQR / LTV-LQR hot path.
 *
 * The reference (Kongx231/ilqr) has no FFI: its boundary is the python API of python/ilqr.py and
 * python/ltvlqr.py.  Each entry point below is what a ctypes binding of that python API binds for one
 * reference function; the citation says which (paths are relative to /root/reference).  The python
 * drop-in classes in ilqr_amd/ are such a binding; INTEGRATION.md shows the stub a maintainer would add
 * to the reference.
 *
 * Conventions
 *   - Plain C, no C++/torch types.  Every function returns 0 on success and a negative code on error,
 *     never throws, never aborts; ilqr_last_error() returns the message of the calling thread's last error.
 *   - All arrays are C-contiguous float64 in the REFERENCE's numpy layout with one extra leading batch
 *     axis B (B independent problems).  "host" pointers are ordinary host memory; the *_device variants
 *     take HIP device pointers in the same layout.
 *   - Inside the library trajectories live in HBM in a tiled structure-of-arrays layout
 *     [tile = b/64][time][component][b%64]; the library converts at the boundary.
 *   - n (states), m (inputs), P (parameters) are fixed by system_id; N = n_timesteps.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_ABI_VERSION 8

/* systems compiled into the library (sympy-generated __device__ dynamics, ilqr_amd/csrc/dynamics/) */
enum { ILQR_SYS_PENDULUM = 0, ILQR_SYS_ACROBOT = 1, ILQR_SYS_CARTPOLE = 2, ILQR_SYS_QUADCOPTER = 3, ILQR_SYS_CAR = 4,
       ILQR_SYS_DOUBLE_INTEGRATOR = 5, ILQR_SYS_DOUBLE_PENDULUM = 6, ILQR_SYS_FLOATING_BASE = 7 };

/* per-problem exit reason of solve(): python/ilqr.py:170-207 */
enum {
    ILQR_STATUS_MAX_ITER = 0,   /* loop ran n_iterations times (ilqr.py:170) */
    ILQR_STATUS_OPTIMAL = 1,    /* |expected reduction| < low_expected_reduction (ilqr.py:175-179) */
    ILQR_STATUS_LOW_LR = 2,     /* line search exhausted (ilqr.py:204-207) */
    ILQR_STATUS_NUMERICAL = 3   /* Q_uu exactly singular: a zero pivot, where np.linalg.inv raises LinAlgError (ilqr.py:106) */
};

/* problem descriptor.  Fill with ilqr_desc_default() first. */
typedef struct ilqr_desc {
    int32_t system_id;            /* ILQR_SYS_* */
    int32_t n_timesteps;          /* N = len(np.arange(start,end,dt)), ilqr.py:15-16 (computed by the caller) */
    int64_t batch;                /* B >= 1 */
    double dt;                    /* ilqr.py:12 */
    int32_t n_iterations;         /* ilqr.py:28 */
    int32_t n_alpha_parallel;     /* line-search trials evaluated concurrently per problem: 0 auto, else 1,2,4,...,64 */
    int32_t zero_first_state;     /* 1 (default): row 0 of states is zero as in the reference (ilqr.py:31-42,143,153) */
    int32_t device;               /* HIP device ordinal */
    double learning_speed;        /* 0.95  ilqr.py:166 */
    double low_learning_rate;     /* 0.05  ilqr.py:167: after the search, learning_rate < this ends the solve (:204) */
    double low_expected_reduction;/* 1e-3  ilqr.py:168 */
    double armijo_threshold;      /* 0.1   ilqr.py:169 */
    double quu_reg;               /* added to diag(Q_uu); 0 = reference behaviour (ilqr.py:106 has none) */
    int32_t target_batched;       /* 0: target_state is (n,) shared; 1: (B,n) */
    int32_t params_batched;       /* 0: parameters is (P,) shared; 1: (B,P) */
    int32_t pipeline;             /* kernel schedule of solve(): 0 auto, 1 fused (backward | line search | accept),
                                     2 split (Jacobians and the accept re-roll in parallel over time; small batches).
                                     Results are bit-identical for both. */
    int32_t n_streams;            /* solve() splits the batch into this many independent sub-batches (contiguous problem
                                     ranges), each with its own launch chain on its own HIP stream, so that one sub-batch's
                                     horizon-sequential kernels overlap another's line search.  0/1: one chain; max 4.
                                     Results do not depend on it. */
    int32_t riccati;              /* mapping of the backward recursion: 0 auto (n <= 4: one lane per problem, or four lanes
                                     per problem for n = 4 in split-pipeline batches of up to 8192 problems; larger n: a row
                                     of 16 lanes), 1 lane, 2 row of 16 lanes, 3 four lanes (n <= 4).  2 and 3 imply
                                     pipeline 2.  1 and 3 give bit-identical results. */
    int32_t overlap;              /* solve() with the four-lane recursion: the accept step of iteration i (parallel over
                                     segments of the horizon, last segments first) and the backward recursion of
                                     iteration i + 1 (which walks the horizon from its end) share launches, chunk by chunk:
                                     one grid holds the recursion over one chunk and the accept step over the chunk below
                                     it, all on the workspace's one stream (no second stream, no events).  0 auto (on when
                                     the horizon gives at least 4 chunks of 4 segments of 64 steps), 1 off (separate
                                     launches), 2 on from two segments up.  Results do not depend on it.  (Rounds 1-2 ran
                                     the two on two streams joined by events; DESIGN.md section 5 has why that went.) */
    double search_floor;          /* 0.05  the LITERAL in `while learning_rate > 0.05 and armijo_flag == 0` (ilqr.py:185):
                                     trials alpha_j = learning_speed^j run while alpha_j > search_floor.  The reference keeps
                                     it apart from low_learning_rate (:167, used at :204 only); so does this library: a trial
                                     accepted at alpha < low_learning_rate is taken AND ends the solve ("low learning rate"),
                                     and an exhausted search ends it only if the next alpha is < low_learning_rate. */
    int32_t n_devices;            /* one-shot host entry points (ilqr_*_batched, ltvlqr_batched): 0/1 = the whole batch on
                                     `device`; G > 1 = the batch is split into G contiguous shards by problem index (sizes
                                     differ by at most one, remainder to the first shards), shard r on device `device + r`,
                                     one host thread per shard, every shard reading and writing its own slice of the
                                     caller's arrays.  Problems are independent (no term of ilqr.py / ltvlqr.py couples
                                     two): no collective, results identical to the single-device call.  The workspace API
                                     is per device (ilqr_ws_create ignores this field). */
    int32_t device_wrap;          /* 0: a shard whose device ordinal is not visible is an error; 1: ordinals wrap modulo the
                                     visible devices (several shards per GPU: tests on a one-GPU box) */
    int32_t tracking;             /* 1: the trajectory-tracking cost of matlab/smooth-ilqr/ilqr_mpc.m -- per-step targets
                                     x_des (N+1 rows, the last one the terminal target) and u_des (fields TARGET_STATES /
                                     TARGET_INPUTS) replace target_state: stage cost (u_des-u)'R_k(u_des-u) + (x_des-x)'Q_k(x_des-x)
                                     (:207-209), l_x = Q_k (x - x_des), l_u = R_k (u - u_des) (:260-263).  Fused schedule.  With
                                     x_des = target_state in every row, u_des = 0 and Q_k = 0 this is ilqr.py's cost, bit for bit. */
} ilqr_desc;

void ilqr_desc_default(ilqr_desc* d);

int ilqr_abi_version(void);
int ilqr_device_count(void);
const char* ilqr_last_error(void);
/* name -> system id; also returns n, m, P.  Replaces the choice of symbolic_dynamics_<name>() (symbolic_dynamics.py:5,28). */
int ilqr_system_lookup(const char* name, int32_t* n, int32_t* m, int32_t* n_params);
int ilqr_system_dims(int32_t system_id, int32_t* n, int32_t* m, int32_t* n_params);

/* The lambdified callables f_disc / A_disc / B_disc: python/symbolic_dynamics.py:23-25, 88-90.
 * which: 0 f -> out (count,n); 1 A -> (count,n,n); 2 B -> (count,n,m).  x (count,n), u (count,m), params (P). */
int ilqr_dynamics_eval(int32_t system_id, int32_t which, int64_t count, const double* x, const double* u, double dt,
                       const double* params, double* out, int32_t device);

/* ilqr.rollout(): python/ilqr.py:30-47.  init_state (B,n), inputs_in (B,N,m), params (P)|(B,P)
 * -> states (B,N+1,n), inputs (B,N,m). */
int ilqr_rollout_batched(const ilqr_desc* d, const double* init_state, const double* inputs_in, const double* params,
                         double* states, double* inputs);

/* ilqr.compute_cost(): python/ilqr.py:49-62.  states (B,N+1,n), inputs (B,N,m), target (n)|(B,n) -> cost (B). */
int ilqr_cost_batched(const ilqr_desc* d, const double* states, const double* inputs, const double* Rk, const double* QT,
                      const double* target, double* cost);

/* ilqr.backwards_pass(): python/ilqr.py:64-135.  -> k (B,N,m), K (B,N,m,n),
 * dJ (B,3) = {expected_cost_reduction_, _grad_, _hess_} (ilqr.py:128-130), status (B) 0 ok / ILQR_STATUS_NUMERICAL. */
int ilqr_backward_batched(const ilqr_desc* d, const double* states, const double* inputs, const double* target,
                          const double* Qk, const double* Rk, const double* QT, const double* params, double* k, double* K,
                          double* dJ, int32_t* status);

/* ilqr.forwards_pass(learning_rate): python/ilqr.py:137-158.  learning_rate (B).  Rk/QT/target/cost may all be NULL;
 * when given, cost (B) = compute_cost(new states, new inputs).
 * Also the closed-loop tracking simulation of python/example_acrobot_swingup_ltvlqr_tracking.py:47-67 and
 * python/example_pendulum_swingup_ltvlqr_tracking.py:61-81 (u = inputs[i] + K[i] @ (x - states[i]), row 0 = the perturbed
 * initial state): call with k = 0, K = the ltvlqr gains and d->zero_first_state = 0 (ilqr_amd.tracking_rollout). */
int ilqr_forward_batched(const ilqr_desc* d, const double* init_state, const double* ref_states, const double* ref_inputs,
                         const double* k, const double* K, const double* learning_rate, const double* params,
                         double* states, double* inputs, const double* Rk, const double* QT, const double* target,
                         double* cost);

/* ilqr.solve(): python/ilqr.py:159-211.  Returns per problem what solve() returns (:211) plus iterations and exit reason.
 * trace (optional, (B,n_iterations,6)): {cost at iteration start, expected reduction, _grad, _hess,
 * accepted line-search trial j (alpha = learning_speed^j by repeated multiplication, -1 none), cost after}; NaN where not run. */
int ilqr_solve_batched(const ilqr_desc* d, const double* init_state, const double* target, const double* guess,
                       const double* Qk, const double* Rk, const double* QT, const double* params, double* states,
                       double* inputs, double* k, double* K, double* cost, int32_t* iters, int32_t* status, double* trace);

/* ltvlqr(): python/ltvlqr.py:3-28.  states (B,n_state_rows>=N,n) (rows 0..N-1 read), inputs (B,N,m) -> K (B,N,m,n).
 * d->n_iterations and the line-search fields are ignored.  np.linalg.inv raises LinAlgError when R + B'PB is exactly
 * singular (ltvlqr.py:24): status (optional, (B)) receives 0 or ILQR_STATUS_NUMERICAL per problem; without it the call
 * returns ILQR_ERR_SINGULAR if any problem hit that case (K is written either way). */
#define ILQR_ERR_SINGULAR (-5)
int ltvlqr_batched(const ilqr_desc* d, int32_t n_state_rows, const double* states, const double* inputs, const double* Qk,
                   const double* Rk, const double* QT, const double* params, double* K, int32_t* status);

/* ltvlqr_batched for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops (an MPC controller refreshing its gain
 * schedule every control step), harnesses that forbid allocation.  Same kernels and results as ltvlqr_batched.
 *   - EVERY pointer is a DEVICE pointer: states (B,n_state_rows,n), inputs (B,N,m), Qk (n,n), Rk (m,m), QT (n,n),
 *     params (P)|(B,P) in, K (B,N,m,n) and the optional status (B) (0 / ILQR_STATUS_NUMERICAL) out.
 *   - The call enqueues its work on `stream` (a hipStream_t; NULL = the default stream) and returns without waiting.
 *   - It calls no hipMalloc / hipFree: the tiled copies of the trajectory and gains live in a static device array of
 *     ILQR_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.  A batch that does
 *     not fit is refused with ILQR_ERR_SCRATCH (the message names the largest batch for that horizon: about
 *     64 * floor(64 MiB / (64 * 8 * ((N+1) n + N m + N m n + P))) problems, e.g. 256 acrobot problems of 3000 steps).
 *   - Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
 *     d->n_devices must be 0 or 1.
 *   - The FIRST call on a device resolves the array's address there (hipGetSymbolAddress, under a mutex): make one
 *     warm-up call per device before a stream capture begins.  Later calls issue kernels and hipMemcpyAsync /
 *     hipMemsetAsync on `stream` only.  Running it UNDER hipStreamBeginCapture has not been exercised by the tests.
 *   - Padding lanes (B not a multiple of 64) are filled by the pack kernels with copies of problem B - 1 on every call:
 *     no lane computes on a previous call's data. */
#define ILQR_NOALLOC_SCRATCH_BYTES (64 * 1024 * 1024)
#define ILQR_ERR_SCRATCH (-6)
int ltvlqr_device_noalloc(const ilqr_desc* d, int32_t n_state_rows, const double* states, const double* inputs,
                          const double* Qk, const double* Rk, const double* QT, const double* params, double* K,
                          int32_t* status, void* stream);

/* ---- device-resident workspace: the same passes with the batch kept in HBM between calls ----
 * (no counterpart in the reference, whose arrays are numpy objects; this is the form bench.py and
 * long-running callers use so that the timed region starts with inputs resident in HBM). */
typedef struct ilqr_ws ilqr_ws;

enum {  /* fields, reference layout with leading B */
    ILQR_F_INIT = 0,      /* (B,n)        in  */
    ILQR_F_TARGET = 1,    /* (n)|(B,n)    in  */
    ILQR_F_PARAMS = 2,    /* (P)|(B,P)    in  */
    ILQR_F_STATES = 3,    /* (B,N+1,n)    in/out */
    ILQR_F_INPUTS = 4,    /* (B,N,m)      in/out (the current trajectory's inputs) */
    ILQR_F_KFF = 5,       /* (B,N,m)      in/out */
    ILQR_F_KFB = 6,       /* (B,N,m,n)    in/out */
    ILQR_F_COST = 7,      /* (B) double   out */
    ILQR_F_DJ = 8,        /* (B,3) double out */
    ILQR_F_ITERS = 9,     /* (B) int32    out */
    ILQR_F_STATUS = 10,   /* (B) int32    out */
    ILQR_F_TRACE = 11,    /* (B,n_iterations,6) double out (only if created with want_trace) */
    ILQR_F_LS_ROUNDS = 12,/* (B) int32 out: line-search rounds (of n_alpha_parallel trials) executed so far */
    ILQR_F_ACCEPTS = 13,  /* (B) int32 out: accepted line-search steps so far */
    ILQR_F_ALPHA = 14,    /* (B) double in: per-problem learning rate for ilqr_ws_forward */
    ILQR_F_STATES_OUT = 15, /* (B,N+1,n) out of ilqr_ws_forward */
    ILQR_F_INPUTS_OUT = 16, /* (B,N,m)   out of ilqr_ws_forward */
    ILQR_F_NEW_COST = 17, /* (B) double out: cost of the last forward / accepted trial */
    ILQR_F_GUESS = 18,    /* (B,N,m)      in: initial guess read by rollout()/solve(); kept, so solve() can be repeated */
    ILQR_F_TARGET_STATES = 19, /* (B,N+1,n) in: tracking cost, x_des of the horizon window (ilqr_mpc.m:203,213) */
    ILQR_F_TARGET_INPUTS = 20  /* (B,N,m)   in: tracking cost, u_des of the horizon window (ilqr_mpc.m:204) */
};

int ilqr_ws_create(const ilqr_desc* d, int32_t want_trace, ilqr_ws** out);
int ilqr_ws_destroy(ilqr_ws* ws);
int ilqr_ws_set_weights(ilqr_ws* ws, const double* Qk, const double* Rk, const double* QT);   /* host pointers */
int ilqr_ws_upload(ilqr_ws* ws, int32_t field, const void* host_src, int64_t rows_per_problem); /* rows: STATES only (0 = N+1) */
int ilqr_ws_upload_device(ilqr_ws* ws, int32_t field, const void* dev_src, int64_t rows_per_problem);
int ilqr_ws_download(ilqr_ws* ws, int32_t field, void* host_dst);
int ilqr_ws_download_device(ilqr_ws* ws, int32_t field, void* dev_dst);
/* passes; all asynchronous on the workspace stream, ilqr_ws_sync() waits */
int ilqr_ws_reset(ilqr_ws* ws);            /* mark every problem active, zero iteration counters */
int ilqr_ws_rollout(ilqr_ws* ws);          /* ilqr.py:30-47 + :49-62 (cost of the rollout) */
int ilqr_ws_cost(ilqr_ws* ws);             /* ilqr.py:49-62 on STATES/INPUTS -> COST */
int ilqr_ws_backward(ilqr_ws* ws, int32_t iteration);   /* ilqr.py:64-135 (+ the "optimal" test :175) */
int ilqr_ws_linesearch(ilqr_ws* ws);       /* ilqr.py:180-202 without the accept copy: finds the first accepted trial */
int ilqr_ws_accept(ilqr_ws* ws, int32_t iteration);     /* ilqr.py:194-207: re-rolls the accepted trial into STATES/INPUTS */
int ilqr_ws_forward(ilqr_ws* ws, int32_t with_cost);    /* ilqr.py:137-158 with ALPHA -> STATES_OUT/INPUTS_OUT (+ NEW_COST) */
int ilqr_ws_solve(ilqr_ws* ws);            /* ilqr.py:159-211: reset, rollout, n_iterations x (backward, linesearch, accept) */
int ilqr_ws_ltvlqr(ilqr_ws* ws);           /* ltvlqr.py:3-28 on STATES/INPUTS -> KFB */
/* Receding-horizon tracking iLQR: matlab/smooth-ilqr/ilqr_mpc.m, solve_ilqr (:63-192), on a workspace created with
 * tracking = 1 and zero_first_state = 0 (row 0 of the states is the current state, :314).  One call of solve_ilqr is
 *   upload INIT (current state), TARGET_STATES / TARGET_INPUTS (the window current_idx .. current_idx + N);
 *   first call: upload STATES / INPUTS = that window (:73-74);  later calls: ilqr_ws_mpc_shift(ws, span) (:100-121: the
 *     previous solution moved span steps, padded from the new window; gains moved span - 1 steps, last gain repeated);
 *   ilqr_ws_mpc_solve(ws, first_call): first call only, one backward pass along the window for the gains (:75-78); then
 *     the closed-loop rollout with learning rate 0 (:127-134) and n_iterations x {backward, line search, accept}
 *     (:142-184) -> STATES, INPUTS, KFF, KFB, COST as solve().  The while condition of the line search is
 *     `learning_rate > low_learning_rate` there (:158): set search_floor = low_learning_rate (the defaults are equal).
 * zero_first_state = 1 is not refused with tracking = 1: it applies ilqr.py's row-0 quirk to the tracking kernels, which is
 * what lets tests/test_gpu_mpc.py reduce them to the pinned ilqr.py path bit for bit; ilqr_mpc.m itself never zeroes row 0
 * and the python class ilqr_amd.ilqr_mpc always passes 0. */
int ilqr_ws_mpc_shift(ilqr_ws* ws, int32_t span);
int ilqr_ws_mpc_solve(ilqr_ws* ws, int32_t first_call);
int ilqr_ws_sync(ilqr_ws* ws);
/* Non-blocking state of the workspace's stream (a solve uses that ONE stream and no events of its own since ABI 8):
 * 0 = every pass enqueued so far has completed, 1 = work still pending, < 0 = the stream reports an error (ilqr_last_error).
 * No counterpart in the reference; a diagnosis aid -- a watchdog can tell "still computing" from "stalled with nothing
 * running" without blocking in ilqr_ws_sync (bench.py prints it when ILQR_BENCH_WATCHDOG expires). */
int ilqr_ws_query(ilqr_ws* ws);
/* kernel timing with HIP events on the workspace stream.  kernel ids: 0 rollout, 1 backward, 2 linesearch, 3 accept,
 * 4 ltvlqr, 5 forward, 6 cost, 7 linearize.  In the split pipeline backward = k_riccati, accept = k_save_boundaries + k_accept_lin (which also writes A_t, B_t); linesearch = all rounds of a pass.
 * ms[i] = summed duration, launches[i] = count since the last enable.  enable: 0 off, 1 every kernel, 2 + id only that kernel
 * (every timed pass puts two event records -- two barrier packets -- on the stream: timing all passes of the headline solve
 * costs it 1 %, timing one kernel a fifth of that). */
int ilqr_ws_profile(ilqr_ws* ws, int32_t enable);
int ilqr_ws_profile_read(ilqr_ws* ws, double* ms /*8*/, int64_t* launches /*8*/);
int ilqr_ws_info(ilqr_ws* ws, int32_t* n_alpha_parallel, int32_t* n_trials, int64_t* hbm_bytes);
/* the line-search schedule of this workspace: n_rounds launches of k_linesearch per search, round r evaluating widths[r]
 * consecutive trials per problem (widths: room for 64 entries).  Fixed width n_alpha_parallel, or 4,4,8,16,... when auto
 * on a large batch; results do not depend on it. */
int ilqr_ws_ls_schedule(ilqr_ws* ws, int32_t* n_rounds, int32_t* widths);
int ilqr_ws_pipeline(ilqr_ws* ws);         /* 1 fused, 2 split: the schedule solve() uses for this workspace */

#ifdef __cplusplus
}
#endif
#endif
