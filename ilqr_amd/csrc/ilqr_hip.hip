// ilqr_hip.hip -- batched iLQR / LTV-LQR hot path for AMD MI355X (gfx950, CDNA4), fp64.
//
// What is here (reference = Kongx231/ilqr, python/ilqr.py and python/ltvlqr.py):
//   kernels_common.h    systems, kernel arguments, the per-step Riccati algebra (ilqr.py:84-125)
//   kernels_backward.h  k_backward (ilqr.backwards_pass, ilqr.py:64-135 + the "optimal" stop test :175), the split
//                       schedule's k_linearize / k_riccati / k_riccati_quad (4 lanes per problem) / k_riccati_grp (16 lanes,
//                       also the n = 12 ltvlqr), k_ltvlqr (ltvlqr.py:3-28)
//   kernels_forward.h   k_rollout (ilqr.rollout :30-47 + compute_cost :49-62), k_cost, k_linesearch (the while loop of
//                       solve, :180-202: W trials per problem on W adjacent lanes, one launch per round, first accepted
//                       trial by ballot), k_ls_compact, k_forward (forwards_pass :137-158 and the fused accept :194-199),
//                       k_accept_lin
//   kernels_util.h      k_dynamics (the lambdified f/A/B, symbolic_dynamics.py:23-25,88-90), k_pack / k_unpack
//                       (reference layout (B,T,C) <-> tiled SoA [b/64][T][C][b%64])
//   this file           the workspace (device-resident batch), the pass schedule of solve(), the C ABI (include/ilqr_hip.h)
//
// Design: problems are independent, the horizon is sequential.  One lane owns one problem in
// the backward / forward / ltvlqr kernels, with V_xx, V_x, A, B, Q_* held in VGPRs (n <= 4: no LDS, no
// cross-lane traffic); a wave streams its own contiguous tile of 64 problems so every global access is a
// 512-byte coalesced row.  No MFMA: the products are n x n with n <= 12, fp64, and serially dependent in time.
// The translation unit is compiled with -ffp-contract=off; fused multiply-adds appear only where written
// (__builtin_fma in the small dense products), never inside the generated dynamics.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ilqr_hip.h"

#include "fast_trig.h"

#define ILQR_DYN_FN __device__ __forceinline__
#define ILQR_DYN_CONST static constexpr
#define ILQR_DYN_SINCOS(x, s, c) ilqr_sincos_fast((x), (s), (c))
#define ILQR_DYN_SINCOS_SLOW(x, s, c) do { const ilqr_sc sc_ = ilqr_sincos_slow(x); (s) = sc_.sinv; (c) = sc_.cosv; } while (0)
#define ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c) ilqr_sincos_sum((a), (b), (sa), (ca), (sb), (cb), (s), (c))
// in f_sc, where both pairs are carried along the rollout (TrigCarry: a few ulp of random walk between re-seeds), the
// correction for the rounding of a + b (|e| <= ulp(a + b) / 2, below the pairs' own error) is dropped: 4 instead of 12
// instructions.  ILQR_SUM_SC_EXACT = 1 keeps it.
#ifndef ILQR_SUM_SC_EXACT
#define ILQR_SUM_SC_EXACT 0
#endif
#if ILQR_SUM_SC_EXACT
#define ILQR_DYN_SINCOS_SUM_SC(a, b, sa, ca, sb, cb, s, c) ilqr_sincos_sum((a), (b), (sa), (ca), (sb), (cb), (s), (c))
#else
#define ILQR_DYN_SINCOS_SUM_SC(a, b, sa, ca, sb, cb, s, c) ilqr_sincos_sum_carried((sa), (ca), (sb), (cb), (s), (c))
#endif
// ordered compare: a NaN angle is NOT "big" -- the fast path turns NaN into NaN just as well, and a diverged trial of the
// line search (NaN state for the rest of its horizon) would otherwise drag its whole wave through the library path at
// every step (131 072 cartpoles: 221 -> 209 ms per solve).  Inf counts as big (one step later it is NaN).
#define ILQR_DYN_TRIG_BIG(x) (__builtin_fabs(x) >= 1.0e6)
// Fused multiply-adds inside the generated f (and only there).  Off: measured on the headline config it removes 15 % of the
// line search's instructions but only 3.6 % of its time (4.96 -> 4.78 ms), and it breaks exact cancellations the reference
// has (the quadcopter's hover rollout stays exactly 0 unfused, drifts to 1e-17 fused).  FMAs only where written.
#ifndef ILQR_F_CONTRACT
#define ILQR_F_CONTRACT 0
#endif
#if ILQR_F_CONTRACT
#define ILQR_DYN_F_PROLOGUE _Pragma("clang fp contract(fast)")
#else
#define ILQR_DYN_F_PROLOGUE
#endif
#define ILQR_DYN_TAN(x) tan(x)
#define ILQR_DYN_TAN_SC(x, s, c) ilqr_div_shared((s), (c), ilqr_rcp(c))   // tan from the pair f computes anyway (<= 2 ulp)
#define ILQR_DYN_RCP(d) ilqr_rcp(d)
#define ILQR_DYN_DIV(a, d, rd) ilqr_div_shared((a), (d), (rd))
#include "dynamics/pendulum.h"
#include "dynamics/acrobot.h"
#include "dynamics/cartpole.h"
#include "dynamics/car.h"
#include "dynamics/double_integrator.h"
#include "dynamics/double_pendulum.h"
#include "dynamics/floating_base.h"
// The quadcopter comes last: its dynamics use the LIBRARY's sin/cos/tan.  Its 21-iteration example solve amplifies
// last-bit rounding most: with the fast pair above (0.79 ulp) the final K is 9.2e-9 from the
// reference, with ocml's (as accurate as the glibc the reference runs on) 3.8e-9 -- a 2.6x margin to the 1e-8 bar for
// 10 % of that system's speed (32 768 problems: 180 -> 200 ms per solve).  IEEE divisions instead of the shared
// reciprocal change nothing (same digits) and stay off.
#ifndef ILQR_QUAD_TRIG_LIB
#define ILQR_QUAD_TRIG_LIB 1
#endif
#ifndef ILQR_QUAD_DIV_IEEE
#define ILQR_QUAD_DIV_IEEE 0
#endif
#if ILQR_QUAD_TRIG_LIB
#undef ILQR_DYN_SINCOS
#undef ILQR_DYN_SINCOS_SUM
#undef ILQR_DYN_SINCOS_SUM_SC
#undef ILQR_DYN_TRIG_BIG
#undef ILQR_DYN_TAN_SC
// (the library's sincos: one argument reduction for the pair, the same two kernels -- the same bits as sin() and cos())
#define ILQR_DYN_SINCOS(x, s, c) do { double s_, c_; sincos((x), &s_, &c_); (s) = s_; (c) = c_; } while (0)
#define ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c) do { double s_, c_; sincos((a) + (b), &s_, &c_); (s) = s_; (c) = c_; } while (0)
#define ILQR_DYN_SINCOS_SUM_SC(a, b, sa, ca, sb, cb, s, c) ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c)
#define ILQR_DYN_TRIG_BIG(x) 0
#define ILQR_DYN_TAN_SC(x, s, c) tan(x)
#endif
#if ILQR_QUAD_DIV_IEEE
#undef ILQR_DYN_RCP
#undef ILQR_DYN_DIV
#define ILQR_DYN_RCP(d) 0.0
#define ILQR_DYN_DIV(a, d, rd) ((a) / (d))
#endif
#include "dynamics/quadcopter.h"

namespace {

#include "kernels_common.h"
#include "kernels_backward.h"
#include "kernels_forward.h"
#include "kernels_util.h"

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
#ifndef ILQR_CKPT_INTERVAL
#define ILQR_CKPT_INTERVAL 64
#endif
constexpr int CKPT_INTERVAL = ILQR_CKPT_INTERVAL;   // time steps between line-search checkpoints (split pipeline)

}  // namespace

#ifndef ILQR_OVERLAP_CHUNKS
#define ILQR_OVERLAP_CHUNKS 10
#endif
constexpr int OVERLAP_CHUNKS = ILQR_OVERLAP_CHUNKS;   // at most; a chunk is at least 4 segments of the horizon
struct ilqr_ws {
    ilqr_desc d;
    int n, m, np, N, W, n_trials, want_trace;
    int cost_monotone;                       // R_k, Q_T diagonal and non-negative (set_weights): stage costs are >= 0 term by term
    double alpha_end;                        // the learning rate an exhausted line search ends with (ilqr.py:185,202)
    int ls_rounds_n, ls_width[MAX_TRIALS];   // line-search schedule: widths of the rounds (sum >= n_trials)
    long long B, Bpad, tiles;
    hipStream_t stream;
    int n_sub;                       // sub-batches of solve(), each a contiguous tile range on its own stream
    hipStream_t sub_stream[4];
    hipEvent_t sub_start, sub_done[4];
    long long sub_tile0[5];
    double *XB, *CKPT, *LIN;
    int split, grp, quad, overlap, n_chunks, ck, n_seg, n_lin;
    double* VST;
    double *X, *U, *G, *KFB, *KFF, *XO, *UO, *INIT, *TARGET, *PARAMS, *WTS, *alphas, *alpha_pp, *cost, *new_cost, *dJ, *trace;
    double *XD, *UD;                         // tracking cost (ilqr_desc.tracking): per-step targets of the horizon window
    int *active, *iters, *status, *acc_j, *ls_rounds, *accepts, *ls_list, *ls_count;
    void* staging;
    size_t staging_bytes;
    size_t hbm_bytes;
    int state_rows;
    bool profiling;
    int prof_only;                           // -1: every kernel id is timed; else only this one (ilqr_ws_profile)
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
    a.cost_monotone = ws->cost_monotone;
    a.quu_reg = ws->d.quu_reg; a.low_lr = ws->d.low_learning_rate; a.alpha_end = ws->alpha_end;
    a.XB = ws->XB; a.CKPT = ws->CKPT; a.LIN = ws->LIN; a.ck = ws->ck; a.n_seg = ws->n_seg;
    a.XD = ws->XD; a.UD = ws->UD; a.fwd_to_cost = 0;
    a.X = ws->X; a.U = ws->U; a.G = ws->G; a.KFB = ws->KFB; a.KFF = ws->KFF; a.XO = ws->X; a.UO = ws->U;
    a.INIT = ws->INIT; a.TARGET = ws->TARGET; a.PARAMS = ws->PARAMS; a.WTS = ws->WTS; a.alphas = ws->alphas;
    a.alpha_pp = ws->alpha_pp; a.cost = ws->cost; a.new_cost = ws->new_cost; a.dJ = ws->dJ; a.active = ws->active;
    a.iters = ws->iters; a.status = ws->status; a.acc_j = ws->acc_j; a.ls_rounds = ws->ls_rounds;
    a.ls_round = 0; a.ls_list = ws->ls_list; a.ls_count = ws->ls_count;
    a.seg_lo = 0; a.seg_cnt = ws->n_seg; a.t_lo = 0; a.t_hi = ws->N; a.piped = 0; a.VST = ws->VST;
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
        case 7: return fn(Sys_floating_base{});
    }
    return fail(-2, "unknown system id %d", sys);
}

struct ProfScope {
    ilqr_ws* ws; int kid; hipStream_t st; hipEvent_t a, b; bool on;
    ProfScope(ilqr_ws* w, int k, hipStream_t s) : ws(w), kid(k), st(s), a(nullptr), b(nullptr), on(w && w->profiling && (w->prof_only < 0 || w->prof_only == k))
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
// Returns false when no kernel is compiled for this (system, width, mode): the caller turns that into an error instead of
// searching on with stale costs.
template <class S, int W>
bool launch_ls(hipStream_t st, const KArgs& a, bool ckpt, bool track)
{
    const long long waves = a.n_tiles * W;          // tiles * 64 problems * W lanes / 64
    const int wpb = track ? ls_waves_per_block<S, W, true>() : ls_waves_per_block<S, W, false>();
    const unsigned grid = (unsigned)((waves + wpb - 1) / wpb);
    if constexpr (!lane_recursion_ok<S>) {
        // m > 4: the split schedule only, and widths the staged rollout's LDS records fit (ilqr_ws_create sees to both)
        if constexpr (W >= 16) {
            if (track || !ckpt) return false;
            hipLaunchKernelGGL((k_linesearch<S, W, true>), dim3(grid), dim3(64 * wpb), 0, st, a);
            return true;
        }
        return false;
    }
    else if (track) {
        // tracking cost: fused schedule only, line-search widths 8 / 16 / 32 (ilqr_ws_create maps the automatic choice onto these)
        if constexpr (W == 8 || W == 16 || W == 32) {
            if (ckpt) return false;
            hipLaunchKernelGGL((k_linesearch<S, W, false, true>), dim3(grid), dim3(64 * wpb), 0, st, a);
            return true;
        }
        return false;
    }
    else if (ckpt) hipLaunchKernelGGL((k_linesearch<S, W, true>), dim3(grid), dim3(64 * wpb), 0, st, a);
    else hipLaunchKernelGGL((k_linesearch<S, W, false>), dim3(grid), dim3(64 * wpb), 0, st, a);
    return true;
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
        case ILQR_F_TARGET_STATES: tc = (long long)ws->state_rows * n; *dptr = ws->XD; break;
        case ILQR_F_TARGET_INPUTS: tc = N * m; *dptr = ws->UD; break;
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

// ------------------------------------------------------------------------------------------------
// Host <-> device copies of the one-shot entry points.  The caller's arrays are ordinary pageable memory (numpy): a plain
// hipMemcpy to or from them runs at 8-10 GB/s (measured: 1 GB of results of the 4096-problem headline batch took 107 ms
// next to a 48 ms solve).  Large copies therefore go through a small pool of PINNED bounce buffers kept for the life of the
// process (HOST_COPY_CHUNK bytes each, at most HOST_COPY_WORKERS per call): every worker thread owns one buffer and one
// stream and alternates a DMA of its chunk with the memcpy between the buffer and the caller's pages, so the DMAs of some
// workers run under the memcpys (and first-touch page faults) of the others.
// ------------------------------------------------------------------------------------------------
constexpr size_t HOST_COPY_CHUNK = (size_t)8 << 20;
constexpr int HOST_COPY_WORKERS = 8;
constexpr size_t HOST_COPY_DIRECT_BELOW = (size_t)4 << 20;     // smaller copies: one plain hipMemcpy
struct PinnedSlot { void* host; hipStream_t st; int device; bool busy; };
std::mutex g_pinned_mutex;
std::vector<PinnedSlot> g_pinned;

int pinned_acquire(int device, int* index)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (size_t i = 0; i < g_pinned.size(); ++i)
        if (!g_pinned[i].busy && g_pinned[i].device == device) { g_pinned[i].busy = true; *index = (int)i; return 0; }
    PinnedSlot sl{nullptr, nullptr, device, true};
    HIPCHK(hipHostMalloc(&sl.host, HOST_COPY_CHUNK, hipHostMallocDefault));
    hipError_t e = hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipHostFree(sl.host); return fail(-100 - (int)e, "hipStreamCreateWithFlags: %s", hipGetErrorString(e)); }
    g_pinned.push_back(sl);
    *index = (int)g_pinned.size() - 1;
    return 0;
}
PinnedSlot g_pinned_at(int index)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    return g_pinned[index];
}
void pinned_release(int index)
{
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    g_pinned[index].busy = false;
}

struct CopySeg { char* dev; char* host; size_t bytes; };
// every segment copied (to_device: host -> device); the device side must be idle / complete on entry (the caller has
// synchronised the stream that produced or will consume it) and the copies are complete on return
int host_copy(int device, const std::vector<CopySeg>& segs, bool to_device)
{
    struct Chunk { char* dev; char* host; size_t bytes; };
    std::vector<Chunk> chunks;
    for (const CopySeg& sg : segs) {
        if (sg.bytes == 0) continue;
        if (sg.bytes < HOST_COPY_DIRECT_BELOW) {
            HIPCHK(to_device ? hipMemcpy(sg.dev, sg.host, sg.bytes, hipMemcpyHostToDevice) : hipMemcpy(sg.host, sg.dev, sg.bytes, hipMemcpyDeviceToHost));
            continue;
        }
        for (size_t off = 0; off < sg.bytes; off += HOST_COPY_CHUNK)
            chunks.push_back({sg.dev + off, sg.host + off, sg.bytes - off < HOST_COPY_CHUNK ? sg.bytes - off : HOST_COPY_CHUNK});
    }
    if (chunks.empty()) return 0;
    const int T = (int)chunks.size() < HOST_COPY_WORKERS ? (int)chunks.size() : HOST_COPY_WORKERS;
    std::vector<int> rcs(T, 0);
    std::vector<std::string> errs(T);
    std::atomic<size_t> next{0};
    auto body = [&](int w) {
        try {
            if (hipSetDevice(device) != hipSuccess) { rcs[w] = fail(-100, "hipSetDevice(%d) failed in a copy thread", device); errs[w] = g_err; return; }
            int slot = -1;
            if ((rcs[w] = pinned_acquire(device, &slot))) { errs[w] = g_err; return; }
            const PinnedSlot sl = g_pinned_at(slot);
            for (size_t c = next.fetch_add(1); c < chunks.size() && !rcs[w]; c = next.fetch_add(1)) {
                const Chunk& ch = chunks[c];
                hipError_t e;
                if (to_device) {
                    memcpy(sl.host, ch.host, ch.bytes);
                    e = hipMemcpyAsync(ch.dev, sl.host, ch.bytes, hipMemcpyHostToDevice, sl.st);
                    if (e == hipSuccess) e = hipStreamSynchronize(sl.st);
                } else {
                    e = hipMemcpyAsync(sl.host, ch.dev, ch.bytes, hipMemcpyDeviceToHost, sl.st);
                    if (e == hipSuccess) e = hipStreamSynchronize(sl.st);
                    if (e == hipSuccess) memcpy(ch.host, sl.host, ch.bytes);
                }
                if (e != hipSuccess) { rcs[w] = fail(-100 - (int)e, "host copy: %s", hipGetErrorString(e)); errs[w] = g_err; }
            }
            pinned_release(slot);
        } catch (...) {
            rcs[w] = -1; errs[w] = "exception in a copy thread";
        }
    };
    std::vector<std::thread> threads;
    try {
        for (int w = 1; w < T; ++w) threads.emplace_back(body, w);
    } catch (...) { /* fewer helpers: the calling thread and the ones that started drain the chunk list */ }
    body(0);
    for (auto& t : threads) t.join();
    for (int w = 0; w < T; ++w)
        if (rcs[w]) return fail(rcs[w], "%s", errs[w].c_str());
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
        // (the staging buffer is free: every earlier use was followed by a stream synchronisation)
        rc = host_copy(ws->d.device, {{(char*)ws->staging, (char*)const_cast<void*>(src), (size_t)copy_elems * sizeof(double)}}, true);
        if (rc) return rc;
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

// Several fields at once: every field is unpacked (tiled -> (B, TC)) into its own slice of the staging buffer, ONE stream
// synchronisation, then all slices travel together (host_copy).  dst_on_device: unpacked straight into the caller's device
// buffers, nothing else.
struct DlItem { int field; void* dst; };
int do_download_many(ilqr_ws* ws, const DlItem* items, int n_items, bool dst_on_device)
{
    int rc = set_device(ws);
    if (rc) return rc;
    struct Plan { int is_int; long long tc; double* dp; int* ip; size_t bytes, off; };
    if (n_items < 0 || n_items > 8) return fail(-3, "download: at most 8 fields per call");
    Plan plan[8];
    size_t total = 0;
    for (int i = 0; i < n_items; ++i) {
        Plan& pl = plan[i];
        const int field = items[i].field;
        if (!field_elems(ws, field, &pl.is_int, &pl.tc, &pl.dp, &pl.ip)) return fail(-3, "download: unknown field %d", field);
        if (field == ILQR_F_TRACE && !ws->want_trace) return fail(-3, "download: workspace was created without a trace");
        if (!pl.dp && !pl.ip) return fail(-3, "download: field %d not allocated", field);
        pl.bytes = (size_t)ws->B * pl.tc * (pl.is_int ? sizeof(int) : sizeof(double));
        pl.off = total;
        total += (pl.bytes + 255) & ~(size_t)255;
    }
    if (!dst_on_device && (rc = ensure_staging(ws, total))) return rc;
    for (int i = 0; i < n_items; ++i) {
        const Plan& pl = plan[i];
        void* ddst = dst_on_device ? items[i].dst : (void*)((char*)ws->staging + pl.off);
        const long long cnt = ws->Bpad * pl.tc;
        const unsigned grid = (unsigned)((cnt + 255) / 256);
        if (pl.is_int) hipLaunchKernelGGL(k_unpack<int>, dim3(grid), dim3(256), 0, ws->stream, pl.ip, (int*)ddst, ws->B, ws->Bpad, pl.tc);
        else hipLaunchKernelGGL(k_unpack<double>, dim3(grid), dim3(256), 0, ws->stream, pl.dp, (double*)ddst, ws->B, ws->Bpad, pl.tc);
        if ((rc = check_launch("k_unpack"))) return rc;
    }
    if (dst_on_device) return 0;
    HIPCHK(hipStreamSynchronize(ws->stream));
    std::vector<CopySeg> segs;
    for (int i = 0; i < n_items; ++i) segs.push_back({(char*)ws->staging + plan[i].off, (char*)items[i].dst, plan[i].bytes});
    return host_copy(ws->d.device, segs, false);
}
int do_download(ilqr_ws* ws, int field, void* dst, bool dst_on_device)
{
    const DlItem it{field, dst};
    return do_download_many(ws, &it, 1, dst_on_device);
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

int ws_linearize(ilqr_ws* ws, int sub = -1, bool all_problems = false)
{
    const int chunks = (ws->N + LIN_TS - 1) / LIN_TS;
    return run_pass(ws, KID_LINEARIZE, 0, "k_linearize", [&](auto S, const KArgs& a0, hipStream_t st) {
        KArgs a = a0;
        a.all_problems = all_problems ? 1 : 0;
        hipLaunchKernelGGL(k_linearize<decltype(S)>, dim3((unsigned)(a.n_tiles * chunks)), dim3(64), 0, st, a);
    }, sub);
}
int ws_riccati(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_BACKWARD, it, "k_riccati", [&](auto S, const KArgs& a, hipStream_t st) {
        using T = decltype(S);
        if (ws->grp) hipLaunchKernelGGL(k_riccati_grp<T>, dim3((unsigned)(a.n_tiles * Grp<T>::GL)), dim3(64), 0, st, a);
        else if (ws->quad) {
            if constexpr (T::n <= 4) hipLaunchKernelGGL(k_riccati_quad<T>, dim3((unsigned)(a.n_tiles * 4)), dim3(64), 0, st, a);
        }
        else if constexpr (lane_recursion_ok<T>) hipLaunchKernelGGL(k_riccati<T>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    }, sub);
}
int ws_accept_lin(ilqr_ws* ws, int it, int sub = -1)
{
    return run_pass(ws, KID_ACCEPT, it, "k_accept_lin", [&](auto S, const KArgs& a, hipStream_t st) {
        const dim3 grid((unsigned)(a.n_tiles * ws->n_seg));
        hipLaunchKernelGGL(k_save_boundaries, grid, dim3(64), 0, st, a, (int)decltype(S)::n);
        hipLaunchKernelGGL(k_accept_lin<decltype(S)>, grid, dim3(64), 0, st, a);
    }, sub);
}
// accept of iteration `it` and the Riccati recursion of iteration it + 1, overlapped: the accept step is parallel over
// segments of the horizon, the recursion walks the horizon from its end -- so the horizon is cut into chunks, last segments
// first, and launch c runs the accept step of chunk c TOGETHER with the recursion of chunk c + 1 (k_accept_riccati: one
// grid, the recursion's workgroups first; value function handed from chunk to chunk through VST).  Everything is on the
// workspace's ONE stream: the recursion of a chunk follows the accept of that chunk by stream order.  (Rounds 1-2 ran the two
// on two streams joined by events; with a second PROCESS on the same GPU that layout stalled for good in 2 of 41 rehearsals
// -- DESIGN.md section 5.  The dependency graph was sound; a solve now has no cross-queue dependency at all.)  Same
// kernels, same arithmetic, same results; only the first accept chunk and the last recursion chunk run alone.
int ws_accept_riccati_overlapped(ilqr_ws* ws, int it)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    KArgs aa = make_args(ws, it), ar = make_args(ws, it + 1);
    if (aa.n_tiles <= 0) return 0;
    ar.piped = 1;
    hipStream_t A = ws->stream;
    return with_system(ws->d.system_id, [&](auto S) {
        using T = decltype(S);
        if constexpr (T::n <= 4) {
            const int C = ws->n_chunks;
            auto seg_lo = [&](int c) { return (int)((long long)ws->n_seg * c / C); };
            const unsigned n_ric = (unsigned)(ar.n_tiles * 4);
            int rc2 = 0;
            {
                ProfScope pa(ws, KID_ACCEPT, A);                       // the exposed part of the accept step: its first chunk
                hipLaunchKernelGGL(k_save_boundaries, dim3((unsigned)(aa.n_tiles * ws->n_seg)), dim3(64), 0, A, aa, (int)T::n);
                aa.seg_lo = seg_lo(C - 1); aa.seg_cnt = ws->n_seg - aa.seg_lo;
                hipLaunchKernelGGL(k_accept_lin<T>, dim3((unsigned)(aa.n_tiles * aa.seg_cnt)), dim3(64), 0, A, aa);
                if ((rc2 = check_launch("k_accept_lin"))) return rc2;
            }
            ProfScope pb(ws, KID_BACKWARD, A);                         // recursion chunks, each next to the accept chunk below it
            for (int c = C - 1; c >= 0; --c) {
                ar.t_lo = seg_lo(c) * ws->ck;
                ar.t_hi = c + 1 == C ? ws->N : seg_lo(c + 1) * ws->ck;
                if (c > 0) {
                    aa.seg_lo = seg_lo(c - 1); aa.seg_cnt = seg_lo(c) - aa.seg_lo;
                }
                if (c > 0 && aa.seg_cnt > 0) {
                    hipLaunchKernelGGL(k_accept_riccati<T>, dim3(n_ric + (unsigned)(aa.n_tiles * aa.seg_cnt)), dim3(64), 0, A, aa, ar, n_ric);
                    if ((rc2 = check_launch("k_accept_riccati"))) return rc2;
                } else {
                    hipLaunchKernelGGL((k_riccati_quad<T, true>), dim3(n_ric), dim3(64), 0, A, ar);
                    if ((rc2 = check_launch("k_riccati_quad"))) return rc2;
                }
            }
            return 0;
        } else {
            return fail(-1, "overlap needs the four-lane Riccati kernel (n <= 4)");
        }
    });
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
        if constexpr (lane_recursion_ok<decltype(S)>) {
            if (ws->d.tracking) hipLaunchKernelGGL((k_backward<decltype(S), true>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
            else hipLaunchKernelGGL(k_backward<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
        }
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
    const int rounds = ws->ls_rounds_n;
    ProfScope ps(ws, KID_LINESEARCH, st);
    if (rounds > 1) HIPCHK(hipMemsetAsync(cnt, 0, MAX_TRIALS * sizeof(int), st));
    int j0 = 0, unsupported = 0;
    for (int r = 0; r < rounds; ++r) {
        const int Wr = ws->ls_width[r];
        rc = run_pass(ws, KID_LINESEARCH, 0, "k_linesearch", [&](auto S, const KArgs& a0, hipStream_t s) {
            using T = decltype(S);
            KArgs a = a0;
            a.ls_round = r;
            a.ls_j0 = j0;
            a.ls_last = r + 1 == rounds;
            a.ls_count = cnt;
            const bool ck = ws->split != 0, tr = ws->d.tracking != 0;
            bool launched = false;
            switch (Wr) {
                case 1: launched = launch_ls<T, 1>(s, a, ck, tr); break;
                case 2: launched = launch_ls<T, 2>(s, a, ck, tr); break;
                case 4: launched = launch_ls<T, 4>(s, a, ck, tr); break;
                case 8: launched = launch_ls<T, 8>(s, a, ck, tr); break;
                case 16: launched = launch_ls<T, 16>(s, a, ck, tr); break;
                case 32: launched = launch_ls<T, 32>(s, a, ck, tr); break;
                case 64: launched = launch_ls<T, 64>(s, a, ck, tr); break;
                default: break;
            }
            if (!launched) { unsupported = Wr; return; }
            if (!a.ls_last) hipLaunchKernelGGL(k_ls_compact, dim3((unsigned)((a.n_tiles * TILE + 1023) / 1024)), dim3(1024), 0, s, a);
        }, sub, false);
        if (rc) return rc;
        if (unsupported)
            return fail(-1, "line search: no kernel is compiled for %d trials per round with this system%s (tracking cost: 8, 16, 32; "
                        "more than 4 inputs: 16, 32, 64 on the split schedule)", unsupported, ws->d.tracking ? " and the tracking cost" : "");
        j0 += Wr;
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
        if ((rc = ws_linearize(ws, sub))) return rc;                 // A_t, B_t of the initial trajectory (:96-97)
        const bool ov = ws->overlap && sub < 0;
        for (int it = 0; it < ws->d.n_iterations; ++it) {            // :170
            // with overlap the recursion of iteration it > 0 was launched together with the accept of it - 1
            if (!(ov && it > 0) && (rc = ws_riccati(ws, it, sub))) return rc;   // :99-135, :173-179
            if ((rc = ws_linesearch(ws, sub))) return rc;            // :180-202 (+ checkpoints)
            if (ov && it + 1 < ws->d.n_iterations) {
                if ((rc = ws_accept_riccati_overlapped(ws, it))) return rc;
                continue;
            }
            if ((rc = ws_accept_lin(ws, it, sub))) return rc;        // :194-207 in place, segments in parallel
            if (!ACCEPT_FUSES_LINEARIZE && (rc = ws_linearize(ws, sub))) return rc;   // A_t, B_t for the next pass
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
    if (d->overlap < 0 || d->overlap > 2) return fail(-1, "overlap must be 0 (auto), 1 (off) or 2 (on whenever the horizon has two segments)");
    if (d->riccati < 0 || d->riccati > 3) return fail(-1, "riccati must be 0 (auto), 1 (lane per problem), 2 (16-lane group) or 3 (4-lane quad)");
    if (d->pipeline < 0 || d->pipeline > 2) return fail(-1, "pipeline must be 0 (auto), 1 (fused) or 2 (split)");
    if (d->device_wrap < 0 || d->device_wrap > 1) return fail(-1, "device_wrap must be 0 or 1");
    if (d->tracking < 0 || d->tracking > 1) return fail(-1, "tracking must be 0 or 1");
    if (d->tracking && (d->pipeline == 2 || d->riccati > 1 || d->n_streams > 1))
        return fail(-1, "the tracking cost runs on the fused schedule (pipeline 0/1, riccati 0/1, one stream)");
    if (!(d->learning_speed > 0.0 && d->learning_speed < 1.0)) return fail(-1, "learning_speed must be in (0,1)");
    if (!(d->low_learning_rate > 0.0 && d->low_learning_rate < 1.0)) return fail(-1, "low_learning_rate must be in (0,1)");
    if (!(d->search_floor > 0.0 && d->search_floor < 1.0)) return fail(-1, "search_floor must be in (0,1) (the reference's literal is 0.05)");
    if (d->n_devices < 0 || d->n_devices > 64) return fail(-1, "n_devices must be 0 (one device) to 64");
    return 0;
}

// Multi-GPU form of the one-shot host entry points (ilqr_desc.n_devices = G > 1): the batch is cut into G contiguous
// shards by problem index (sizes differ by at most one, remainder to the first shards -- ilqr_amd/sharding.py
// shard_range is the same rule), shard r runs on device `device + r` on its own host thread and reads / writes its own
// slice of the caller's arrays.  No term of ilqr.py / ltvlqr.py couples two problems: there is nothing to exchange.
// fn(sub_descriptor, first_problem) runs one shard through the single-device path.
template <class F>
int run_sharded(const ilqr_desc* d, F&& fn)
{
    int rc = validate_desc(d);
    if (rc) return rc;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(-1, "no HIP device visible");
    const int G = d->n_devices;
    if (!d->device_wrap && (d->device < 0 || d->device + G > ndev))
        return fail(-1, "devices %d..%d requested, %d visible (device_wrap = 1 lets shards share a device)", d->device, d->device + G - 1, ndev);
    const long long B = d->batch, base = B / G, rem = B % G;
    std::vector<int> rcs(G, 0), devs(G, -1);
    std::vector<std::string> errs(G);
    std::vector<std::thread> threads;
    threads.reserve(G);
    for (int r = 0; r < G; ++r) {
        const long long lo = r * base + (r < rem ? r : rem), cnt = base + (r < rem ? 1 : 0);
        if (cnt == 0) continue;                          // more shards than problems
        ilqr_desc sd = *d;
        sd.n_devices = 1;
        sd.batch = cnt;
        sd.device = d->device_wrap ? (((d->device + r) % ndev) + ndev) % ndev : d->device + r;
        try {
            devs[r] = sd.device;
            threads.emplace_back([&, r, lo, sd] {
                try {
                    rcs[r] = fn(sd, lo);
                    if (rcs[r]) errs[r] = g_err;         // g_err is per thread: hand the message to the caller's
                } catch (const std::exception& e) {      // (std::bad_alloc of a host vector, ...): an error code, never std::terminate
                    rcs[r] = -4;
                    errs[r] = std::string("exception in the shard's host thread: ") + e.what();
                } catch (...) {
                    rcs[r] = -4;
                    errs[r] = "unknown exception in the shard's host thread";
                }
            });
        } catch (const std::exception& e) {              // no thread to be had: nothing may leave this function as a C++ exception
            rcs[r] = -1;
            errs[r] = std::string("could not start the shard's host thread: ") + e.what();
            break;
        }
    }
    for (auto& t : threads) t.join();
    for (int r = 0; r < G; ++r)
        if (rcs[r]) return fail(rcs[r], "shard %d of %d (device %d): %s", r, G, devs[r], errs[r].c_str());
    return 0;
}
// p + count elements, keeping an optional (null) argument null
template <class T>
T* shard_ptr(T* p, long long count) { return p ? p + count : nullptr; }

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
    d->overlap = 0;
    d->riccati = 0;
    d->search_floor = 0.05;
    d->n_devices = 0;
    d->device_wrap = 0;
    d->tracking = 0;
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
    return fail(-2, "unknown system '%s' (compiled in: pendulum, acrobot, cartpole, quadcopter, car, double_integrator, double_pendulum, floating_base)", name);
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
    // every exit frees the four buffers
    auto run = [&]() -> int {
        HIPCHK(hipMalloc((void**)&dx, count * n * sizeof(double)));
        HIPCHK(hipMalloc((void**)&du, count * m * sizeof(double)));
        HIPCHK(hipMalloc((void**)&dp, np * sizeof(double)));
        HIPCHK(hipMalloc((void**)&dout, count * osz * sizeof(double)));
        HIPCHK(hipMemcpy(dx, x, count * n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(du, u, count * m * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dp, params, np * sizeof(double), hipMemcpyHostToDevice));
        const unsigned grid = (unsigned)((count + 63) / 64);
        const int rc = with_system(sys, [&](auto S) {
            using T = decltype(S);
            hipLaunchKernelGGL(k_dynamics<T>, dim3(grid), dim3(64), 0, 0, which, (long long)count, dx, du, dt, dp, dout);
            return check_launch("k_dynamics");
        });
        if (rc) return rc;
        HIPCHK(hipMemcpy(out, dout, count * osz * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = run();
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
    ws->prof_only = -1;
    memset(ws->prof_ms, 0, sizeof(ws->prof_ms));
    memset(ws->prof_cnt, 0, sizeof(ws->prof_cnt));
    // alpha table by repeated multiplication (ilqr.py:180,202); the trials are those with alpha > the literal 0.05 of the
    // while condition (:185, ilqr_desc.search_floor), and an exhausted search ends with the first alpha that is not
    double alphas[MAX_TRIALS];
    double lr = 1.0;
    int n_trials = 0;
    for (int j = 0; j < MAX_TRIALS; ++j) {
        alphas[j] = lr;
        if (lr > d->search_floor && n_trials == j) n_trials = j + 1;
        lr = d->learning_speed * lr;
    }
    if (n_trials == MAX_TRIALS) {
        delete ws;
        return fail(-1, "learning_speed/search_floor need %d or more line-search trials (at most %d)", MAX_TRIALS, MAX_TRIALS - 1);
    }
    ws->n_trials = n_trials;
    ws->alpha_end = alphas[n_trials];
    int W = d->n_alpha_parallel;
    if (W == 0) {   // auto: about two waves per SIMD (256 CUs x 4 SIMDs), 16 to 32 trials per problem and round.  The floor
        // is 16 even when the batch fills the chip many times over: every round is a full pass over the horizon, and with
        // the geometric 0.95 schedule most searches need more than 8 trials (131 072 cartpoles: W=8 265 ms, W=16 240 ms).
        W = 16;
        while (W < 32 && ws->Bpad * (long long)(W * 2) / TILE <= 4096) W *= 2;
    }
    if (d->tracking) W = W < 8 ? 8 : (W > 32 ? 32 : W);      // the tracking line search is compiled for 8, 16 and 32 trials per round
    ws->W = W;
    // Round widths.  Fixed (every round W trials) when the caller chose W or the batch is small enough that a round is a
    // latency-bound pass over the horizon (fewer, wider rounds win).  For batches that fill the chip several times over
    // (auto W, >= 32 768 problems) the rounds start narrow and widen -- 4, 4, 8, 16, 16, ... -- so easy searches (pendulum:
    // 4 trials on average) cost 4-8 trials instead of 16 and hard ones (acrobot: 22) no more than before: 131 072 pendulums
    // 1.24 -> 1.9 M solves/s.  Systems whose records are staged through LDS (n > 4: one wave per SIMD, a trial step costs
    // ~550 instructions) start with 8 and go on in SMALL steps -- 8, 2, 2, 4, 8, 16, 32: a search that fails its first
    // round usually ends just behind it (the quadcopter batch accepts trial 8 in every problem from iteration 2 on: 10
    // trials evaluated instead of 16; 8, 8, 16, ... 5.13 ms per search, this 4.23).  Results do not depend on it.
    ws->ls_rounds_n = 0;
    {
        const bool widen = d->n_alpha_parallel == 0 && ws->Bpad >= 32768 && !d->tracking;
        static const int staged_widths[] = {8, 2, 2, 4, 8, 16, 32};
        int covered = 0;
        while (covered < n_trials) {
            int w = W;
            if (widen && ws->n > 4) w = staged_widths[ws->ls_rounds_n < 7 ? ws->ls_rounds_n : 6];
            else if (widen) w = covered < 8 ? 4 : (covered < 16 ? 8 : 16);
            ws->ls_width[ws->ls_rounds_n++] = w;
            covered += w;
        }
        // development aid: ILQR_LS_SCHEDULE="4,4,8,16,32" replaces the widths (powers of two <= 64; the last one repeats)
        if (const char* env = getenv("ILQR_LS_SCHEDULE")) {
            int n_r = 0, cov = 0, last = 0;
            const char* q = env;
            while (cov < n_trials && n_r < MAX_TRIALS) {
                int w = last;
                if (*q) { w = atoi(q); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
                if (w < 1 || w > 64 || (w & (w - 1)) != 0) { n_r = 0; break; }
                ws->ls_width[n_r++] = w; cov += w; last = w;
            }
            if (n_r > 0 && cov >= n_trials) ws->ls_rounds_n = n_r;
        }
        if (ws->m > 4) {   // the staged rollout's records (n + 2m + mn doubles per problem and step) fit LDS from 16 lanes per problem up
            int cov = 0, n_r = 0;
            while (cov < n_trials) {
                const int w = ws->ls_width[n_r < ws->ls_rounds_n ? n_r : ws->ls_rounds_n - 1];
                ws->ls_width[n_r++] = w < 16 ? 16 : w;
                cov += ws->ls_width[n_r - 1];
            }
            ws->ls_rounds_n = n_r;
            if (ws->W < 16) ws->W = 16;
        }
    }
    // pipeline: with few tiles the horizon-sequential kernels occupy one wave per tile and leave most of the 1024 SIMDs
    // idle, so the parallel parts (accept re-roll, Jacobians) are split off; large batches fill the chip either way and
    // keep the fused kernels (no LIN / checkpoint traffic).
    // Crossover measured in round 3 (profiles/r3_pipeline_crossover.txt): acrobot 16 384 problems split 125 k vs fused 117 k
    // solves/s, 32 768 problems split 131 k vs fused 148 k; cartpole and pendulum the same picture (32 768: fused +16 % / +10 %)
    ws->split = d->pipeline == 2 || (d->pipeline == 0 && ws->tiles < 512);
    if (d->n_iterations == 0 || d->tracking) ws->split = 0;
    // Riccati recursion: one lane per problem while V_xx fits its registers (n <= 4), a lane group with LDS hand-offs
    // beyond (always through the split schedule, which provides LIN); riccati = 1 / 2 forces lane / group
    ws->grp = !d->tracking && (d->riccati == 2 || (d->riccati == 0 && ws->n > 4));
    if (ws->m > 4 && !ws->grp) {
        // one lane cannot hold Q_uu, its inverse and K beyond 4 inputs: only the lane-group recursion is compiled for them
        const int trk = d->tracking;
        delete ws;
        return fail(-1, trk ? "the tracking cost (ilqr_mpc) is compiled for systems with at most 4 inputs"
                            : "systems with more than 4 inputs run the lane-group recursion only (riccati = 0 or 2)");
    }
    if (ws->grp) ws->split = 1;
    // four lanes per problem (k_riccati_quad): a shorter instruction chain per step on a chip the lane kernel leaves
    // idle.  Measured on the acrobot (N = 3000): 4096 problems 1.87 -> 1.44 ms, 8192 1.88 -> 1.62, 12 288 1.93 -> 2.15
    // (every lane of a quad loads the whole record), hence up to 128 tiles; riccati = 3 forces it (n <= 4)
    ws->quad = !ws->grp && ws->n <= 4 && (d->riccati == 3 || (d->riccati == 0 && ws->n == 4 && ws->split && ws->tiles <= 128));
    if (ws->quad) ws->split = 1;
    // accept of iteration i overlapped with the recursion of iteration i + 1 (ws_accept_riccati_overlapped): whenever
    // the quad kernel runs the whole batch as one chain; overlap = 1 switches it off
    ws->overlap = ws->quad && d->overlap != 1 && d->n_iterations > 1 && d->n_streams <= 1 && ACCEPT_FUSES_LINEARIZE;
    ws->ck = CKPT_INTERVAL;
    ws->n_seg = (ws->N + ws->ck - 1) / ws->ck;
    // chunks of at least 4 segments (a smaller accept launch does not fill the chip), at most OVERLAP_CHUNKS; measured on
    // the headline config: 3 chunks 69.5 ms per solve (worse than no overlap, 66.1), 4: 64.3, 6: 63.3, 10: 63.0
    // overlap = 2 forces it with chunks of one segment or more (tests)
    const int min_seg = d->overlap == 2 ? 1 : 4;
    ws->n_chunks = ws->n_seg / min_seg < OVERLAP_CHUNKS ? ws->n_seg / min_seg : OVERLAP_CHUNKS;
    if (ws->n_chunks < (d->overlap == 2 ? 2 : 4)) ws->overlap = 0;
    ws->n_lin = with_system(d->system_id, [&](auto S) { return Lin<decltype(S)>::n_lin; });
    const size_t Bp = (size_t)ws->Bpad, n = ws->n, m = ws->m, np = ws->np, N = ws->N;
    const size_t pad = (size_t)LS_GATHER_PAD_STEPS * TILE * 8;          // bytes per component row x padded steps
    struct { void** p; size_t bytes; } allocs[] = {
        // (+ pad: the line search's cooperative record loads run a few time steps ahead of the step they serve and read, never
        // use, up to LS_GATHER_PAD_STEPS rows behind the last tile's last step -- kernels_forward.h closed_loop_gather)
        {(void**)&ws->X, Bp * (N + 1) * n * 8 + pad * n}, {(void**)&ws->U, Bp * N * m * 8 + pad * m}, {(void**)&ws->G, Bp * N * m * 8},
        {(void**)&ws->KFB, Bp * N * m * n * 8 + pad * m * n}, {(void**)&ws->KFF, Bp * N * m * 8 + pad * m}, {(void**)&ws->INIT, Bp * n * 8}, {(void**)&ws->TARGET, Bp * n * 8},
        {(void**)&ws->PARAMS, Bp * np * 8}, {(void**)&ws->WTS, (2 * n * n + m * m) * 8}, {(void**)&ws->alphas, MAX_TRIALS * 8},
        {(void**)&ws->alpha_pp, Bp * 8}, {(void**)&ws->cost, Bp * 8}, {(void**)&ws->new_cost, Bp * 8}, {(void**)&ws->dJ, 3 * Bp * 8},
        {(void**)&ws->active, Bp * 4}, {(void**)&ws->iters, Bp * 4}, {(void**)&ws->status, Bp * 4}, {(void**)&ws->acc_j, Bp * 4},
        {(void**)&ws->ls_rounds, Bp * 4}, {(void**)&ws->accepts, Bp * 4}, {(void**)&ws->ls_list, 2 * Bp * 4},
        {(void**)&ws->ls_count, 5 * MAX_TRIALS * 4},
        {(void**)&ws->trace, ws->want_trace ? Bp * (size_t)d->n_iterations * 6 * 8 : 0},
        {(void**)&ws->XB, ws->split ? Bp * (size_t)ws->n_seg * n * 8 : 0},
        {(void**)&ws->CKPT, ws->split ? Bp * (size_t)ws->n_seg * n * TILE * 8 : 0},
        {(void**)&ws->LIN, ws->split ? Bp * N * (size_t)ws->n_lin * 8 : 0},
        {(void**)&ws->VST, ws->overlap ? Bp * (n * n + n + 4) * 8 : 0},
        {(void**)&ws->XD, d->tracking ? Bp * (N + 1) * n * 8 : 0}, {(void**)&ws->UD, d->tracking ? Bp * N * m * 8 : 0},
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
    // from here on a failure frees the workspace and everything it holds before returning (the caller never sees `ws`)
#define WSCHK(expr)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            const int c_ = fail(-100 - (int)e_, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            ilqr_ws_destroy(ws);                                                                       \
            return c_;                                                                                 \
        }                                                                                              \
    } while (0)
    WSCHK(hipStreamCreate(&ws->stream));
    ws->n_sub = d->n_streams > 0 ? d->n_streams : 1;
    if (ws->n_sub > ws->tiles) ws->n_sub = (int)ws->tiles;
    for (int sb = 0; sb <= ws->n_sub; ++sb) ws->sub_tile0[sb] = ws->tiles * sb / ws->n_sub;
    if (ws->n_sub > 1) {
        WSCHK(hipEventCreateWithFlags(&ws->sub_start, hipEventDisableTiming));
        for (int sb = 0; sb < ws->n_sub; ++sb) {
            WSCHK(hipStreamCreate(&ws->sub_stream[sb]));
            WSCHK(hipEventCreateWithFlags(&ws->sub_done[sb], hipEventDisableTiming));
        }
    }
    WSCHK(hipMemcpy(ws->alphas, alphas, sizeof(alphas), hipMemcpyHostToDevice));
    WSCHK(hipMemsetAsync(ws->WTS, 0, (2 * n * n + m * m) * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->dJ, 0, 3 * Bp * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->KFB, 0, Bp * N * m * n * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->KFF, 0, Bp * N * m * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->X, 0, Bp * (N + 1) * n * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->U, 0, Bp * N * m * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->G, 0, Bp * N * m * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->cost, 0, Bp * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->new_cost, 0, Bp * 8, ws->stream));
    WSCHK(hipMemsetAsync(ws->TARGET, 0, Bp * n * 8, ws->stream));
    *out = ws;
    rc = ilqr_ws_reset(ws);
    if (rc) { ilqr_ws_destroy(ws); *out = nullptr; return rc; }
    {
        hipError_t e_ = hipStreamSynchronize(ws->stream);
        if (e_ != hipSuccess) {
            const int c_ = fail(-100 - (int)e_, "hipStreamSynchronize: %s", hipGetErrorString(e_));
            ilqr_ws_destroy(ws); *out = nullptr;
            return c_;
        }
    }
#undef WSCHK
    return 0;
}

int ilqr_ws_destroy(ilqr_ws* ws)
{
    if (!ws) return 0;
    (void)hipSetDevice(ws->d.device);
    if (ws->stream) (void)hipStreamSynchronize(ws->stream);
    for (auto& ev : ws->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    void* ptrs[] = {ws->XD, ws->UD, ws->VST, ws->XB, ws->CKPT, ws->LIN, ws->X, ws->U, ws->G, ws->KFB, ws->KFF, ws->XO, ws->UO, ws->INIT, ws->TARGET, ws->PARAMS, ws->WTS, ws->alphas,
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
    // compute_cost (ilqr.py:56,60) adds (u'R)u per step and (d'Q_T)d at the end: with diagonal, non-negative R_k and Q_T every
    // term is a rounded non-negative product, so the running cost of a rollout never decreases (k_linesearch uses that)
    auto diag_nonneg = [](const double* M, size_t k) {
        for (size_t i = 0; i < k; ++i)
            for (size_t j = 0; j < k; ++j)
                if (i == j ? !(M[i * k + j] >= 0.0) : M[i * k + j] != 0.0) return false;
        return true;
    };
    ws->cost_monotone = diag_nonneg(w.data() + n * n, m) && diag_nonneg(w.data() + n * n + m * m, n);
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
        if (ws->d.tracking) { if constexpr (lane_recursion_ok<decltype(S)>) hipLaunchKernelGGL((k_cost<decltype(S), true>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a); }
        else hipLaunchKernelGGL(k_cost<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
}
int ilqr_ws_backward(ilqr_ws* ws, int32_t iteration)
{
    if (ws && ws->split) {
        int rc = ws_linearize(ws);
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
        if (ws->d.tracking) { if constexpr (lane_recursion_ok<decltype(S)>) hipLaunchKernelGGL((k_forward<decltype(S), false, true>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a); }
        else hipLaunchKernelGGL((k_forward<decltype(S), false>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
}

// ---- receding-horizon tracking solve: matlab/smooth-ilqr/ilqr_mpc.m ----
int ilqr_ws_mpc_shift(ilqr_ws* ws, int32_t span)
{
    if (!ws) return fail(-1, "null workspace");
    if (!ws->d.tracking) return fail(-1, "ilqr_ws_mpc_shift needs a workspace created with tracking = 1");
    if (span < 0) return fail(-1, "span must be >= 0");
    if (span == 0) return 0;
    const int n = ws->n, m = ws->m;
    return run_pass(ws, KID_FORWARD, 0, "k_mpc_shift", [&](auto, const KArgs& a, hipStream_t st) {
        hipLaunchKernelGGL(k_mpc_shift, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a, n, m, (int)span);
    }, -1, false);
}

int ilqr_ws_mpc_solve(ilqr_ws* ws, int32_t first_call)
{
    if (!ws) return fail(-1, "null workspace");
    if (!ws->d.tracking) return fail(-1, "ilqr_ws_mpc_solve needs a workspace created with tracking = 1");
    int rc;
    if ((rc = ilqr_ws_reset(ws))) return rc;
    if (first_call) {
        // ilqr_mpc.m:68-78: the gain schedule of the first call is one backward pass along the reference window itself
        // (STATES / INPUTS hold it); no stop test, and it does not count as an iteration
        const double low_red = ws->d.low_expected_reduction;
        ws->d.low_expected_reduction = 0.0;
        rc = ws_backward(ws, ws->d.n_iterations);      // trace row out of range: nothing recorded
        ws->d.low_expected_reduction = low_red;
        if (rc) return rc;
        if ((rc = fill_i32(ws, ws->iters, 0))) return rc;
        // (a problem whose Q_uu was exactly singular there stays stopped with status 3, as the reference would have raised)
    }
    // :127-134  closed-loop rollout from the current state around the warm start with learning rate 0, in place; its cost
    // is the current cost
    if ((rc = set_device(ws))) return rc;
    HIPCHK(hipMemsetAsync(ws->alpha_pp, 0, (size_t)ws->Bpad * 8, ws->stream));
    rc = run_pass(ws, KID_FORWARD, 0, "k_forward<tracking>", [&](auto S, const KArgs& a0, hipStream_t st) {
        KArgs a = a0;
        a.XO = ws->X; a.UO = ws->U; a.fwd_to_cost = 1;
        if constexpr (lane_recursion_ok<decltype(S)>) hipLaunchKernelGGL((k_forward<decltype(S), false, true>), dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
    });
    if (rc) return rc;
    for (int it = 0; it < ws->d.n_iterations; ++it) {                // :142-184
        if ((rc = ws_backward(ws, it))) return rc;
        if ((rc = ws_linesearch(ws))) return rc;
        if ((rc = ws_accept(ws, it))) return rc;
    }
    return 0;
}
int ilqr_ws_ltvlqr(ilqr_ws* ws)
{
    if (ws && ws->grp && ws->LIN) {        // n > 4: P does not fit one lane -- Jacobians in parallel, then the lane-group recursion
        int rc = ws_linearize(ws, -1, true);
        if (rc) return rc;
        return run_pass(ws, KID_LTVLQR, 0, "k_riccati_grp<ltv>", [&](auto S, const KArgs& a, hipStream_t st) {
            using T = decltype(S);
            hipLaunchKernelGGL((k_riccati_grp<T, true>), dim3((unsigned)(a.n_tiles * Grp<T>::GL)), dim3(64), 0, st, a);
        });
    }
    return run_pass(ws, KID_LTVLQR, 0, "k_ltvlqr", [&](auto S, const KArgs& a, hipStream_t st) {
        if constexpr (lane_recursion_ok<decltype(S)>) hipLaunchKernelGGL(k_ltvlqr<decltype(S)>, dim3((unsigned)a.n_tiles), dim3(64), 0, st, a);
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

int ilqr_ws_query(ilqr_ws* ws)
{
    if (!ws) return fail(-1, "null workspace");
    int rc = set_device(ws);
    if (rc) return rc;
    const hipError_t e = hipStreamQuery(ws->stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    return fail(-100 - (int)e, "hipStreamQuery: %s", hipGetErrorString(e));
}

int ilqr_ws_profile(ilqr_ws* ws, int32_t enable)
{
    if (!ws) return fail(-1, "null workspace");
    for (auto& ev : ws->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    ws->events.clear();
    memset(ws->prof_ms, 0, sizeof(ws->prof_ms));
    memset(ws->prof_cnt, 0, sizeof(ws->prof_cnt));
    if (enable < 0 || enable > 1 + N_KERNEL_IDS) return fail(-1, "ilqr_ws_profile: enable must be 0 (off), 1 (every kernel) or 2 + kernel id");
    ws->profiling = enable != 0;
    ws->prof_only = enable >= 2 ? enable - 2 : -1;
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

int ilqr_ws_ls_schedule(ilqr_ws* ws, int32_t* n_rounds, int32_t* widths)
{
    if (!ws) return fail(-1, "null workspace");
    if (n_rounds) *n_rounds = ws->ls_rounds_n;
    if (widths) for (int r = 0; r < ws->ls_rounds_n; ++r) widths[r] = ws->ls_width[r];
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
    if (d && d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, np = SYSTEMS[sd.system_id].np, N = sd.n_timesteps;
            return ilqr_rollout_batched(&sd, init_state + lo * n, inputs_in + lo * N * m, sd.params_batched ? params + lo * np : params,
                                        states + lo * (N + 1) * n, inputs + lo * N * m);
        });
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
    if (d && d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, N = sd.n_timesteps;
            return ilqr_cost_batched(&sd, states + lo * (N + 1) * n, inputs + lo * N * m, Rk, QT, sd.target_batched ? target + lo * n : target,
                                     cost + lo);
        });
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
    if (d && d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, np = SYSTEMS[sd.system_id].np, N = sd.n_timesteps;
            return ilqr_backward_batched(&sd, states + lo * (N + 1) * n, inputs + lo * N * m, sd.target_batched ? target + lo * n : target, Qk,
                                         Rk, QT, sd.params_batched ? params + lo * np : params, k + lo * N * m, K + lo * N * m * n,
                                         dJ + lo * 3, shard_ptr(status, lo));
        });
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
    if (d && d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, np = SYSTEMS[sd.system_id].np, N = sd.n_timesteps;
            return ilqr_forward_batched(&sd, init_state + lo * n, ref_states + lo * (N + 1) * n, ref_inputs + lo * N * m, k + lo * N * m,
                                        K + lo * N * m * n, learning_rate + lo, sd.params_batched ? params + lo * np : params,
                                        states + lo * (N + 1) * n, inputs + lo * N * m, Rk, QT,
                                        target && sd.target_batched ? target + lo * n : target, shard_ptr(cost, lo));
        });
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
    if (d && d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, np = SYSTEMS[sd.system_id].np, N = sd.n_timesteps;
            return ilqr_solve_batched(&sd, init_state + lo * n, sd.target_batched ? target + lo * n : target, guess + lo * N * m, Qk, Rk, QT,
                                      sd.params_batched ? params + lo * np : params, shard_ptr(states, lo * (N + 1) * n),
                                      shard_ptr(inputs, lo * N * m), shard_ptr(k, lo * N * m), shard_ptr(K, lo * N * m * n),
                                      shard_ptr(cost, lo), shard_ptr(iters, lo), shard_ptr(status, lo),
                                      shard_ptr(trace, lo * (long long)sd.n_iterations * 6));
        });
    TRY(ilqr_ws_create(d, trace != nullptr, &ws));
    TRY(ilqr_ws_set_weights(ws, Qk, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_INIT, init_state, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_TARGET, target, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_GUESS, guess, 0));
    TRY(ilqr_ws_solve(ws));
    {
        DlItem items[8];
        int cnt = 0;
        if (states) items[cnt++] = {ILQR_F_STATES, states};
        if (inputs) items[cnt++] = {ILQR_F_INPUTS, inputs};
        if (k) items[cnt++] = {ILQR_F_KFF, k};
        if (K) items[cnt++] = {ILQR_F_KFB, K};
        if (cost) items[cnt++] = {ILQR_F_COST, cost};
        if (iters) items[cnt++] = {ILQR_F_ITERS, iters};
        if (status) items[cnt++] = {ILQR_F_STATUS, status};
        if (trace && ws->want_trace) items[cnt++] = {ILQR_F_TRACE, trace};
        TRY(do_download_many(ws, items, cnt, false));
    }
    return ilqr_ws_destroy(ws);
}

int ltvlqr_batched(const ilqr_desc* d, int32_t n_state_rows, const double* states, const double* inputs, const double* Qk,
                   const double* Rk, const double* QT, const double* params, double* K, int32_t* status)
{
    ilqr_ws* ws = nullptr;
    if (!states || !inputs || !Qk || !Rk || !QT || !params || !K) return fail(-1, "null argument");
    if (!d) return fail(-1, "null descriptor");
    if (n_state_rows < d->n_timesteps) return fail(-1, "states needs at least N=%d rows (got %d)", d->n_timesteps, n_state_rows);
    if (d->n_devices > 1)
        return run_sharded(d, [&](const ilqr_desc& sd, long long lo) {
            const long long n = SYSTEMS[sd.system_id].n, m = SYSTEMS[sd.system_id].m, np = SYSTEMS[sd.system_id].np, N = sd.n_timesteps;
            return ltvlqr_batched(&sd, n_state_rows, states + lo * n_state_rows * n, inputs + lo * N * m, Qk, Rk, QT,
                                  sd.params_batched ? params + lo * np : params, K + lo * N * m * n, shard_ptr(status, lo));
        });
    TRY(ilqr_ws_create(d, 0, &ws));
    TRY(ilqr_ws_set_weights(ws, Qk, Rk, QT));
    TRY(ilqr_ws_upload(ws, ILQR_F_PARAMS, params, 0));
    TRY(ilqr_ws_upload(ws, ILQR_F_STATES, states, n_state_rows));
    TRY(ilqr_ws_upload(ws, ILQR_F_INPUTS, inputs, 0));
    TRY(ilqr_ws_ltvlqr(ws));
    TRY(ilqr_ws_download(ws, ILQR_F_KFB, K));
    // R + B'PB exactly singular at some step (np.linalg.inv raises LinAlgError at ltvlqr.py:24): reported per problem when
    // the caller passes `status`, else as an error of the call (the gains are written either way: inf / NaN from that step)
    std::vector<int32_t> st((size_t)d->batch);
    TRY(ilqr_ws_download(ws, ILQR_F_STATUS, st.data()));
    long long bad = -1;
    for (long long b = 0; b < d->batch; ++b) {
        if (status) status[b] = st[b] == ILQR_STATUS_NUMERICAL ? ILQR_STATUS_NUMERICAL : 0;
        if (st[b] == ILQR_STATUS_NUMERICAL && bad < 0) bad = b;
    }
    ilqr_ws_destroy(ws);
    if (bad >= 0 && !status) return fail(ILQR_ERR_SINGULAR, "ltvlqr: R + B'PB is exactly singular (problem %lld): Singular matrix", bad);
    return 0;
}

}  // extern "C"

// -------------------------------------------------------------------------- ltvlqr without allocating
// ltvlqr_device_noalloc: ltvlqr_batched for callers that cannot allocate (include/ilqr_hip.h).  Everything
// ltvlqr_batched keeps in a workspace it hipMallocs -- the tiled copies of the trajectory and of the gains, the
// parameters and weights, the status flags, LIN for n > 4 -- lives in ONE static device array instead.  The array's name
// is a host-side stand-in, not a device address: the address on each GPU comes from hipGetSymbolAddress, once per device
// with that device current, and only that pointer is handed to kernels, memsets and copies.
static __device__ __align__(256) unsigned char scratch[ILQR_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES];     // per device: each GPU holds its own copy of the array

// the address of `scratch` on `device` (which must be the current device), resolved on the first call per device
int scratch_on_device(int device, unsigned char** out)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(-1, "device %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(-100, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return 0;
}
}  // namespace

extern "C" int ltvlqr_device_noalloc(const ilqr_desc* d, int32_t n_state_rows, const double* states, const double* inputs,
                                     const double* Qk, const double* Rk, const double* QT, const double* params, double* K,
                                     int32_t* status, void* stream)
{
    if (!states || !inputs || !Qk || !Rk || !QT || !params || !K) return fail(-1, "null argument");
    int rc = validate_desc(d);
    if (rc) return rc;
    if (d->n_devices > 1) return fail(-1, "ltvlqr_device_noalloc runs on one device (n_devices = %d)", d->n_devices);
    if (n_state_rows < d->n_timesteps) return fail(-1, "states needs at least N=%d rows (got %d)", d->n_timesteps, n_state_rows);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(-1, "device %d out of range (%d visible)", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));
    unsigned char* base = nullptr;
    if ((rc = scratch_on_device(d->device, &base))) return rc;
    // a workspace whose arrays are slices of the scratch array; nothing in it owns memory
    ilqr_ws w{};
    w.d = *d;
    w.n = SYSTEMS[d->system_id].n; w.m = SYSTEMS[d->system_id].m; w.np = SYSTEMS[d->system_id].np;
    w.N = d->n_timesteps; w.B = d->batch;
    w.tiles = (w.B + TILE - 1) / TILE; w.Bpad = w.tiles * TILE;
    w.state_rows = w.N + 1;
    w.stream = (hipStream_t)stream;
    w.n_sub = 1;
    w.ck = CKPT_INTERVAL;
    w.grp = w.n > 4;                                    // as ilqr_ws_create: P of n > 4 does not fit one lane
    w.n_lin = with_system(d->system_id, [&](auto S) { return Lin<decltype(S)>::n_lin; });
    const size_t Bp = (size_t)w.Bpad, n = w.n, m = w.m, np = w.np, N = w.N;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t oX = carve(Bp * (N + 1) * n * 8), oU = carve(Bp * N * m * 8), oK = carve(Bp * N * m * n * 8),
                 oP = carve(Bp * np * 8), oW = carve((2 * n * n + m * m) * 8), oS = carve(Bp * 4),
                 oL = carve(w.grp ? Bp * N * (size_t)w.n_lin * 8 : 0);
    if (off > (size_t)ILQR_NOALLOC_SCRATCH_BYTES) {
        const size_t per_tile = off / (size_t)w.tiles + 1;
        return fail(ILQR_ERR_SCRATCH, "ltvlqr_device_noalloc: %lld problems x %d steps need %zu bytes of scratch, the static array holds %zu "
                    "(about %zu problems at this horizon); use ltvlqr_batched or a workspace for larger batches",
                    (long long)w.B, w.N, off, (size_t)ILQR_NOALLOC_SCRATCH_BYTES, (size_t)ILQR_NOALLOC_SCRATCH_BYTES / per_tile * TILE);
    }
    w.X = (double*)(base + oX); w.U = (double*)(base + oU); w.KFB = (double*)(base + oK); w.PARAMS = (double*)(base + oP);
    w.WTS = (double*)(base + oW); w.status = (int*)(base + oS); w.LIN = w.grp ? (double*)(base + oL) : nullptr;
    w.hbm_bytes = off;
    // the array keeps the last call's contents: the status flags must start at zero (the kernels only ever set them)
    HIPCHK(hipMemsetAsync(w.status, 0, Bp * 4, w.stream));
    HIPCHK(hipMemcpyAsync(w.WTS, Qk, n * n * 8, hipMemcpyDeviceToDevice, w.stream));
    HIPCHK(hipMemcpyAsync(w.WTS + n * n, Rk, m * m * 8, hipMemcpyDeviceToDevice, w.stream));
    HIPCHK(hipMemcpyAsync(w.WTS + n * n + m * m, QT, n * n * 8, hipMemcpyDeviceToDevice, w.stream));
    if ((rc = do_upload(&w, ILQR_F_PARAMS, params, true, 0))) return rc;
    if ((rc = do_upload(&w, ILQR_F_STATES, states, true, n_state_rows))) return rc;
    if ((rc = do_upload(&w, ILQR_F_INPUTS, inputs, true, 0))) return rc;
    if ((rc = ilqr_ws_ltvlqr(&w))) return rc;
    if ((rc = do_download(&w, ILQR_F_KFB, K, true))) return rc;
    if (status && (rc = do_download(&w, ILQR_F_STATUS, status, true))) return rc;
    return 0;
}
