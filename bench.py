#!/usr/bin/env python3
"""bench.py -- batched iLQR solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: ``solve()`` (rollout, then n_iterations x {backward pass,
concurrent-alpha line search, accept re-roll}; python/ilqr.py:159-211) of B acrobot swing-up problems per GPU,
BASELINE.json configs[1] / SURVEY.md section 8d C2: n=4, m=1, N=3000, B=4096 random initial states, 10 iterations,
reference line search and stop tests active.  Inputs are resident in HBM before the timed region starts.
With N GPUs every rank solves its own 4096 problems (weak scaling, no data-path collective) and the per-problem
{cost, iterations, status} records are all-gathered over RCCL inside the timed region.

Prints ONE JSON line on rank 0 with the contract keys plus ``roofline`` (dominant kernel, HIP-event timed on the
library's stream) and ``cpu_baseline`` (the CPU oracle, OpenMP, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks used for the roofline fractions (DESIGN.md section "Roofline")
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s (MI355X_MICROARCH.md, chip-level parameters)
FP64_PEAK_TFLOPS = 78.6        # FP64 vector = FP64 matrix peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz

# algorithmic work per problem per time step
def per_step_work(n, m, f_ops, ab_ops):
    return dict(
        bwd_bytes=8 * (n + m) + 8 * (m * n + m),
        fwd_round_bytes=8 * (n + 2 * m + m * n),
        accept_bytes=8 * (n + 2 * m + m * n) + 8 * (n + m),
        bwd_flops=4 * n**3 + 10 * n * n * m + 6 * n * m * m + 2 * n * n + 8 * n * m + 2 * m**3 + 8 * m * m + ab_ops,
        fwd_flops=n + 2 * m * n + 3 * m + 2 * m * m + 2 * m + f_ops)

SYSTEM_OPS = {'pendulum': (12, 8), 'cartpole': (31, 63), 'acrobot': (93, 189), 'quadcopter': (134, 164)}


def acrobot_c2(B, seed):
    """SURVEY.md 8d C2 (example_acrobot_swingup.py:14-48 + random initial states); problem 0 of seed 0 is the example."""
    dt, N = 1e-3, np.arange(0, 3, 1e-3).shape[0]
    rng = np.random.default_rng(seed)
    init = np.array([-np.pi / 2, 0, 0, 0]) + rng.uniform(-0.2, 0.2, (B, 4))
    guess = 0.5 * rng.standard_normal((B, N, 1))
    if seed == 0:
        np.random.seed(1)
        guess[0] = 0.5 * np.random.normal(loc=0, scale=1, size=(N, 1))
        init[0] = [-np.pi / 2, 0, 0, 0]
    return dict(system='acrobot', dt=dt, N=N, init=init, guess=guess, target=np.array([np.pi / 2, 0, 0, 0]),
                Qk=np.zeros((4, 4)), Rk=0.01 * np.eye(1), QT=np.diag([1000.0, 1000.0, 10.0, 10.0]),
                params=np.array([1, 1, 2, 1, 9.8]))


def cpu_baseline(cfg, n_iterations, sample, threads):
    """The CPU oracle (plain-C restatement of the reference, OpenMP over problems) on the first `sample` problems."""
    from oracle import oracle as O
    O.lib()
    t0 = time.perf_counter()
    r = O.solve_batched(cfg['system'], cfg['init'][:sample], cfg['target'], cfg['guess'][:sample], cfg['Qk'], cfg['Rk'],
                        cfg['QT'], cfg['dt'], cfg['params'], n_iterations, n_threads=threads, want_trace=False)
    dt = time.perf_counter() - t0
    return sample / dt, dt, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=4096, help='problems per GPU')
    ap.add_argument('--iterations', type=int, default=10)
    ap.add_argument('--alphas', type=int, default=0, help='concurrent line-search trials per problem (0 = auto)')
    ap.add_argument('--cpu-sample', type=int, default=4096, help='problems of the batch the CPU baseline solves')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--streams', type=int, default=0, help='independent sub-batches of solve(), one HIP stream each (0 = library default)')
    ap.add_argument('--riccati', type=int, default=0, help='backward recursion mapping: 0 auto, 1 lane per problem, 2 lane group')
    ap.add_argument('--force-dist', action='store_true', help='take the torch.distributed (RCCL) path even with one rank')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world

    # stdout carries exactly ONE JSON line: everything libraries print there (RCCL's version banner, ...) goes to stderr
    json_out = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)

    import ilqr_amd
    from ilqr_amd import _lib
    from ilqr_amd.sharding import all_gather_summary, pack_summary

    dist = torch = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    cfg = acrobot_c2(args.batch, seed=rank)
    B, N = args.batch, cfg['N']
    ws = ilqr_amd.Workspace(cfg['system'], N, B, cfg['dt'], want_trace=True, n_iterations=args.iterations,
                            n_alpha_parallel=args.alphas, device=local_rank, n_streams=args.streams, riccati=args.riccati)
    ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
    ws.upload(_lib.F_INIT, cfg['init'])
    ws.upload(_lib.F_TARGET, cfg['target'])
    ws.upload(_lib.F_PARAMS, cfg['params'])
    ws.upload(_lib.F_GUESS, cfg['guess'])
    ws.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ws.sync()

    def step():
        ws.solve()
        if dist is not None:
            ws.sync()
            rec = pack_summary(ws.download(_lib.F_COST), ws.download(_lib.F_ITERS), ws.download(_lib.F_STATUS))
            return all_gather_summary(rec, B * world, device=torch.device('cuda', local_rank))
        return None

    for _ in range(args.warmup):
        step()
    barrier()
    ws.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ws.profile_read()
    ws.profile(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- algorithmic work of one step, from the per-problem trace the kernels keep
    info = ws.info()
    W = info['n_alpha_parallel']
    trace = ws.download(_lib.F_TRACE)                       # (B, it, 6)
    status = ws.download(_lib.F_STATUS)
    iters = ws.download(_lib.F_ITERS)
    rounds = ws.download(_lib.F_LS_ROUNDS)
    accepts = ws.download(_lib.F_ACCEPTS)
    accj = trace[:, :, 4]
    last_it = np.arange(args.iterations)[None] == iters[:, None] - 1
    ran_ls = np.isfinite(trace[:, :, 1]) & ~(((status[:, None] == 1) | (status[:, None] == 3)) & last_it)
    ref_trials = np.where(accj >= 0, accj + 1, info['n_trials']) * ran_ls   # trials the reference's sequential search runs
    n, m = ws.n, ws.m
    work = per_step_work(n, m, *SYSTEM_OPS[cfg['system']])
    tot = dict(bwd=int(iters.sum()), rounds=int(rounds.sum()), accepts=int(accepts.sum()), trials=int(ref_trials.sum()))
    kern = {}
    for name, bytes_, flops_ in (('backward', tot['bwd'] * N * work['bwd_bytes'], tot['bwd'] * N * work['bwd_flops']),
                                 ('linesearch', tot['rounds'] * N * work['fwd_round_bytes'], tot['trials'] * N * work['fwd_flops']),
                                 ('accept', tot['accepts'] * N * work['accept_bytes'], tot['accepts'] * N * work['fwd_flops']),
                                 ('linearize', 0, 0), ('rollout', 0, 0)):
        ms, passes = prof[name]
        if passes == 0:
            continue
        # one HIP-event pair brackets a pass; a line-search pass is ceil(n_trials / W) launches of k_linesearch (one
        # per round of W trials, later rounds on the compacted list) -- count launches the way rocprofv3 does
        cnt = passes * (-(-info['n_trials'] // W) if name == 'linesearch' else 1)
        per_launch_ms = ms / max(cnt, 1)
        launches_per_step = cnt / max(args.steps, 1)
        b_launch = bytes_ / max(launches_per_step, 1)
        f_launch = flops_ / max(launches_per_step, 1)
        kern[name] = dict(ms_per_launch=per_launch_ms, launches=cnt, passes=passes, ms_per_pass=ms / passes, ms_total=ms,
                          gbs=b_launch / per_launch_ms / 1e6 if per_launch_ms > 0 else 0.0,
                          tflops=f_launch / per_launch_ms / 1e9 if per_launch_ms > 0 else 0.0,
                          bytes_per_launch=b_launch, flops_per_launch=f_launch)
    dominant = max(kern, key=lambda k_: kern[k_]['ms_total'])
    dk = kern[dominant]
    hbm_frac = dk['gbs'] / HBM_PEAK_GBS
    fp_frac = dk['tflops'] / FP64_PEAK_TFLOPS
    if fp_frac >= hbm_frac:
        roofline = dict(bound='mfma', achieved=round(dk['tflops'], 4), peak=FP64_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=round(fp_frac, 5), traffic=None)
    else:
        roofline = dict(bound='hbm', achieved=round(dk['gbs'], 2), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(hbm_frac, 5),
                        traffic=None)
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (collected separately: counters cannot be read
    # from inside this process), only when they were taken on this very configuration
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        c = pmc['config']
        if (c['system'], c['batch'], c['n_timesteps'], c['n_alpha_parallel'], c['pipeline']) == \
                (cfg['system'], B, N, W, info['pipeline']) and args.iterations == 10:
            roofline['traffic'] = round(pmc['kernels']['k_' + dominant]['hbm_bytes_per_launch'] / 1e9, 3)
            roofline['traffic_unit'] = 'GB per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)'
            roofline['algorithmic_bytes_per_launch_GB'] = round(dk['bytes_per_launch'] / 1e9, 3)
    except (OSError, KeyError, ValueError):
        pass
    roofline.update(kernel='k_' + dominant, ms_per_launch=round(dk['ms_per_launch'], 4), launches=dk['launches'],
                    hbm_gbs=round(dk['gbs'], 2), hbm_frac=round(hbm_frac, 5), fp64_tflops=round(dk['tflops'], 4),
                    fp64_frac=round(fp_frac, 5), n_alpha_parallel=W,
                    note='peak = FP64 vector rate (= FP64 MFMA rate on MI355X); MFMA itself is not used, see DESIGN.md')

    total_solves = B * world * args.steps
    value = total_solves / elapsed
    # one backward + line search + accept pass, from the wall clock of the step (the accept step of iteration i and the
    # recursion of iteration i + 1 run overlapped on two streams: their per-kernel times below overlap, the sum would
    # count that time twice): (step - initial rollout - initial linearisation) / iterations
    once_ms = sum(kern[k_]['ms_per_pass'] for k_ in ('rollout', 'linearize') if k_ in kern)
    pass_ms = (1e3 * elapsed / args.steps - once_ms) / max(args.iterations, 1)
    out = {
        'metric': 'iLQR solves/sec (fixed iters, batched)', 'value': round(value, 2), 'unit': 'solves/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'acrobot swing-up iLQR (BASELINE configs[1]): n=4 m=1 N=%d, B=%d problems per GPU, '
                               '%d iterations, reference line search + stop tests active' % (N, B, args.iterations),
                   'problems_per_gpu': B, 'n_timesteps': N, 'n_iterations': args.iterations, 'n_alpha_parallel': W, 'pipeline': info['pipeline'],
                   'sharding': 'by problem index, %d rank(s), all-gather of {cost,iters,status} only' % world},
        'ms_per_bwd_fwd_pass': round(pass_ms, 4),
        'roofline': roofline,
        'kernels': {k_: {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in kv.items()} for k_, kv in kern.items()},
        'exit_status_counts': {_lib.STATUS_NAMES[s]: int((status == s).sum()) for s in range(4)},
        'mean_ref_trials_per_linesearch': round(float(ref_trials.sum()) / max(int(ran_ls.sum()), 1), 2),
    }
    # sanity: problem 0 is the reference's own example; compare its first iterations with the committed golden trace
    gpath = os.path.join(ROOT, 'tests', 'golden', 'acrobot_solve.npz')
    if os.path.exists(gpath) and cfg['system'] == 'acrobot':
        g = np.load(gpath)
        k_it = min(args.iterations, g['trace'].shape[0])
        out['check_problem0_cost_rel_err'] = float(np.max(np.abs(trace[0, :k_it, 0] - g['trace'][:k_it, 0]) / np.abs(g['trace'][:k_it, 0])))
    if not args.no_cpu:
        threads = min(16, os.cpu_count() or 1)
        sample = min(args.cpu_sample, B)
        v, secs, r = cpu_baseline(cfg, args.iterations, sample, threads)
        cost = ws.download(_lib.F_COST)
        out['cpu_baseline'] = {'value': round(v, 3), 'unit': 'solves/s', 'cores': threads, 'kind': 'port',
                               'sample': 'first %d of the %d problems, same %d iterations, oracle/ilqr_oracle.c with OpenMP, %.1f s'
                                         % (sample, B, args.iterations, secs),
                               'gpu_vs_cpu_cost_max_rel_diff': float(np.max(np.abs(cost[:sample] - r['cost']) / np.abs(r['cost']))),
                               'gpu_vs_cpu_cost_frac_within_1e-8': float(np.mean(np.abs(cost[:sample] - r['cost']) <= 1e-8 * np.abs(r['cost']))),
                               'gpu_vs_cpu_same_exit_status_frac': float(np.mean(status[:sample] == r['status']))}
        one = min(64, B)                                  # the same code on ONE core (SURVEY 8d asks for both figures)
        v1, secs1, _ = cpu_baseline(cfg, args.iterations, one, 1)
        out['cpu_baseline']['single_core'] = {'value': round(v1, 3), 'unit': 'solves/s', 'cores': 1,
                                              'sample': 'first %d problems, %.1f s' % (one, secs1)}
    json_out.write(json.dumps(out) + '\n')
    json_out.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
