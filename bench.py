#!/usr/bin/env python3
"""bench.py -- batched iLQR solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1          # starts its own 8 ranks (one per GPU) and prints rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the same ranks started by torchrun

A "step" is one pass of the hot path over one batch: ``solve()`` (rollout, then n_iterations x {backward pass,
concurrent-alpha line search, accept re-roll}; python/ilqr.py:159-211) of B acrobot swing-up problems per GPU,
BASELINE.json configs[1] / SURVEY.md section 8d C2: n=4, m=1, N=3000, B=4096 random initial states, 10 iterations,
reference line search and stop tests active.  Inputs are resident in HBM before the timed region starts.
With N GPUs every rank solves its own 4096 problems (weak scaling, no data-path collective) and the per-problem
{cost, iterations, status} records are all-gathered over RCCL inside the timed region.  ``--total-batch T`` instead
splits ONE batch of T problems over the ranks by problem index (strong scaling: BASELINE configs[3], [4]);
``--config`` picks another of BASELINE's configurations as the workload.

Prints ONE JSON line on rank 0 with the contract keys plus ``roofline`` (dominant kernel, HIP-event timed on the
library's stream), ``cpu_baseline`` (the CPU oracle, OpenMP, on a bounded sample of the same workload; one GPU only)
and ``extra_configs`` (the other BASELINE configurations at their per-GPU size, each with its dominant kernel's time
and roofline fractions; one GPU only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# MI355X peaks used for the roofline fractions (DESIGN.md section "Roofline")
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s (MI355X_MICROARCH.md, chip-level parameters)
FP64_PEAK_TFLOPS = 78.6        # FP64 vector = FP64 matrix peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


# algorithmic work per problem per time step
def per_step_work(n, m, f_ops, ab_ops):
    return dict(
        bwd_bytes=8 * (n + m) + 8 * (m * n + m),
        fwd_round_bytes=8 * (n + 2 * m + m * n),
        accept_bytes=8 * (n + 2 * m + m * n) + 8 * (n + m),
        ltv_bytes=8 * (n + m) + 8 * m * n,
        bwd_flops=4 * n**3 + 10 * n * n * m + 6 * n * m * m + 2 * n * n + 8 * n * m + 2 * m**3 + 8 * m * m + ab_ops,
        fwd_flops=n + 2 * m * n + 3 * m + 2 * m * m + 2 * m + f_ops,
        ltv_flops=4 * n**3 + 8 * n * n * m + 4 * n * m * m + 2 * m**3 + ab_ops)


SYSTEM_OPS = {'pendulum': (12, 8), 'cartpole': (31, 63), 'acrobot': (93, 189), 'quadcopter': (134, 164),
              'floating_base': (402, 2964)}     # (operations of the generated f, and of A + B, after CSE)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


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


def _tiled(block, B):
    """B rows made of a block of distinct rows repeated (large batches: the kernels' time does not depend on the values)"""
    reps = -(-B // block.shape[0])
    return np.ascontiguousarray(np.broadcast_to(block[None], (reps,) + block.shape).reshape((-1,) + block.shape[1:])[:B])


def workload(name, B, seed):
    """The synthetic inputs of one BASELINE configuration, B problems (one GPU's share)."""
    if name in ('c2', 'c2_no_stop'):
        cfg = acrobot_c2(B, seed)
        cfg.update(kind='solve', n_iterations=10, W=0, label='acrobot swing-up iLQR (BASELINE configs[1]): n=4 m=1 N=3000')
        if name == 'c2_no_stop':
            cfg.update(no_stop=True, label=cfg['label'] + ', stop tests disabled (every problem runs every pass: the pure per-pass time)')
        return cfg
    if name == 'c3':                                       # ltvlqr along perturbed copies of the reference's own CSV trajectory
        csv, g = golden('acrobot_csv'), golden('acrobot_ltvlqr')
        N, D = 3000, 512
        rng = np.random.default_rng(1 + seed)
        ds, du = 1e-2 * rng.standard_normal((D, N + 1, 4)), 1e-1 * rng.standard_normal((D, N, 1))
        if seed == 0:
            ds[0] = 0.0; du[0] = 0.0
        return dict(kind='ltvlqr', system='acrobot', dt=float(g['dt']), N=N, Qk=g['Qk'], Rk=g['Rk'], QT=g['QT'], params=g['params'],
                    states=_tiled(csv['states'][None] + ds, B), inputs=_tiled(csv['inputs'][None] + du, B),
                    label='ltvlqr gain schedules, acrobot (BASELINE configs[2]): n=4 m=1 N=3000, perturbed copies of acrobot_swingup_*.csv')
    if name in ('c4', 'c4_w8'):                             # cartpole swing-up, configs[3]: 8 concurrent alphas as stated, or auto
        g = golden('cartpole_solve')
        N = g['guess'].shape[0]
        rng = np.random.default_rng(2 + seed)
        return dict(kind='solve', system='cartpole', dt=float(g['dt']), N=N, init=rng.uniform(-0.1, 0.1, (B, 4)), target=g['target'],
                    guess=_tiled(0.05 * rng.standard_normal((512, N, 1)), B), Qk=g['Qk'], Rk=g['Rk'], QT=g['QT'], params=g['params'],
                    n_iterations=10, W=8 if name == 'c4_w8' else 0,
                    label='cartpole swing-up iLQR (BASELINE configs[3], one GPU of 8): n=4 m=1 N=%d, %s' %
                          (N, '8 concurrent alphas' if name == 'c4_w8' else 'auto line-search widths'))
    if name == 'c5':                                       # quadcopter, configs[4]: per-problem targets
        g = golden('quadcopter_solve')
        N = g['guess'].shape[0]
        rng = np.random.default_rng(3 + seed)
        target = np.tile(g['target'], (B, 1))
        target[:, :3] += rng.uniform(-0.5, 0.5, (B, 3))
        return dict(kind='solve', system='quadcopter', dt=float(g['dt']), N=N, init=np.zeros((B, 12)), target=target,
                    guess=_tiled(g['guess'][None], B), Qk=g['Qk'], Rk=g['Rk'], QT=g['QT'], params=g['params'], n_iterations=10, W=0,
                    label='quadcopter iLQR (BASELINE configs[4], one GPU of 8): n=12 m=4 N=%d' % N)
    if name == 'floating_base':                            # 12 states, 12 inputs (synthetic stance change, see
        g = golden('floating_base_solve')                  # tools/make_golden.py cfg_floating_base), per-problem targets
        N = g['guess'].shape[0]
        rng = np.random.default_rng(9 + seed)
        target = np.tile(g['target'], (B, 1))
        target[:, :6] += rng.uniform(-0.05, 0.05, (B, 6))
        return dict(kind='solve', system='floating_base', dt=float(g['dt']), N=N, init=np.zeros((B, 12)), target=target,
                    guess=_tiled(g['guess'][None], B), Qk=g['Qk'], Rk=g['Rk'], QT=g['QT'], params=g['params'], n_iterations=10, W=0,
                    label='floating-base trunk iLQR (symbolic_dynamics_floating_base.m, synthetic stance change): n=12 m=12 N=%d' % N)
    if name in ('pendulum', 'acrobot_big'):                # the north star's large batches of the two example problems
        system = 'pendulum' if name == 'pendulum' else 'acrobot'
        g = golden(system + '_solve')
        N, n = g['guess'].shape[0], g['init'].shape[0]
        rng = np.random.default_rng(5 + seed)
        blk = g['guess'][None] * (1.0 + 0.1 * rng.standard_normal((1024, 1, 1))) + 0.01 * rng.standard_normal((1024, N, 1))
        return dict(kind='solve', system=system, dt=float(g['dt']), N=N, init=g['init'][None] + rng.uniform(-0.2, 0.2, (B, n)),
                    target=g['target'], guess=_tiled(blk, B), Qk=g['Qk'], Rk=g['Rk'], QT=g['QT'], params=g['params'], n_iterations=10,
                    W=0, label='%s swing-up iLQR (the reference example, random initial states): N=%d' % (system, N))
    if name == 'mpc':                                      # receding-horizon tracking, example_acrobot_ilqr_mpc.m:5-28
        csv, g = golden('acrobot_csv'), golden('acrobot_solve')
        H = 100
        rng = np.random.default_rng(7 + seed)
        xs = csv['states'].copy()
        xs[0] = g['init']
        start = 1500                                       # a window in the middle of the swing-up
        return dict(kind='mpc', system='acrobot', dt=float(g['dt']), N=H, n_iterations=10, W=0, params=g['params'],
                    Qk=0.001 * np.eye(4), Rk=0.01 * np.eye(1), QT=1000.0 * np.eye(4),
                    x_des=_tiled(xs[None, start:start + H + 1], B), u_des=_tiled(csv['inputs'][None, start:start + H], B),
                    init=xs[start] + 0.1 * rng.uniform(-1, 1, (B, 4)),
                    label='receding-horizon tracking iLQR (matlab ilqr_mpc, example_acrobot_ilqr_mpc.m): acrobot n=4 m=1, horizon %d, '
                          'warm-started solve_ilqr calls' % H)
    raise ValueError('unknown config %r' % name)


DEFAULT_BATCH = {'c2': 4096, 'c2_no_stop': 4096, 'c3': 65536, 'c4': 131072, 'c4_w8': 131072, 'c5': 32768, 'pendulum': 131072, 'acrobot_big': 65536,
                 'mpc': 65536, 'floating_base': 16384}


def cpu_baseline(cfg, n_iterations, sample, threads):
    """The CPU oracle (plain-C restatement of the reference, OpenMP over problems) on the first `sample` problems."""
    from oracle import oracle as O
    O.lib()
    t0 = time.perf_counter()
    r = O.solve_batched(cfg['system'], cfg['init'][:sample], cfg['target'], cfg['guess'][:sample], cfg['Qk'], cfg['Rk'],
                        cfg['QT'], cfg['dt'], cfg['params'], n_iterations, n_threads=threads, want_trace=False)
    dt = time.perf_counter() - t0
    return sample / dt, dt, r


def source_hash():
    """hash of the kernel sources: ties profiles/pmc_traffic.json to the code it was measured on"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'ilqr_amd', 'csrc')
    for dirpath, _, files in sorted(os.walk(src)):
        for f in sorted(files):
            if f.endswith(('.h', '.hip')):
                h.update(f.encode())
                h.update(open(os.path.join(dirpath, f), 'rb').read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(n, argv, rank0_stdout=None, extra_env=None, poll_s=0.2, progress_s=30.0):
    """Start `argv` n times, one process per rank, with the environment torch.distributed.run would give them
    (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port).  Rank 0 writes to
    `rank0_stdout` (default: this process's stdout), the other ranks' stdout goes to stderr.  Returns the exit codes; if a
    rank fails the others (which would wait for it in the rendezvous forever) are terminated -- by PID, these children only.
    The caller must not have touched the GPU: the children are fresh processes, each initialising its own device.
    A line goes to stderr every `progress_s` seconds while the ranks run (the first `import torch` on a fresh machine takes
    minutes; a silent launcher looks hung)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.update(extra_env or {})
        out = (rank0_stdout if rank0_stdout is not None else None) if r == 0 else sys.stderr
        procs.append(subprocess.Popen(argv, env=env, stdout=out))
    codes = [None] * n
    t0 = t_last = time.time()
    while any(c is None for c in codes):
        if progress_s and time.time() - t_last >= progress_s:
            t_last = time.time()
            print('bench.py launcher: %d of %d ranks still running after %.0f s' % (sum(c is None for c in codes), n, t_last - t0),
                  file=sys.stderr, flush=True)
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(poll_s)
    return codes


# ------------------------------------------------------------------------------------------------ one workload on one GPU
def make_workspace(cfg, B, device, args, want_trace=True):
    import ilqr_amd
    from ilqr_amd import _lib
    if cfg['kind'] == 'mpc':
        ws = ilqr_amd.Workspace(cfg['system'], cfg['N'], B, cfg['dt'], want_trace=want_trace, n_iterations=cfg['n_iterations'],
                                n_alpha_parallel=args.alphas, device=device, zero_first_state=False, tracking=True)
        ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
        ws.upload(_lib.F_PARAMS, cfg['params'])
        ws.upload(_lib.F_INIT, cfg['init'])
        ws.upload(_lib.F_TARGET_STATES, cfg['x_des'])
        ws.upload(_lib.F_TARGET_INPUTS, cfg['u_des'])
        ws.upload(_lib.F_STATES, cfg['x_des'])              # first call: the reference window is the warm start (ilqr_mpc.m:73-74)
        ws.upload(_lib.F_INPUTS, cfg['u_des'])
        ws.mpc_solve(True)
    elif cfg['kind'] == 'ltvlqr':
        ws = ilqr_amd.Workspace(cfg['system'], cfg['N'], B, cfg['dt'], device=device)
        ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
        ws.upload(_lib.F_PARAMS, cfg['params'])
        ws.upload(_lib.F_STATES, cfg['states'])
        ws.upload(_lib.F_INPUTS, cfg['inputs'])
    else:
        W = args.alphas if args.alphas else cfg['W']
        # no_stop (the second figure): the "optimal" test never fires (threshold 0) and an exhausted search does not
        # end the solve (ilqr.py:204 with a threshold below every trial's rate): every problem runs every pass
        stops = dict(low_expected_reduction=0.0, low_learning_rate=1e-300) if cfg.get('no_stop') else {}
        ws = ilqr_amd.Workspace(cfg['system'], cfg['N'], B, cfg['dt'], want_trace=want_trace, n_iterations=cfg['n_iterations'],
                                n_alpha_parallel=W, device=device, n_streams=args.streams, riccati=args.riccati, pipeline=getattr(args, 'pipeline', 0),
                                target_batched=np.ndim(cfg['target']) == 2,
                                overlap=args.overlap if getattr(args, 'overlap', None) is not None else 0, **stops)
        ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
        ws.upload(_lib.F_INIT, cfg['init'])
        ws.upload(_lib.F_TARGET, cfg['target'])
        ws.upload(_lib.F_PARAMS, cfg['params'])
        ws.upload(_lib.F_GUESS, cfg['guess'])
    ws.sync()
    return ws


def kernel_table(ws, cfg, prof, steps):
    """Per kernel: HIP-event time per pass and per launch, algorithmic bytes / flops per launch (SURVEY.md section 8d
    accounting, from the per-problem counters the kernels keep), achieved GB/s and TFLOP/s."""
    from ilqr_amd import _lib
    info = ws.info()
    n, m, N = ws.n, ws.m, cfg['N']
    work = per_step_work(n, m, *SYSTEM_OPS[cfg['system']])
    if cfg['kind'] == 'mpc':                                 # the tracking cost also reads x_des, u_des at every step
        for key in ('bwd_bytes', 'fwd_round_bytes', 'accept_bytes'):
            work[key] += 8 * (n + m)
    extra = {}
    if cfg['kind'] == 'ltvlqr':
        rows = [('ltvlqr', ws.B * N * work['ltv_bytes'] * steps, ws.B * N * work['ltv_flops'] * steps, 1), ('linearize', 0, 0, 1)]
    else:
        trace = ws.download(_lib.F_TRACE)                       # (B, it, 6)
        status, iters = ws.download(_lib.F_STATUS), ws.download(_lib.F_ITERS)
        rounds, accepts = ws.download(_lib.F_LS_ROUNDS), ws.download(_lib.F_ACCEPTS)
        accj = trace[:, :, 4]
        n_it = cfg['n_iterations']
        last_it = np.arange(n_it)[None] == iters[:, None] - 1
        ran_ls = np.isfinite(trace[:, :, 1]) & ~(((status[:, None] == 1) | (status[:, None] == 3)) & last_it)
        # trials the reference's sequential search runs: first accepted j + 1, or all of them
        ref_trials = np.where(accj >= 0, accj + 1, info['n_trials']) * ran_ls
        # rounds are counted in launches of varying width: bytes = one record per problem, step and round it took part in
        tot = dict(bwd=int(iters.sum()), rounds=int(rounds.sum()), accepts=int(accepts.sum()), trials=int(ref_trials.sum()))
        rows = [('backward', tot['bwd'] * N * work['bwd_bytes'], tot['bwd'] * N * work['bwd_flops'], 1),
                ('linesearch', tot['rounds'] * N * work['fwd_round_bytes'], tot['trials'] * N * work['fwd_flops'], len(info['ls_widths'])),
                ('accept', tot['accepts'] * N * work['accept_bytes'], tot['accepts'] * N * work['fwd_flops'], 1),
                ('linearize', 0, 0, 1), ('rollout', 0, 0, 1)]
        # the device-side counters accumulate over one solve (reset by solve()): scale to the timed steps
        rows = [(k_, b_ * steps, f_ * steps, l_) for k_, b_, f_, l_ in rows]
        # trials the scheme evaluated: a search takes part in rounds until the cumulative width covers what it needed (every
        # lane of a participating group computes, also those beyond the last trial of the schedule)
        cum = np.cumsum(info['ls_widths'])
        need = np.where(accj >= 0, accj + 1, info['n_trials'])
        eval_trials = cum[np.searchsorted(cum, need)] * ran_ls
        extra = dict(status=status, iters=iters, trace=trace, ref_trials=ref_trials, ran_ls=ran_ls, eval_trials=eval_trials)
    kern = {}
    for name, bytes_, flops_, launches_per_pass in rows:
        ms, passes = prof[name]
        if passes == 0:
            continue
        cnt = passes * launches_per_pass                      # the way rocprofv3 counts launches of the kernel
        per_launch_ms = ms / cnt
        kern[name] = dict(ms_per_launch=per_launch_ms, launches=cnt, passes=passes, ms_per_pass=ms / passes, ms_total=ms,
                          gbs=bytes_ / cnt / per_launch_ms / 1e6, tflops=flops_ / cnt / per_launch_ms / 1e9,
                          bytes_per_launch=bytes_ / cnt, flops_per_launch=flops_ / cnt)
    return kern, info, extra


def roofline_of(kern, timed=None):
    # the dominant kernel: the one run_steps found (and timed over the timed steps), else the largest total in the table
    dominant = timed if timed in kern else max(kern, key=lambda k_: kern[k_]['ms_total'])
    dk = kern[dominant]
    hbm_frac, fp_frac = dk['gbs'] / HBM_PEAK_GBS, dk['tflops'] / FP64_PEAK_TFLOPS
    if fp_frac >= hbm_frac:
        # contract vocabulary: "mfma" = the compute roof.  The kernels issue no MFMA instruction: the
        # roof is the FP64 VALU rate, which on MI355X equals the FP64 MFMA rate.
        r = dict(bound='mfma', bound_detail='fp64_valu', achieved=round(dk['tflops'], 4), peak=FP64_PEAK_TFLOPS, unit='TFLOP/s',
                 frac=round(fp_frac, 5), traffic=None)
    else:
        r = dict(bound='hbm', achieved=round(dk['gbs'], 2), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(hbm_frac, 5), traffic=None)
    r.update(kernel='k_' + dominant, ms_per_launch=round(dk['ms_per_launch'], 4), launches=dk['launches'],
             hbm_gbs=round(dk['gbs'], 2), hbm_frac=round(hbm_frac, 5), fp64_tflops=round(dk['tflops'], 4), fp64_frac=round(fp_frac, 5),
             algorithmic_bytes_per_launch_GB=round(dk['bytes_per_launch'] / 1e9, 4),
             algorithmic_gflop_per_launch=round(dk['flops_per_launch'] / 1e9, 4))
    return r, dominant


def pmc_for(name, cfg, B, info):
    """The committed rocprofv3 PMC summary of this configuration (profiles/pmc_traffic_<name>.json, written by
    tools/collect_profiles.sh), or None unless it was taken on this very configuration AND on these very kernel sources."""
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic_%s.json' % name)))
        c = pmc['config']
        same = (c['system'], c['batch'], c['n_timesteps'], c['pipeline'], list(c['ls_widths'])) == \
               (cfg['system'], B, cfg['N'], info['pipeline'], list(info['ls_widths']))
        return pmc if same and pmc.get('source_hash') == source_hash() else None
    except (OSError, KeyError, ValueError, TypeError):
        return None


def add_pmc(roofline, dominant, pmc, extra, work, cfg):
    """roofline.traffic (HBM bytes per launch of the dominant kernel, PMC) and roofline.factors: the three factors the
    FP64 fraction of a line search is the product of -- VALU-busy share of SIMD cycles (PMC), trials the reference needs /
    trials the scheme evaluates (this run's device trace), credited flops / peak flops of the issue slots spent (the
    kernel's VALU instructions per trial step, PMC, against 2 flops per lane and slot)."""
    if pmc is None:
        return
    # bench.py times passes (HIP events around a pass); the kernel behind a pass depends on the schedule
    names = {'backward': ('k_backward', 'k_riccati_grp', 'k_riccati', 'k_accept_riccati', 'k_riccati_quad'), 'accept': ('k_forward', 'k_accept_lin')}
    name = next((c for c in names.get(dominant, ('k_' + dominant,)) if c in pmc['kernels']), None)
    if name is None:
        return
    k = pmc['kernels'][name]
    roofline['kernel_profiled'] = name
    roofline['traffic'] = round(k['hbm_bytes_per_launch'] / 1e9, 3)
    roofline['traffic_unit'] = ('GB per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic_%s.json, kernel sources %s)'
                                % (pmc['config'].get('name', 'c2'), pmc['source_hash']))
    if dominant == 'linesearch' and 'eval_trials' in extra and 'valu_busy_frac_of_simd_cycles' in k:
        ev, need = float(extra['eval_trials'].sum()), float(extra['ref_trials'].sum())
        # VALU instructions per evaluated trial step: instructions per launch x launches per solve / (lanes-worth of trials x N)
        insts = k.get('SQ_INSTS_VALU', 0.0) * k['launches'] * 64.0 / max(ev * cfg['N'], 1.0)
        f = dict(valu_busy=round(k['valu_busy_frac_of_simd_cycles'], 4), trials_needed_over_evaluated=round(need / max(ev, 1.0), 4),
                 valu_insts_per_trial_step=round(insts, 1),
                 credited_flops_over_issue_slots=round(work['fwd_flops'] / (2.0 * insts), 4) if insts else None,
                 trials_needed=need, trials_evaluated=ev)
        if insts:
            f['product'] = round(f['valu_busy'] * f['trials_needed_over_evaluated'] * f['credited_flops_over_issue_slots'], 4)
        f['note'] = ('frac ~ valu_busy x needed/evaluated x credited/slots x (clock under load / 2.4 GHz); valu_busy and the '
                     'instruction count come from the PMC passes of one solve, the trial counts from this run')
        roofline['factors'] = f


def run_steps(ws, cfg, steps, warmup, after_step=None, barrier=None):
    """W untimed + K timed steps; returns (seconds, per-kernel HIP-event profile scaled to the K steps, the kernel timed live)"""
    mpc_rng = np.random.default_rng(99)

    def mpc_call():
        # a later solve_ilqr call: a new measured state (the window's first state plus a fresh disturbance: 32 B per problem
        # over PCIe, the only host data of the call), the warm start moved one step, then the solve
        from ilqr_amd import _lib
        ws.upload(_lib.F_INIT, cfg['x_des'][:, 0] + 0.1 * mpc_rng.uniform(-1, 1, cfg['init'].shape))
        ws.mpc_shift(1)
        ws.mpc_solve(False)

    one = {'ltvlqr': ws.ltvlqr, 'mpc': mpc_call}.get(cfg['kind'], ws.solve)

    def step():
        one()
        return after_step() if after_step else None

    for _ in range(warmup):
        step()
    # Kernel times come from HIP events on the workspace's stream.  Every timed pass puts two event records (barrier packets)
    # on the stream -- for the headline solve, all passes timed, 0.44 ms of 46 -- so ONE untimed step times every kernel
    # (the `kernels` table, and which kernel is the dominant one), and in the timed region only the dominant kernel is timed:
    # its events are the ones `roofline` is computed from.
    (barrier or ws.sync)()
    ws.profile(True)
    step()
    ws.sync()
    once = ws.profile_read()
    dominant = max(once, key=lambda k_: once[k_][0])
    (barrier or ws.sync)()
    ws.profile(True, only=dominant)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    (barrier or ws.sync)()
    elapsed = time.perf_counter() - t0
    timed = ws.profile_read()
    ws.profile(False)
    prof = {k_: (ms * steps, cnt * steps) for k_, (ms, cnt) in once.items()}     # the untimed step's figures, scaled to the timed steps
    prof[dominant] = timed[dominant]
    return elapsed, prof, dominant


def extra_config(name, args):
    """one of the other BASELINE configurations at its per-GPU size, on this GPU: time, dominant kernel, roofline fractions"""
    B = DEFAULT_BATCH[name]
    t_setup = time.perf_counter()
    cfg = workload(name, B, 0)
    ws = make_workspace(cfg, B, 0, args)
    t_setup = time.perf_counter() - t_setup
    steps = 3
    elapsed, prof, timed_kernel = run_steps(ws, cfg, steps, 1)
    kern, info, extra = kernel_table(ws, cfg, prof, steps)
    roof, dominant = roofline_of(kern, timed_kernel)
    add_pmc(roof, dominant, pmc_for(name, cfg, B, info), extra, per_step_work(ws.n, ws.m, *SYSTEM_OPS[cfg['system']]), cfg)
    out = {'workload': '%s, B=%d on one GPU%s' % (cfg['label'], B, '' if cfg['kind'] == 'ltvlqr' else ', %d iterations' % cfg['n_iterations']),
           'value': round(B * steps / elapsed, 1),
           'unit': {'ltvlqr': 'gain schedules/s', 'mpc': 'solve_ilqr calls/s'}.get(cfg['kind'], 'solves/s'),
           'ms_per_step': round(1e3 * elapsed / steps, 3), 'steps': steps, 'pipeline': info['pipeline'], 'ls_widths': info['ls_widths'],
           'hbm_resident_GB': round(info['hbm_bytes'] / 1e9, 2), 'setup_s': round(t_setup, 1), 'roofline': roof,
           'kernels_ms_per_pass': {k_: round(v['ms_per_pass'], 3) for k_, v in kern.items()},
           'kernels_hbm_frac': {k_: round(v['gbs'] / HBM_PEAK_GBS, 4) for k_, v in kern.items() if v['gbs'] > 0},
           'kernels_fp64_frac': {k_: round(v['tflops'] / FP64_PEAK_TFLOPS, 4) for k_, v in kern.items() if v['tflops'] > 0}}
    if cfg.get('no_stop'):
        once = sum(kern[k_]['ms_per_pass'] for k_ in ('rollout', 'linearize') if k_ in kern)
        out['ms_per_bwd_fwd_pass'] = round((1e3 * elapsed / steps - once) / cfg['n_iterations'], 4)
    if cfg['kind'] in ('solve', 'mpc'):
        st = extra['status']
        out['exit_status_counts'] = {n_: int((st == s).sum()) for s, n_ in ((0, 'max_iter'), (1, 'optimal'), (2, 'low_learning_rate'), (3, 'numerical'))}
        out['mean_ref_trials_per_linesearch'] = round(float(extra['ref_trials'].sum()) / max(int(extra['ran_ls'].sum()), 1), 2)
        if 'eval_trials' in extra:
            out['trials_evaluated_over_needed'] = round(float(extra['eval_trials'].sum()) / max(float(extra['ref_trials'].sum()), 1.0), 3)
    ws.close()
    return out


def api_config(name):
    """The drop-in boundary itself under the clock: ilqr_amd.ilqr(...).solve() with host numpy in and host numpy out (workspace
    creation, H2D, solve, D2H, destruction).  c1 = BASELINE configs[0], the pendulum example, one problem
    (example_pendulum_swingup.py:14-47); c2_api = the headline batch through the class (example_acrobot_swingup.py:51-54 with a
    leading batch axis)."""
    import contextlib
    import io
    import ilqr_amd
    if name == 'c1':
        g = golden('pendulum_solve')
        fab = ilqr_amd.symbolic_dynamics_pendulum()
        ts = []
        for _ in range(21):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                s = ilqr_amd.ilqr(g['init'], g['target'], g['guess'], float(g['dt']), float(g['start']), float(g['end']), *fab, g['Qk'],
                                  g['Rk'], g['QT'], g['params'], int(g['n_iterations']))
                out = s.solve()
            ts.append(time.perf_counter() - t0)
        ts = ts[1:]
        return {'workload': 'pendulum swing-up example (BASELINE configs[0]): one problem, n=2 m=1 N=%d, ilqr(...).solve() end to end through '
                            'the drop-in class, host arrays in and out' % g['guess'].shape[0],
                'value': round(1.0 / float(np.median(ts)), 2), 'unit': 'solves/s', 'ms_per_solve_median': round(1e3 * float(np.median(ts)), 3),
                'ms_per_solve_min': round(1e3 * min(ts), 3), 'runs': len(ts), 'iterations': int(s.iterations_),
                'cost_rel_err_vs_reference': float(abs(out[4] - g['cost']) / abs(g['cost'])),
                'reference_python_seconds': 1.11, 'reference_note': 'BASELINE.md: the reference python, one core of the build container'}
    cfg = acrobot_c2(DEFAULT_BATCH['c2'], 0)
    fab = ilqr_amd.symbolic_dynamics_acrobot()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        s = ilqr_amd.ilqr(cfg['init'], cfg['target'], cfg['guess'], cfg['dt'], 0, 3, *fab, cfg['Qk'], cfg['Rk'], cfg['QT'], cfg['params'], 10)
        out = s.solve()
        ts.append(time.perf_counter() - t0)
    ts = ts[1:]
    nbytes = sum(o.nbytes for o in out[:4]) + cfg['guess'].nbytes + cfg['init'].nbytes
    B = DEFAULT_BATCH['c2']
    return {'workload': 'the headline batch through the drop-in class: ilqr(init (B,4), ..., guess (B,N,1)).solve(), B=%d, host numpy in, '
                        'host numpy out (states, inputs, k, K, cost), 10 iterations' % B,
            'value': round(B / float(np.median(ts)), 1), 'unit': 'solves/s', 'ms_per_step': round(1e3 * float(np.median(ts)), 2),
            'ms_min': round(1e3 * min(ts), 2), 'runs': len(ts), 'bytes_across_the_boundary_GB': round(nbytes / 1e9, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c2', choices=sorted(DEFAULT_BATCH), help='workload (BASELINE configuration); default c2, the headline')
    ap.add_argument('--batch', type=int, default=0, help='problems per GPU (weak scaling); 0 = the configuration\'s per-GPU size')
    ap.add_argument('--total-batch', type=int, default=0, help='ONE batch of this many problems split over the ranks by problem index (strong scaling)')
    ap.add_argument('--iterations', type=int, default=10)
    ap.add_argument('--alphas', type=int, default=0, help='concurrent line-search trials per problem (0 = the configuration\'s / auto)')
    ap.add_argument('--cpu-sample', type=int, default=4096, help='problems of the batch the CPU baseline solves')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the extra_configs block (the other BASELINE configurations)')
    ap.add_argument('--streams', type=int, default=0, help='independent sub-batches of solve(), one HIP stream each (0 = library default)')
    ap.add_argument('--overlap', type=int, default=None, choices=(0, 1, 2), help='accept / recursion overlap: 0 auto, 1 off, 2 on '
                    '(default: auto)')
    ap.add_argument('--riccati', type=int, default=0, help='backward recursion mapping: 0 auto, 1 lane per problem, 2 lane group')
    ap.add_argument('--pipeline', type=int, default=0, help='kernel schedule: 0 auto, 1 fused, 2 split (same results)')
    ap.add_argument('--force-dist', action='store_true', help='take the torch.distributed (RCCL) path even with one rank')
    ap.add_argument('--backend', default='nccl', choices=('nccl', 'gloo'), help='torch.distributed backend: nccl (= RCCL over xGMI, the '
                    'real thing) or gloo (rehearsal of the N > 1 path where the ranks cannot each have a GPU)')
    ap.add_argument('--device-wrap', action='store_true', help='rehearsal: rank r uses GPU r modulo the visible GPUs (several ranks per GPU; '
                    'RCCL refuses that, so it needs --backend gloo)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # not under torchrun: this process becomes the launcher (it never touches a GPU) and its children are the ranks
        codes = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad:
            sys.exit('bench.py: rank(s) failed: %s' % ', '.join('rank %d exit %s' % rc for rc in bad))
        return

    rank = int(os.environ.get('RANK', '0'))
    watched = {}
    if os.environ.get('ILQR_BENCH_WATCHDOG'):
        # development aid: if the run has not finished after that many seconds, write every thread's stack and the state of
        # the workspace's stream (ilqr_ws_query: work pending or not -- a solve has no other queue or event to wait on) to a file
        # that is kept (gpurun_out/watchdog_rank<r>_pid<p>.txt), and exit non-zero
        import faulthandler
        import threading

        def expire():
            path = os.path.join(ROOT, 'gpurun_out', 'watchdog_rank%d_pid%d.txt' % (rank, os.getpid()))
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, 'w') as fh:
                    fh.write('bench.py watchdog: rank %d still running after %s s\n' % (rank, os.environ['ILQR_BENCH_WATCHDOG']))
                    ws_ = watched.get('ws')
                    try:
                        fh.write('workspace stream: %s\n' % ({0: 'idle (nothing pending)', 1: 'work pending'}[ws_.query()] if ws_ else 'no workspace yet'))
                    except Exception as e:
                        fh.write('workspace stream query failed: %r\n' % (e,))
                    fh.flush()
                    faulthandler.dump_traceback(file=fh, all_threads=True)
                sys.stderr.write(open(path).read())
            finally:
                os._exit(3)

        timer = threading.Timer(float(os.environ['ILQR_BENCH_WATCHDOG']), expire)
        timer.daemon = True
        timer.start()
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world

    # stdout carries exactly ONE JSON line: everything libraries print there (RCCL's version banner, ...) goes to stderr
    json_out = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)

    dist = torch = None
    if world > 1 or args.force_dist:
        # torch first: a process that loads libilqr_hip.so (system HIP runtime) before torch ends up with two HIP runtimes
        import torch
        import torch.distributed as dist
        if args.device_wrap:
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')

    import ilqr_amd
    from ilqr_amd import _lib
    from ilqr_amd.sharding import all_gather_summary, pack_summary, shard_range

    strong = args.total_batch > 0
    if strong:
        lo, hi = shard_range(args.total_batch, rank, world)
        B, total = hi - lo, args.total_batch
    else:
        B = args.batch or DEFAULT_BATCH[args.config]
        total = B * world
    cfg = workload(args.config, B, seed=rank)
    if cfg['kind'] == 'solve':
        cfg['n_iterations'] = args.iterations
    N = cfg['N']
    ws = make_workspace(cfg, B, local_rank, args)
    watched['ws'] = ws

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ws.sync()

    gathered = {}

    def after_step():
        if dist is None or cfg['kind'] != 'solve':
            return
        ws.sync()
        rec = pack_summary(ws.download(_lib.F_COST), ws.download(_lib.F_ITERS), ws.download(_lib.F_STATUS))
        # rank r's records land at shard_range(total, r, world) (weak scaling: equal shards of B)
        gathered['rec'] = all_gather_summary(rec, total, device=torch.device('cuda', local_rank) if args.backend == 'nccl' else None)

    elapsed, prof, timed_kernel = run_steps(ws, cfg, args.steps, args.warmup, after_step=after_step, barrier=barrier)
    per_rank_ms = None
    if dist is not None:
        # every rank's own wall time of the timed region (all-gather, so a straggler shows in rank 0's line); value uses the MAX
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        ts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 3) for x in ts]
        elapsed = max(float(x.item()) for x in ts)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    kern, info, extra = kernel_table(ws, cfg, prof, args.steps)
    roofline, dominant = roofline_of(kern, timed_kernel)
    W = info['n_alpha_parallel']
    # HBM traffic and VALU-busy of that kernel from the committed rocprofv3 PMC passes (collected separately: counters cannot
    # be read from inside this process), only when they were taken on this very configuration AND on these very kernel sources
    if cfg['kind'] == 'solve' and args.iterations == 10 and not strong:
        add_pmc(roofline, dominant, pmc_for(args.config, cfg, B, info), extra, per_step_work(ws.n, ws.m, *SYSTEM_OPS[cfg['system']]), cfg)
    roofline.update(n_alpha_parallel=W, ls_widths=info['ls_widths'],
                    note='peak = FP64 vector rate (= FP64 MFMA rate on MI355X); MFMA itself is not used, see DESIGN.md')

    value = total * args.steps / elapsed
    unit = {'ltvlqr': 'gain schedules/s', 'mpc': 'solve_ilqr calls/s'}.get(cfg['kind'], 'solves/s')
    out = {
        'metric': {'solve': 'iLQR solves/sec (fixed iters, batched)', 'ltvlqr': 'LTV-LQR gain schedules/sec (batched)',
                   'mpc': 'receding-horizon iLQR solves/sec (warm-started, batched)'}[cfg['kind']],
        'value': round(value, 2), 'unit': unit,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
        'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'baseline_config': args.config, 'workload': '%s, %s%s' % (cfg['label'], ('ONE batch of %d problems split over %d GPU(s)' % (total, world)) if strong
                                             else 'B=%d problems per GPU' % B,
                                             ', %d iterations, reference line search + stop tests active' % cfg['n_iterations'] if cfg['kind'] != 'ltvlqr' else ''),
                   'problems_per_gpu': B, 'problems_total': total, 'n_timesteps': N, 'n_alpha_parallel': W, 'pipeline': info['pipeline'],
                   'sharding': 'by problem index, %d rank(s), all-gather of {cost,iters,status} only (%s)'
                               % (world, 'RCCL' if args.backend == 'nccl' else 'gloo rehearsal')},
        'roofline': roofline,
        'kernels_note': 'HIP events on the workspace stream: k_%s over the timed steps, the other kernels over one untimed step '
                        'before them (timing every pass costs the headline step 1 %%)' % timed_kernel,
        'kernels': {k_: {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in kv.items()} for k_, kv in kern.items()},
    }
    if per_rank_ms is not None:
        out['per_rank_ms_per_step'] = per_rank_ms
    if dist is not None and 'rec' in gathered:
        # a rank whose shard arrived: every problem of its slice carries a backward-pass count >= 1 (a record that never
        # arrived is the all-gather buffer's zeros)
        rec = gathered['rec']
        out['ranks_seen'] = sum(1 for r in range(world) if (lambda lo, hi: hi > lo and bool(np.all(rec[lo:hi, 1] >= 1)))(*shard_range(total, r, world)))
        out['ranks_expected'] = world
        out['gathered_records'] = int(rec.shape[0])
        out['gathered_status_counts'] = {_lib.STATUS_NAMES[s]: int((gathered['rec'][:, 2] == s).sum()) for s in range(4)}
    if cfg['kind'] in ('solve', 'mpc'):
        status, trace = extra['status'], extra['trace']
        out['config']['n_iterations'] = cfg['n_iterations']
        # one backward + line search + accept pass, from the wall clock of the step (the accept step of iteration i and the
        # recursion of iteration i + 1 share launches: the per-kernel times are not additive): (step - initial rollout -
        # initial linearisation) / iterations
        once_ms = sum(kern[k_]['ms_per_pass'] for k_ in ('rollout', 'linearize') if k_ in kern)
        out['ms_per_bwd_fwd_pass'] = round((1e3 * elapsed / args.steps - once_ms) / max(cfg['n_iterations'], 1), 4)
        out['exit_status_counts'] = {_lib.STATUS_NAMES[s]: int((status == s).sum()) for s in range(4)}
        out['mean_ref_trials_per_linesearch'] = round(float(extra['ref_trials'].sum()) / max(int(extra['ran_ls'].sum()), 1), 2)
        out['trials_evaluated_over_needed'] = round(float(extra['eval_trials'].sum()) / max(float(extra['ref_trials'].sum()), 1.0), 3)
        # sanity: problem 0 is the reference's own example; compare its first iterations with the committed golden trace
        if args.config == 'c2' and not strong:
            g = golden('acrobot_solve')
            k_it = min(cfg['n_iterations'], g['trace'].shape[0])
            out['check_problem0_cost_rel_err'] = float(np.max(np.abs(trace[0, :k_it, 0] - g['trace'][:k_it, 0]) / np.abs(g['trace'][:k_it, 0])))
        if not args.no_cpu and world == 1 and cfg['kind'] == 'solve':
            threads = min(16, os.cpu_count() or 1)
            sample = min(args.cpu_sample, B)
            v, secs, r = cpu_baseline(cfg, cfg['n_iterations'], sample, threads)
            cost = ws.download(_lib.F_COST)
            out['cpu_baseline'] = {'value': round(v, 3), 'unit': 'solves/s', 'cores': threads, 'kind': 'port',
                                   'sample': 'first %d of the %d problems, same %d iterations, oracle/ilqr_oracle.c with OpenMP, %.1f s'
                                             % (sample, B, cfg['n_iterations'], secs),
                                   'gpu_vs_cpu_cost_max_rel_diff': float(np.max(np.abs(cost[:sample] - r['cost']) / np.abs(r['cost']))),
                                   'gpu_vs_cpu_cost_frac_within_1e-8': float(np.mean(np.abs(cost[:sample] - r['cost']) <= 1e-8 * np.abs(r['cost']))),
                                   'gpu_vs_cpu_same_exit_status_frac': float(np.mean(status[:sample] == r['status']))}
            one = min(64, B)                                  # the same code on ONE core (SURVEY 8d asks for both figures)
            v1, secs1, _ = cpu_baseline(cfg, cfg['n_iterations'], one, 1)
            out['cpu_baseline']['single_core'] = {'value': round(v1, 3), 'unit': 'solves/s', 'cores': 1,
                                                  'sample': 'first %d problems, %.1f s' % (one, secs1)}
    ws.close()
    if world == 1 and not args.no_extra and args.config == 'c2' and not strong and args.batch in (0, DEFAULT_BATCH['c2']):
        out['extra_configs'] = {}
        for name in ('c2_no_stop', 'c3', 'c4', 'c4_w8', 'c5', 'pendulum', 'acrobot_big', 'mpc', 'floating_base'):
            try:
                out['extra_configs'][name] = extra_config(name, args)
            except Exception as e:                             # an extra block must never cost the headline line
                out['extra_configs'][name] = {'error': repr(e)[:300]}
        for name in ('c1', 'c2_api'):
            try:
                out['extra_configs'][name] = api_config(name)
                if name == 'c2_api':
                    out['extra_configs'][name]['resident_ms_per_step'] = out['ms_per_step']
            except Exception as e:
                out['extra_configs'][name] = {'error': repr(e)[:300]}
    json_out.write(json.dumps(out) + '\n')
    json_out.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
