"""The process-per-GPU path of bench.py on real kernels: N ranks, each its own process with its own workspace, the summary
records all-gathered.  A one-GPU box can only rehearse it with the ranks SHARING the card (--device-wrap; RCCL refuses two
ranks on one device, hence gloo for the gather) -- the shard / solve / gather / report code is the same the 8-GPU run takes.
Since round 3 a solve uses ONE stream and no events, so two processes on one card is an ordinary
workload; the run is still bounded by bench.py's own watchdog and by a timeout here."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(extra, timeout=240):
    env = dict(os.environ, ILQR_BENCH_WATCHDOG='150', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-extra', '--no-cpu'] + extra
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)                                  # exactly the process group started above
        out, err = p.communicate()
        pytest.fail('bench.py %s did not finish in %d s\n%s' % (' '.join(extra), timeout, err[-2000:]))
    assert p.returncode == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out                              # ONE JSON line on rank 0's stdout
    return json.loads(lines[0])


def test_two_ranks_sharing_the_gpu_weak_and_strong():
    one = run_bench(['--gpus', '1', '--batch', '512'])
    two = run_bench(['--gpus', '2', '--batch', '512', '--device-wrap', '--backend', 'gloo'])
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak'
    assert two['ranks_seen'] == 2 and two['ranks_expected'] == 2 and two['gathered_records'] == 1024
    assert len(two['per_rank_ms_per_step']) == 2 and all(t > 0 for t in two['per_rank_ms_per_step'])
    assert two['config']['problems_total'] == 1024 and two['value'] > 0
    # rank 0 of the two solves the same 512 problems as the single process (seed = rank): problem 0 is the reference's example
    assert two['check_problem0_cost_rel_err'] < 1e-8 and one['check_problem0_cost_rel_err'] < 1e-8
    assert two['exit_status_counts'] == one['exit_status_counts']
    assert sum(two['gathered_status_counts'].values()) == 1024
    # strong scaling: ONE batch of 1000 problems cut by problem index into 500 + 500
    st = run_bench(['--gpus', '2', '--total-batch', '1000', '--device-wrap', '--backend', 'gloo'])
    assert st['scaling'] == 'strong' and st['ranks_seen'] == 2 and st['gathered_records'] == 1000
    assert st['config']['problems_per_gpu'] == 500 and st['config']['problems_total'] == 1000
