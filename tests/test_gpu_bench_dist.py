"""`bench.py --gpus N` end to end as the driver launches it, so that a plumbing failure shows up here
and not on the driver's 8-GPU node.  One GPU per test box, hence two legs:

* N = 2 over gloo (PMG_DIST_BACKEND=gloo; both ranks share the GPU, so only the plumbing is tested:
  self-spawn of the ranks, rendezvous on 127.0.0.1, sharded Philox rows, the flat gradient all-reduce,
  max-over-ranks timing, replica identity, exactly one JSON line from rank 0);
* N = 1 under torch.distributed.run over RCCL ("nccl") with PMG_DIST_SINGLE=1 -- the only RCCL leg a
  one-GPU box allows: communicator bound to the device, the same collectives with world size 1.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--config', 'cfg1', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-phases']


def _one_json_line(out):
    lines = [ln for ln in out.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_two_ranks_gloo(gpu):
    env = dict(os.environ, PMG_DIST_BACKEND='gloo', PMG_HANG_DUMP='100')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + ARGS, env=env,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _one_json_line(r.stdout)
    assert j['n_gpus'] == 2 and j['dist_backend'] == 'gloo' and j['replicas_identical'] is True
    assert j['config']['global_batch'] == 2 * j['config']['batch_per_gpu'] == 128
    assert j['config']['parallelism'] == 'dp2' and j['scaling'] == 'weak'
    assert j['value'] > 0 and abs(j['value'] - 2 * 64 * 100 / (j['ms_per_step'] * 1e-3)) < 1e-6 * j['value']


def test_bench_one_rank_rccl(gpu):
    env = dict(os.environ, PMG_DIST_SINGLE='1', PMG_HANG_DUMP='100')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr',
           '127.0.0.1', '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + ARGS
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _one_json_line(r.stdout)
    assert j['n_gpus'] == 1 and j['dist_backend'] == 'nccl' and j['replicas_identical'] is True
