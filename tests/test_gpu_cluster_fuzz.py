"""Random shapes through the clustered kernels against the oracle: hidden size, stack width and depth, vocabulary,
embedding and latent sizes, batch sizes around the cluster boundaries (one sequence, a ragged last row group, one
more row group than resident clusters, several row groups per cluster) and sequence lengths down to two steps.
Same bars as the parity modules.  `python -m tests.test_gpu_cluster_fuzz SEED N` runs a longer sweep by hand
(120 cases over four seeds passed when this was written)."""
import ctypes as C
import random
import sys

import pytest

from tests import parity as P

pytestmark = pytest.mark.gpu


def random_case(rng):
    H = rng.choice([128, 256, 384, 512])
    W = rng.choice([1, 2, 7, 16, 20, 33, 50, 64])
    D = rng.choice([2, 3, 5, 12, 30, 50])
    V = min(rng.choice([5, 17, 40, 64, 65, 97, 150]), 256 - 3 - W)
    E = rng.choice([8, 32, 128])
    Z = rng.choice([32, 64, 256])
    cfg = dict(embedding_size=E, rnn_cell_size=H, vocab_size=V, stack_width=W, stack_depth=D, n_layers=1,
               latent_dim=Z, pad_index=0)
    per = 16 * 8 * (32 // (H // 16))               # sequences that one row group per cluster serves
    B = rng.choice([1, 3, 15, 16, 17, 31, per - 1, per, per + 1, 2 * per - 5, 3 * per])
    T = rng.choice([2, 3, 9, 20, 45])
    return cfg, B, T


def run_case(cfg, B, T, seed):
    P.check_train(cfg, B=B, T=max(T, 3), tol=1e-4, seed=seed)
    g = P.check_greedy(cfg, B, T, seed=seed)
    assert g['mismatched_rows'] == g.get('near_ties', 0), g
    r = P.check_sampling(cfg, B, T, temperature=1.0, seed=seed)
    assert r['exact'] + r['boundary'] == B * T and r['boundary'] <= max(2, B * T // 2000), r


def sweep(seed, n):
    from paccmann_generator_amd import _ffi, ops
    old, ops.USE_CLUSTER = ops.USE_CLUSTER, True
    rng = random.Random(seed)
    failures = []
    try:
        for case in range(n):
            cfg, B, T = random_case(rng)
            s = _ffi.make_shape(B, max(T, 2), cfg['rnn_cell_size'], cfg['embedding_size'], cfg['stack_width'],
                                cfg['stack_depth'], cfg['vocab_size'], 1, cfg['latent_dim'], 0)
            tag = (f"case {case}: {cfg} B={B} T={T} clustered decode="
                   f"{_ffi.lib.pmg_uses_cluster(C.byref(s), _ffi.PMG_MODE_DECODE, 0)} train="
                   f"{_ffi.lib.pmg_uses_cluster(C.byref(s), _ffi.PMG_MODE_TRAIN, 0)}")
            try:
                run_case(cfg, B, T, case)
                print('ok  ', tag, flush=True)
            except Exception as e:           # noqa: BLE001 -- report every failing shape, then fail
                failures.append(f'{tag}: {e!r}'[:400])
                print('FAIL', failures[-1], flush=True)
    finally:
        ops.USE_CLUSTER = old
    return failures


@pytest.mark.timeout(300)
def test_random_shapes_on_the_clustered_kernels(gpu):
    failures = sweep(seed=0, n=30)
    assert not failures, '\n'.join(failures)


if __name__ == '__main__':
    bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
    print('failures:', len(bad))
    sys.exit(1 if bad else 0)
