"""A/B of the two forward decompositions against the fp64 oracle: worst element-wise error per tensor over
several seeds, at the shape where one element of the clustered run crossed 1e-4 (scratch, uncollected)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import parity as P
from paccmann_generator_amd import ops
cfg = dict(P.TINY, rnn_cell_size=512, stack_width=3, stack_depth=2, vocab_size=97)
for use in (True, False):
    ops.USE_CLUSTER = use
    worst = []
    for seed in range(12):
        try:
            r = P.check_train(cfg, B=5, T=6, seed=seed, tol=1e-3, elem_tol=1e-2)
            k, v = max(r['elem'].items(), key=lambda kv: kv[1])
            worst.append((v, k))
        except AssertionError as e:
            worst.append((9.9, str(e)[:80]))
    print('cluster' if use else 'batch  ', ' '.join(f'{v:.1e}' for v, _ in worst), '| max', max(worst)[0], max(worst)[1], flush=True)
