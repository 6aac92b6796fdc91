"""scratch: parity of the clustered kernels with several row groups per cluster + phase times vs batch"""
import json, subprocess, sys
from tests import parity as P
from paccmann_generator_amd import ops

MID = dict(embedding_size=32, stack_width=20, stack_depth=12, n_layers=1, latent_dim=64, pad_index=0, vocab_size=40)
ops.USE_CLUSTER = True
for H, B in [(256, 300), (512, 200), (128, 600), (384, 250), (256, 1000), (512, 700)]:
    cfg = dict(MID, rnn_cell_size=H)
    P.check_train(cfg, B=B, T=14, tol=1e-4)
    r = P.check_greedy(cfg, B, 16)
    s = P.check_sampling(cfg, B, 12, temperature=1.1)
    print('ok', H, B, r['mismatched_rows'], r.get('near_ties'), s['exact'], s['boundary'], flush=True)
cfg = dict(P.CFG1, rnn_cell_size=512)
P.check_train(cfg, B=300, T=70, tol=1e-4)
print('ok cfg H512 B300 T70', flush=True)
