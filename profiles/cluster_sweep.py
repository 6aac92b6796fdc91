"""Clustered vs batch-partitioned kernels over hidden size and batch: sampling decode and the training pair
(teacher-forced forward + BPTT + weight gradients) at stack 50x50, vocabulary 60, T=100.
Run on the GPU box: python -m profiles.cluster_sweep > profiles/r03_cluster_groups.txt"""
import os
sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, sys_path)
import torch
from tests import parity as P
from paccmann_generator_amd import ops

def timed(f, n=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

T = 100
for H, Bs in [(128, [256, 512]), (256, [64, 256, 512, 768, 1024]), (384, [128, 256, 512, 896]), (512, [128, 256, 512, 768])]:
    cfg = dict(P.CFG1, rnn_cell_size=H)
    _, pdec = P.make_pair(cfg, seed=0)
    w = pdec.weights()
    for B in Bs:
        torch.manual_seed(0)
        latent = torch.randn(1, B, cfg['latent_dim'], device='cuda')
        toks = P.synthetic_tokens(T + 2, B, cfg['vocab_size'], seed=1).cuda()
        h0 = torch.tanh(torch.randn(1, B, H, device='cuda'))
        rew = torch.rand(B, device='cuda')
        out = []
        for uc in (True, False):
            ops.USE_CLUSTER = uc
            d = timed(lambda: ops.decode(w, latent, 2, 3, T, temperature=0.9, seed=3, early_exit=False))
            def pair():
                h = h0.clone().requires_grad_(True)
                loss, _ = ops.reinforce_loss(w, toks, h, rew, 1.0 / B, trusted_tokens=True)
            p = timed(pair)
            out.append((d, p))
        print(f'H={H} B={B}: decode {out[0][0]:.2f} vs {out[1][0]:.2f} ms   train pair {out[0][1]:.2f} vs {out[1][1]:.2f} ms  (clustered vs batch-partitioned)', flush=True)
