"""Sustained run of the clustered kernels: N decode + training steps per shape, every step's decode lengths and loss
checked on the host (a cluster barrier that gave up would poison them: lengths -1, loss NaN).  Not collected by pytest.
python profiles/tools/soak.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import parity as P
from paccmann_generator_amd import ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = 100
for H, B in [(256, 64), (256, 1024), (384, 896), (512, 128), (512, 768), (128, 512)]:
    cfg = dict(P.CFG1, rnn_cell_size=H)
    _, pdec = P.make_pair(cfg, seed=0)
    w = pdec.weights()
    torch.manual_seed(0)
    latent = torch.randn(1, B, cfg['latent_dim'], device='cuda')
    toks = P.synthetic_tokens(T + 2, B, cfg['vocab_size'], seed=1).cuda()
    h0 = torch.tanh(torch.randn(1, B, H, device='cuda'))
    rew = torch.rand(B, device='cuda')
    bad = 0
    first_loss = None
    t0 = time.perf_counter()
    for i in range(steps):
        tok, lens = ops.decode(w, latent, 2, 3, T, temperature=0.9, seed=3, offset=i * T, early_exit=False)
        h = h0.clone().requires_grad_(True)
        loss, _ = ops.reinforce_loss(w, toks, h, rew, 1.0 / B, trusted_tokens=True)
        loss.backward()
        lv = float(loss)
        if first_loss is None:
            first_loss = lv
        if int(lens.min()) < 0 or lv != lv or lv != first_loss or not bool(torch.isfinite(h.grad).all()):
            bad += 1
    dt = time.perf_counter() - t0
    print(f'H={H} B={B}: {steps} steps, {1e3 * dt / steps:.2f} ms per step incl. host checks, bad steps {bad}, loss {first_loss:.6f} '
          f'(bit-identical every step: {bad == 0})', flush=True)
