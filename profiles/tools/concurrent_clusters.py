"""Two streams launching clustered kernels at the same time (not collected by pytest): do they starve each other?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import parity as P
from paccmann_generator_amd import ops
cfg = dict(P.CFG1, rnn_cell_size=512)
_, pdec = P.make_pair(cfg, seed=0)
w = pdec.weights()
B, T = int(os.environ.get('B', 768)), 40
lat = [torch.randn(1, B, cfg['latent_dim'], device='cuda') for _ in range(2)]
ref = [ops.decode(w, l, 2, 3, T, temperature=0.9, seed=5 + i, early_exit=False) for i, l in enumerate(lat)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
out = [None, None]
def work(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(int(os.environ.get('REPS', 10))):
            out[i] = ops.decode(w, lat[i], 2, 3, T, temperature=0.9, seed=5 + i, early_exit=False)
    streams[i].synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for i in range(2):
    same = bool(torch.equal(out[i][0], ref[i][0])) and bool(torch.equal(out[i][1], ref[i][1]))
    print(f'stream {i}: identical to the serial run {same}, poisoned rows {int((out[i][1] < 0).sum())}')
print(f'{dt * 1e3:.1f} ms for 2 x {os.environ.get("REPS", 10)} launches')
