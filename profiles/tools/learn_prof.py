"""Where learn()'s time goes beyond the three kernels (not collected by pytest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paccmann_generator_amd import ops
from paccmann_generator_amd.reinforce import pad_rows_on_device
cfg = dict(bench.CONFIGS['cfg2']); B, T = cfg['batch'], cfg['length']
dev = torch.device('cuda')
learner = bench.build_learner(cfg, dev)
dec = learner.generator.decoder
torch.manual_seed(1)
latent = torch.randn(1, B, cfg['latent_dim'], device=dev)
rewards = torch.rand(B, device=dev)
w = dec.weights()
tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.8, seed=1, early_exit=False)
padded = pad_rows_on_device(tokens, lens, 3, t_max=T + 2)
idx = torch.arange(B, device=dev)
dec_params = list(dec.parameters())
def tm(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return r, sorted(ts)[len(ts) // 2]
res = {}
hidden, res['l2h+index'] = tm(lambda: dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1)[:, idx, :]))
_, res['weights()'] = tm(lambda: dec.weights())
def fwdloss():
    h = dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1)[:, idx, :])
    return ops.reinforce_loss(dec.weights(), padded, h, rewards, 1.0 / B, trusted_tokens=True)[0]
loss, res['l2h+reinforce_loss op'] = tm(fwdloss)
def bwd():
    l = fwdloss(); torch.cuda.synchronize(); t0 = time.perf_counter(); l.backward(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
res['loss.backward()'] = sorted(bwd() for _ in range(5))[2]
_, res['zero_grad'] = tm(lambda: learner.optimizer.zero_grad())
fwdloss().backward()
_, res['clip_grad_norm_'] = tm(lambda: torch.nn.utils.clip_grad_norm_(dec_params, 0.1))
_, res['optimizer.step'] = tm(lambda: learner.optimizer.step())
_, res['pad_rows_on_device'] = tm(lambda: pad_rows_on_device(tokens, lens, 3, t_max=T + 2))
_, res['decode op'] = tm(lambda: ops.decode(w, latent, 2, 3, T, temperature=0.8, seed=1, early_exit=False))
_, res['whole learn()'] = tm(lambda: learner.learn(latent, padded, idx, rewards))
ph = bench.time_phases(learner, cfg, latent, rewards, B, reps=3)
print({k: round(v, 3) for k, v in res.items()})
print({k: round(v, 3) for k, v in ph.items() if not k.startswith('_')})
