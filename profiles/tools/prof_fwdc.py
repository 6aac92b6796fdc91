"""Phase counters of the clustered forward (uncollected).  Build: g.build(extra_flags=['-DPMG_PROFILE'], libname='libpmg_prof.so');
run: PMG_LIB=.../libpmg_prof.so python profiles/tools/prof_fwdc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paccmann_generator_amd import ops
from paccmann_generator_amd.reinforce import pad_rows_on_device
cfg = dict(bench.CONFIGS[os.environ.get('CFG', 'cfg1')])
B, T = int(os.environ.get('B', cfg['batch'])), cfg['length']
dev = torch.device('cuda')
learner = bench.build_learner(cfg, dev)
dec = learner.generator.decoder
w = dec.weights()
torch.manual_seed(1)
latent = torch.randn(1, B, cfg['latent_dim'], device=dev)
tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.8, seed=1, early_exit=False)
torch.cuda.synchronize()
os.environ['PMG_DEBUG'] = '0x8000'
print('decode:', flush=True)
ops.decode(w, latent, 2, 3, T, temperature=0.8, seed=1, early_exit=False)
torch.cuda.synchronize()
padded = pad_rows_on_device(tokens, lens, 3, t_max=T + 2)
with torch.no_grad():
    h0 = dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1)).contiguous()
print('teacher:', flush=True)
ops.teacher_forced_logprobs(w, padded, h0)
torch.cuda.synchronize()
print('bptt:', flush=True)
rewards = torch.rand(B, device=dev)
from paccmann_generator_amd import _ffi
loss, logp = ops.reinforce_loss(w, padded, h0.requires_grad_(True), rewards, 1.0 / B)
torch.cuda.synchronize()
