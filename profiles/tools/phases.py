"""Phase timings of one library variant (not collected by pytest).  PMG_LIB selects the .so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
cfg = dict(bench.CONFIGS[os.environ.get('CFG', 'cfg2')])
B = int(os.environ.get('B', cfg['batch']))
dev = torch.device('cuda')
learner = bench.build_learner(cfg, dev)
torch.manual_seed(1)
latent = torch.randn(1, B, cfg['latent_dim'], device=dev)
rewards = torch.rand(B, device=dev)
ph = bench.time_phases(learner, cfg, latent, rewards, B, reps=int(os.environ.get('REPS', '5')))
ph.pop('_steps'); ph.pop('_batch', None)
print(os.path.basename(os.environ.get('PMG_LIB', 'libpmg.so')), ' '.join(f'{k}={v:.3f}' for k, v in ph.items()),
      f'sum={sum(ph.values()):.3f}', flush=True)
