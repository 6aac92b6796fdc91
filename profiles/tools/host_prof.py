"""Host-side cost of one bench.py step at a small batch (cProfile over N steps; not collected by pytest).
python profiles/tools/host_prof.py [config] [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paccmann_generator_amd.reinforce import pad_rows_on_device
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = dict(bench.CONFIGS[name]); B, T = cfg['batch'], cfg['length']
dev = torch.device('cuda')
learner = bench.build_learner(cfg, dev)
torch.manual_seed(1)
latent = torch.randn(1, B, cfg['latent_dim'], device=dev)
rewards = torch.rand(B, device=dev)
idx = torch.arange(B, device=dev)
STOP = learner.generator.smiles_language.stop_index
def step():
    tokens, lens = learner.sample_tokens(latent, seq_index_base=0, early_exit=False)
    padded = pad_rows_on_device(tokens, lens, STOP, t_max=T + 2)
    learner.learn(latent, padded, idx, rewards, lrps=1, trusted_tokens=True)
for _ in range(10):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize(); print('ms per step, unprofiled:', 1e3 * (time.perf_counter() - t0) / steps)
pr = cProfile.Profile(); pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)
