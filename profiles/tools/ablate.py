"""Timing ablations (not collected by pytest): PMG_DEBUG bits skip phases of the kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paccmann_generator_amd import _ffi, ops
from paccmann_generator_amd.reinforce import pad_rows_on_device

cfg = dict(bench.CONFIGS[os.environ.get('CFG', 'cfg2')])
B, T = int(os.environ.get('B', cfg['batch'])), cfg['length']
dev = torch.device('cuda')
learner = bench.build_learner(cfg, dev)
dec = learner.generator.decoder
w = dec.weights()
torch.manual_seed(1)
latent = torch.randn(1, B, cfg['latent_dim'], device=dev)
rewards = torch.rand(B, device=dev)
tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.8, seed=1, early_exit=False)
padded = pad_rows_on_device(tokens, lens, 3, t_max=T + 2)
Tt = padded.shape[0]
with torch.no_grad():
    h0 = dec.latent_to_hidden(latent.repeat(dec.n_layers, 1, 1)).contiguous()
shape = w.shape(B, Tt)
ws = torch.empty(_ffi.workspace_bytes(shape, _ffi.PMG_MODE_TRAIN), dtype=torch.uint8, device=dev)
wstruct, keep = w.struct(False)
logp = torch.empty(Tt - 1, B, device=dev)
grads = [torch.empty_like(p) for p in w.flat()]
gstruct, keep2 = _ffi.pack_weight_struct(_ffi.PmgGrads, ops.DecoderWeights.unflatten(grads, dec.n_layers), dec.n_layers)
gh0 = torch.empty_like(h0)
loss = torch.empty(1, device=dev)
st = _ffi.current_stream()

def fwd():
    _ffi.check(_ffi.lib.pmg_logprob_fwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), _ffi._ptr(h0), _ffi._ptr(logp), None, 0, _ffi._ptr(ws), st), 'fwd')
def bptt():
    _ffi.check(_ffi.lib.pmg_reinforce_bptt(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), _ffi._ptr(rewards), 1.0 / B, 0, _ffi._ptr(gh0), _ffi._ptr(ws), st), 'bptt')
def wgrad():
    _ffi.check(_ffi.lib.pmg_reinforce_wgrad(C.byref(shape), C.byref(wstruct), _ffi._ptr(padded), 0, C.byref(gstruct), _ffi._ptr(loss), _ffi._ptr(ws), st), 'wgrad')
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

names = {0: 'full', 0x100: '-stack', 0x200: '-extras', 0x400: '-gemm_hh', 0x800: '-gemm_ih/top', 0x1000: '-history stores',
         0x2000: '-softmax', 0x1F00 | 0x2000: 'only sync skeleton', 0x100 | 0x1000 | 0x2000: '-stack-hist-softmax',
         0x400 | 0x800 | 0x200: 'no gemm at all'}
for bits, nm in names.items():
    os.environ['PMG_DEBUG'] = hex(bits)
    print(f'fwd  {nm:28s} {timed(fwd):8.3f} ms', flush=True)
os.environ['PMG_DEBUG'] = '0'
fwd()
for bits, nm in names.items():
    if bits & 0x2000 and bits != (0x1F00 | 0x2000) and bits != (0x100 | 0x1000 | 0x2000): continue
    os.environ['PMG_DEBUG'] = hex(bits)
    print(f'bptt {nm:28s} {timed(bptt):8.3f} ms', flush=True)
os.environ['PMG_DEBUG'] = '0'
print(f'wgrad {timed(wgrad):8.3f} ms')
