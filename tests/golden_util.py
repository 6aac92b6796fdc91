"""Loading of the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ('tiny_l1', 'tiny_l2', 'cfg1')
COMPACT = ('cfg2_t130', 'cfg4_t130')     # weights from the seed (checksummed), gradients as samples + statistics


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    keys = [str(k) for k in g.pop('params_keys')]
    vals = [int(v) for v in g.pop('params_vals')]
    params = {k: (None if (k == 'pad_index' and v < 0) else v) for k, v in zip(keys, vals)}
    return params, g


def state_dict(g, dtype=torch.float32):
    return {k[2:]: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in g.items() if k.startswith('w.')}


def grads(g):
    return {k[5:]: np.asarray(v, dtype=np.float64) for k, v in g.items() if k.startswith('grad.')}


def sample_index(n: int, k: int = 4096):
    """Evenly spaced flat indices (all of them when n <= k), first and last element included."""
    return np.arange(n) if n <= k else np.linspace(0, n - 1, k).astype(np.int64)


def weight_checksum(state_dict):
    """float64 [sum, sum of squares, count] over every tensor of a state_dict, in key order."""
    s = q = 0.0
    n = 0
    for k in sorted(state_dict):
        v = state_dict[k].detach().double()
        s += float(v.sum())
        q += float((v * v).sum())
        n += v.numel()
    return np.array([s, q, float(n)])


def compact_decoder(params, g):
    """The oracle decoder a compact case was made from: build_decoder(params, seed=0), refused when
    this torch does not reproduce the weights the fixture was generated with."""
    from oracle.stack_gru import build_decoder
    dec = build_decoder(params, seed=0)
    got, ref = weight_checksum(dec.state_dict()), g['w_checksum']
    if not np.allclose(got, ref, rtol=1e-12, atol=0):
        raise RuntimeError('seeded weights differ from the ones this golden case was generated with')
    return dec


def compact_grads(g):
    """{name: (samples, [sum, l2, max|.|])} of a compact case."""
    return {k[8:]: (np.asarray(v, dtype=np.float64), np.asarray(g['gstat.' + k[8:]], dtype=np.float64))
            for k, v in g.items() if k.startswith('gsample.')}


def check_compact_grad(name, got, samples, stat, tol):
    """got: full gradient tensor (numpy).  Samples within tol of the tensor's largest entry, L2 norm
    and sum within tol (sum: relative to the L2 norm times sqrt(n), its natural scale)."""
    flat = np.asarray(got, dtype=np.float64).reshape(-1)
    s = flat[sample_index(flat.size)]
    assert np.abs(s - samples).max() <= tol * stat[2], (name, 'samples', float(np.abs(s - samples).max() / stat[2]))
    l2 = np.sqrt((flat * flat).sum())
    assert abs(l2 - stat[1]) <= tol * stat[1], (name, 'l2', l2, stat[1])
    assert abs(flat.sum() - stat[0]) <= tol * stat[1] * np.sqrt(flat.size), (name, 'sum')
