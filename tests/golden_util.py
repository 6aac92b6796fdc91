"""Loading of the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ('tiny_l1', 'tiny_l2', 'cfg1')


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
