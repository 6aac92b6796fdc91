#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ from the CPU oracle.

PARITY UNPINNED (oracle/__init__.py): the reference holds no fixture for this path and
its decoder (paccmann_chemistry) is not importable here, so these vectors come from the
build's own restatement (oracle/stack_gru.py driven through the loop of
/root/reference/paccmann_generator/reinforce_omics.py:304-335, restated in
oracle/reinforce_loop.py).  They freeze the oracle: a later change of the oracle or of
torch's GRU arithmetic shows up as a diff against these files.

Every case stores inputs (weights by state_dict name, latent, tokens, h0, reward) and
expected outputs computed in float64 (greedy token matrix from the fp32 oracle, which
is what "bit-exact greedy" is judged against).  Run from the repo root:

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import reinforce_loop as RL  # noqa: E402
from oracle.philox import sample_row, uniform  # noqa: E402
from oracle.stack_gru import GreedySearch, OracleGenerator, OracleStackGRUDecoder, build_decoder  # noqa: E402
from tests.golden_util import sample_index, weight_checksum  # noqa: E402
from tests.parity import CFG1, CFG2, CFG4, START, STOP, TINY, synthetic_tokens  # noqa: E402

CASES = {
    # name: (params, B_train, T_train, B_decode, gen_len)
    'tiny_l1': (dict(TINY), 20, 9, 20, 12),
    'tiny_l2': (dict(TINY, n_layers=2), 19, 7, 17, 10),
    'cfg1': (dict(CFG1), 64, 100, 64, 100),
}


# Compact cases (BASELINE.json configs[1] and configs[3] model shapes at a length that crosses every
# stack regime): weights are NOT stored -- they are build_decoder(params, seed=0), pinned by a
# checksum -- and each parameter gradient is stored as <= 4096 evenly spaced samples plus its sum
# and L2 norm, so that a 10-million-parameter case stays a small fixture.
COMPACT = {
    'cfg2_t130': (dict(CFG2), 18, 130, 24, 128),
    'cfg4_t130': (dict(CFG4), 18, 130, 20, 110),
}


def as64(dec):
    d = OracleStackGRUDecoder(dec.params, dtype=torch.float64)
    d.load_state_dict({k: v.double() for k, v in dec.state_dict().items()})
    return d.eval()


def sampled_tokens(dec, latent, gen_len, temperature, seed):
    """Philox inverse-CDF sampling replayed through the fp32 oracle (the CPU twin of the
    kernel's draw, oracle/philox.py)."""
    L = dec.n_layers
    hidden = dec.latent_to_hidden(latent.repeat(L, 1, 1))
    B = hidden.shape[1]
    dec._update_batch_size(B)
    stack = dec.init_stack
    tok = torch.full((1, B), START, dtype=torch.long)
    out = np.zeros((B, gen_len), dtype=np.int64)
    margins = np.zeros((B, gen_len))
    with torch.no_grad():
        for t in range(gen_len):
            o, hidden, stack = dec(tok, hidden, stack)
            logits = dec.output_layer(o).squeeze(0).numpy()
            for b in range(B):
                c, m, _ = sample_row(logits[b], np.float32(temperature), uniform(seed, b, t))
                out[b, t] = c
                margins[b, t] = m
            tok = torch.from_numpy(out[:, t]).view(1, B)
    return out, margins


def make_case(name, params, Bt, Tt, Bd, gen_len, compact=False):
    dec = build_decoder(params, seed=0)
    L, H = params['n_layers'], params['rnn_cell_size']
    out = {}
    if compact:
        out['w_checksum'] = weight_checksum(dec.state_dict())
    else:
        for k, v in dec.state_dict().items():
            out['w.' + k] = v.numpy()
    out['params_keys'] = np.array(sorted(params.keys()))
    out['params_vals'] = np.array([-1 if params[k] is None else params[k] for k in sorted(params.keys())])

    # ---- decode -------------------------------------------------------------------
    torch.manual_seed(1)
    latent = torch.randn(1, Bd, params['latent_dim'])
    gen = OracleGenerator(dec)
    greedy = gen.generate_raw(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len, GreedySearch())
    out['dec.latent'] = latent.numpy()
    out['dec.greedy_tokens'] = greedy.numpy()
    # top-2 logit gap of the fp64 oracle along the greedy path: certifies no near tie
    d64 = as64(dec)
    g64 = OracleGenerator(d64).generate_raw(latent.double(), torch.tensor([START]), torch.tensor([STOP]), gen_len,
                                            GreedySearch())
    out['dec.greedy_tokens_fp64'] = g64.numpy()
    if name != 'cfg1' and not compact:
        toks, margins = sampled_tokens(dec, latent, gen_len, 0.8, 1234)
        out['dec.sampled_tokens'] = toks
        out['dec.sampled_margin'] = margins
        out['dec.temperature'] = np.float32(0.8)
        out['dec.philox_seed'] = np.int64(1234)

    # ---- teacher-forced log-probs + REINFORCE gradients (float64) --------------------
    tokens = synthetic_tokens(Tt, Bt, params['vocab_size'], seed=3)
    torch.manual_seed(1)
    h0 = torch.tanh(torch.randn(L, Bt, H))
    torch.manual_seed(2)
    reward = torch.rand(Bt)
    h64 = h0.double().requires_grad_(True)
    loss = RL.reinforce_loss(d64, tokens, h64, reward.double()[:, None])  # mean over B == scale 1/B
    loss.backward()
    with torch.no_grad():
        logp = RL.teacher_forced_logprobs(d64, tokens, h64.detach())
    out['train.tokens'] = tokens.numpy()
    out['train.h0'] = h0.numpy()
    out['train.reward'] = reward.numpy()
    out['train.scale'] = np.float64(1.0 / Bt)
    out['train.loss'] = np.float64(loss.item())
    out['train.logp'] = logp.numpy()
    out['train.grad_h0'] = h64.grad.numpy()
    for n, p in d64.named_parameters():
        if p.grad is not None:
            out['grad.' + n] = p.grad.numpy()
    full = {k[5:]: v for k, v in out.items() if k.startswith('grad.')}
    # fp32 round-off bound: the same computation in fp32 autograd
    h32 = h0.clone().requires_grad_(True)
    for p in dec.parameters():
        p.grad = None
    RL.reinforce_loss(dec, tokens, h32, reward[:, None]).backward()
    worst = 0.0
    for n, p in dec.named_parameters():
        if p.grad is not None:
            ref = full[n]
            worst = max(worst, float(np.abs(p.grad.numpy() - ref).max() / (np.abs(ref).max() + 1e-30)))
    out['train.fp32_autograd_max_rel_err'] = np.float64(worst)
    # big float64 gradients are stored as float32 for the cfg1 case to keep the file small
    if name == 'cfg1':
        for k in list(out):
            if k.startswith('grad.') or k.startswith('w.'):
                out[k] = out[k].astype(np.float32)
    if compact:
        for n, gfull in full.items():
            flat = out.pop('grad.' + n).reshape(-1)
            out['gsample.' + n] = flat[sample_index(flat.size)]
            out['gstat.' + n] = np.array([flat.sum(), np.sqrt((flat * flat).sum()), np.abs(flat).max()])
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1e6:.2f} MB, greedy fp32==fp64: '
          f'{bool((greedy.numpy() == g64.numpy()).all())}, fp32 autograd rel err {worst:.2e}')


if __name__ == '__main__':
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (params, Bt, Tt, Bd, gl) in CASES.items():
        if not only or name in only:
            make_case(name, params, Bt, Tt, Bd, gl)
    for name, (params, Bt, Tt, Bd, gl) in COMPACT.items():
        if not only or name in only:
            make_case(name, params, Bt, Tt, Bd, gl, compact=True)
