"""numpy model of the *algorithm the HIP kernels implement* (hand-derived BPTT).

TEST INFRASTRUCTURE (see oracle/__init__.py).  It restates, in plain numpy, the same
arithmetic as oracle/stack_gru.py + oracle/reinforce_loop.py (reference loop:
/root/reference/paccmann_generator/reinforce_omics.py:304-335) but organised the way
the kernels organise it, so that each kernel stage has a line-by-line CPU twin:

* the embedding half of layer 0's input GEMM is a per-token table
  ``Gtab[v] = W_ih0[:, :E] @ emb[v] + b_ih0``;
* the backward is an explicit reverse-time loop (no autograd) producing every
  parameter gradient, ``grad_h0`` and the loss;
* the embedding / W_ih0[:, :E] / b_ih0 gradients are recovered from ``dGtab``.

tests/test_oracle.py checks it against torch autograd in float64.
"""
from __future__ import annotations

import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def params_from_decoder(decoder, dtype=np.float64):
    """Pull torch-layout weights out of an oracle (or product) decoder module."""
    g = lambda t: t.detach().cpu().numpy().astype(dtype)
    L = decoder.n_layers
    P = dict(
        emb=g(decoder.embedding.weight),
        w_ih=[g(getattr(decoder.gru, f'weight_ih_l{l}')) for l in range(L)],
        w_hh=[g(getattr(decoder.gru, f'weight_hh_l{l}')) for l in range(L)],
        b_ih=[g(getattr(decoder.gru, f'bias_ih_l{l}')) for l in range(L)],
        b_hh=[g(getattr(decoder.gru, f'bias_hh_l{l}')) for l in range(L)],
        wc=g(decoder.stack_controls_layer.weight), bc=g(decoder.stack_controls_layer.bias),
        wp=g(decoder.stack_input_layer.weight), bp=g(decoder.stack_input_layer.bias),
        wo=g(decoder.output_layer.weight), bo=g(decoder.output_layer.bias),
        pad_idx=decoder.embedding.padding_idx if decoder.embedding.padding_idx is not None else -1,
        D=decoder.stack_depth,
    )
    return P


def gate_table(P):
    E = P['emb'].shape[1]
    return P['emb'] @ P['w_ih'][0][:, :E].T + P['b_ih'][0]


def forward_step(P, gtab, tok, h, S):
    """One decoder step.  tok [B] int; h list of L [B,H]; S [B,D,W].  Returns new (h, S) and a cache."""
    L = len(h)
    H = h[0].shape[1]
    E = P['emb'].shape[1]
    hl = h[L - 1]
    c = hl @ P['wc'].T + P['bc']
    c = c - c.max(axis=1, keepdims=True)
    a = np.exp(c)
    a = a / a.sum(axis=1, keepdims=True)          # (push, pop, no-op)
    v = np.tanh(hl @ P['wp'].T + P['bp'])         # [B,W]
    up = np.concatenate([v[:, None, :], S[:, :-1]], axis=1)
    down = np.concatenate([S[:, 1:], np.zeros_like(S[:, :1])], axis=1)
    Sn = a[:, 2, None, None] * S + a[:, 0, None, None] * up + a[:, 1, None, None] * down
    top = Sn[:, 0, :]
    hn, cache_l = [], []
    x = None
    for l in range(L):
        if l == 0:
            gi = gtab[tok] + top @ P['w_ih'][0][:, E:].T
        else:
            gi = x @ P['w_ih'][l].T + P['b_ih'][l]
        gh = h[l] @ P['w_hh'][l].T + P['b_hh'][l]
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        ghn = gh[:, 2 * H:]
        n = np.tanh(gi[:, 2 * H:] + r * ghn)
        hnew = (1.0 - z) * n + z * h[l]
        cache_l.append((r, z, n, ghn, h[l], x))
        hn.append(hnew)
        x = hnew
    cache = dict(a=a, v=v, S_prev=S, up=up, down=down, top=top, layers=cache_l, hl=hl, tok=tok)
    return hn, Sn, cache


def logits_of(P, h_top):
    return h_top @ P['wo'].T + P['bo']


def log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    s = e.sum(axis=1, keepdims=True)
    return x - m - np.log(s), e / s


def forward_backward(P, tokens, h0, reward, scale):
    """Teacher-forced forward + reward-weighted BPTT.

    tokens [T,B] int; h0 [L,B,H]; reward [B]; scale = 1/B_valid.
    Returns dict(logp [T-1,B], loss, grads{...}, grad_h0 [L,B,H])."""
    dt = h0.dtype
    T, B = tokens.shape
    L, _, H = h0.shape
    E = P['emb'].shape[1]
    W = P['wp'].shape[0]
    D = P['D']
    V = P['wo'].shape[0]
    gtab = gate_table(P)
    h = [h0[l].copy() for l in range(L)]
    S = np.zeros((B, D, W), dtype=dt)
    caches, probs, logp = [], [], []
    for p in range(T - 1):
        h, S, cache = forward_step(P, gtab, tokens[p], h, S)
        cache['h_out'] = [x.copy() for x in h]
        lsm, pr = log_softmax(logits_of(P, h[L - 1]))
        logp.append(lsm[np.arange(B), tokens[p + 1]])
        probs.append(pr)
        caches.append(cache)
    logp = np.stack(logp) if logp else np.zeros((0, B), dtype=dt)
    loss = -(logp * reward[None, :]).sum() * scale

    G = dict(
        emb=np.zeros_like(P['emb']),
        w_ih=[np.zeros_like(w) for w in P['w_ih']], w_hh=[np.zeros_like(w) for w in P['w_hh']],
        b_ih=[np.zeros_like(b) for b in P['b_ih']], b_hh=[np.zeros_like(b) for b in P['b_hh']],
        wc=np.zeros_like(P['wc']), bc=np.zeros_like(P['bc']),
        wp=np.zeros_like(P['wp']), bp=np.zeros_like(P['bp']),
        wo=np.zeros_like(P['wo']), bo=np.zeros_like(P['bo']),
    )
    dgtab = np.zeros_like(gtab)
    dh = [np.zeros((B, H), dtype=dt) for _ in range(L)]
    dS = np.zeros((B, D, W), dtype=dt)
    for p in range(T - 2, -1, -1):
        c = caches[p]
        onehot = np.zeros((B, V), dtype=dt)
        onehot[np.arange(B), tokens[p + 1]] = 1.0
        dlog = (probs[p] - onehot) * (reward * scale)[:, None]
        dh[L - 1] = dh[L - 1] + dlog @ P['wo']
        G['wo'] += dlog.T @ c['h_out'][L - 1]
        G['bo'] += dlog.sum(0)
        dh_prev = [None] * L
        dtop = None
        for l in range(L - 1, -1, -1):
            r, z, n, ghn, hp, x = c['layers'][l]
            d = dh[l]
            dn = d * (1.0 - z)
            dz = d * (hp - n)
            dn_pre = dn * (1.0 - n * n)
            dr = dn_pre * ghn
            dghn = dn_pre * r
            dz_pre = dz * z * (1.0 - z)
            dr_pre = dr * r * (1.0 - r)
            dgi = np.concatenate([dr_pre, dz_pre, dn_pre], axis=1)
            dgh = np.concatenate([dr_pre, dz_pre, dghn], axis=1)
            dh_prev[l] = d * z + dgh @ P['w_hh'][l]
            G['w_hh'][l] += dgh.T @ hp
            G['b_hh'][l] += dgh.sum(0)
            if l > 0:
                dh[l - 1] = dh[l - 1] + dgi @ P['w_ih'][l]
                G['w_ih'][l] += dgi.T @ x
                G['b_ih'][l] += dgi.sum(0)
            else:
                dtop = dgi @ P['w_ih'][0][:, E:]
                G['w_ih'][0][:, E:] += dgi.T @ c['top']
                np.add.at(dgtab, c['tok'], dgi)
        dSn = dS.copy()
        dSn[:, 0, :] += dtop
        a = c['a']
        da_noop = (dSn * c['S_prev']).sum(axis=(1, 2))
        da_push = (dSn * c['up']).sum(axis=(1, 2))
        da_pop = (dSn * c['down']).sum(axis=(1, 2))
        dv = a[:, 0, None] * dSn[:, 0, :]
        dS_prev = a[:, 2, None, None] * dSn
        dS_prev[:, :-1] += a[:, 0, None, None] * dSn[:, 1:]
        dS_prev[:, 1:] += a[:, 1, None, None] * dSn[:, :-1]
        da = np.stack([da_push, da_pop, da_noop], axis=1)
        dc = a * (da - (a * da).sum(axis=1, keepdims=True))
        dvp = dv * (1.0 - c['v'] * c['v'])
        dh_prev[L - 1] = dh_prev[L - 1] + dc @ P['wc'] + dvp @ P['wp']
        G['wc'] += dc.T @ c['hl']
        G['bc'] += dc.sum(0)
        G['wp'] += dvp.T @ c['hl']
        G['bp'] += dvp.sum(0)
        dh = dh_prev
        dS = dS_prev
    # table -> embedding / W_ih0[:, :E] / b_ih0
    G['w_ih'][0][:, :E] += dgtab.T @ P['emb']
    G['b_ih'][0] += dgtab.sum(0)
    G['emb'] += dgtab @ P['w_ih'][0][:, :E]
    if P['pad_idx'] is not None and P['pad_idx'] >= 0:
        G['emb'][P['pad_idx']] = 0.0
    return dict(logp=logp, loss=loss, grads=G, grad_h0=np.stack(dh))


def greedy_decode(P, h0, start_tok, gen_len):
    """Greedy token matrix [B, gen_len] (argmax, first index on ties)."""
    L, B, H = h0.shape
    W = P['wp'].shape[0]
    gtab = gate_table(P)
    h = [h0[l].copy() for l in range(L)]
    S = np.zeros((B, P['D'], W), dtype=h0.dtype)
    tok = np.full((B,), start_tok, dtype=np.int64)
    out = np.zeros((B, gen_len), dtype=np.int64)
    for t in range(gen_len):
        h, S, _ = forward_step(P, gtab, tok, h, S)
        tok = np.argmax(logits_of(P, h[L - 1]), axis=1)
        out[:, t] = tok
    return out
