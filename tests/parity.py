"""Shared parity helpers: HIP path (through the C ABI) vs the CPU oracle.

Imported by tests/test_gpu_*.py, by __graft_entry__.smoke() and by nothing in the
product package.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import fused_model as FM
from oracle import reinforce_loop as RL
from oracle.philox import sample_row, uniform
from oracle.stack_gru import GreedySearch, OracleGenerator, build_decoder

TINY = dict(embedding_size=16, rnn_cell_size=128, vocab_size=23, stack_width=10, stack_depth=7,
            n_layers=1, latent_dim=32, pad_index=0)
CFG1 = dict(embedding_size=128, rnn_cell_size=256, vocab_size=60, stack_width=50, stack_depth=50,
            n_layers=1, latent_dim=256, pad_index=0)
CFG2 = dict(embedding_size=128, rnn_cell_size=512, vocab_size=60, stack_width=50, stack_depth=50,
            n_layers=1, latent_dim=256, pad_index=0)
CFG4 = dict(embedding_size=128, rnn_cell_size=1024, vocab_size=60, stack_width=50, stack_depth=50,
            n_layers=2, latent_dim=256, pad_index=0)
START, STOP, PAD = 2, 3, 0


def make_pair(params, seed=0, device='cuda'):
    """(oracle decoder on CPU fp32, product decoder on `device`) with identical weights."""
    from paccmann_generator_amd import StackGRUDecoder
    odec = build_decoder(params, seed=seed)
    pdec = StackGRUDecoder(params)
    pdec.load_state_dict(odec.state_dict())
    pdec = pdec.to(device).eval()
    return odec, pdec


def synthetic_tokens(T, B, V, seed=3):
    """Rows shaped like the harness builds them: random body, then stop, stop, then pad 0
    (reinforce.py:177-184 + pad_sequence, reinforce_omics.py:304)."""
    g = torch.Generator().manual_seed(seed)
    toks = torch.zeros(T, B, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(1, max(2, T - 2), (1,), generator=g))
        if b == 0:
            n = T - 2  # one row with the maximum length, as pad_sequence guarantees
        toks[:n, b] = torch.randint(4, V, (n,), generator=g)
        toks[n:n + 2, b] = STOP
    return toks


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def elem_err(a, b, floor=1e-2):
    """Element-wise bar next to the max-norm one: the largest |a-b| / (|b| + floor*max|b|), i.e.
    |a-b| <= tol*|b| + tol*floor*max|b| for every element (tol=1e-4, floor=1e-2: the absolute floor
    is 1e-6 of the tensor's largest entry).  Returns (worst value, flat index of the worst element)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    e = np.abs(a - b) / (np.abs(b) + floor * np.abs(b).max() + 1e-300)
    i = int(e.argmax())
    return float(e[i]), i


def oracle_train(odec, tokens, h0, reward, scale, dtype=torch.float32):
    """Loss, logp and every gradient by torch autograd over the reference loop."""
    dec = odec if dtype == torch.float32 else _clone_as(odec, dtype)
    for p in dec.parameters():
        p.grad = None
    h0 = h0.detach().to(dtype).clone().requires_grad_(True)
    R = reward.to(dtype)
    B = tokens.shape[1]
    # reference: rl_loss -= mean(lp * reward) over B_valid rows == scale * sum
    loss = RL.reinforce_loss(dec, tokens, h0, R[:, None]) * (scale * B)
    loss.backward()
    with torch.no_grad():
        logp = RL.teacher_forced_logprobs(dec, tokens, h0.detach())
    grads = {n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None}
    return loss.item(), logp, grads, h0.grad.detach().clone()


def _clone_as(odec, dtype):
    from oracle.stack_gru import OracleStackGRUDecoder
    d = OracleStackGRUDecoder(odec.params, dtype=dtype)
    d.load_state_dict({k: v.to(dtype) for k, v in odec.state_dict().items()})
    return d.eval()


def product_train(pdec, tokens, h0, reward, scale, chunk=0):
    from paccmann_generator_amd import ops
    dev = pdec.output_layer.weight.device
    for p in pdec.parameters():
        p.grad = None
    h0d = h0.to(dev).clone().requires_grad_(True)
    loss, logp = ops.reinforce_loss(pdec.weights(), tokens.to(dev), h0d, reward.to(dev), scale, chunk=chunk)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().clone() for n, p in pdec.named_parameters() if p.grad is not None}
    return loss.item(), logp.cpu(), grads, h0d.grad.detach().cpu().clone()


def check_train(params, B, T, seed=0, tol=1e-4, verbose=False, chunk=0, elem_tol=1e-4, allow_noise=None):
    """Teacher-forced log-probs + REINFORCE gradients, HIP vs oracle autograd in fp64.

    Two bars per tensor: max-norm (max|a-b| / max|b| <= tol) and element-wise (elem_err <= elem_tol).
    An element that misses the element-wise bar is accepted only if the fp32 CPU oracle -- the same
    arithmetic in the same precision, another summation order -- is itself at least a third as far
    from the fp64 oracle there: then the miss is fp32 summation noise on a small entry, not a defect.
    Such entries are listed in report['noise'].  At the BASELINE configurations (CFG1 / CFG2 / CFG4
    model shapes) no such exception is accepted: there the element-wise bar holds outright (it did in
    every recorded run), and a miss fails the test.  ``allow_noise`` overrides that default."""
    if allow_noise is None:
        allow_noise = not any(all(params.get(k) == c[k] for k in c) for c in (CFG1, CFG2, CFG4))
    odec, pdec = make_pair(params, seed=seed)
    L, H = params['n_layers'], params['rnn_cell_size']
    tokens = synthetic_tokens(T, B, params['vocab_size'], seed=seed + 3)
    torch.manual_seed(seed + 1)
    h0 = torch.tanh(torch.randn(L, B, H))
    torch.manual_seed(seed + 2)
    reward = torch.rand(B)
    scale = 1.0 / B
    o_loss, o_logp, o_grads, o_gh0 = oracle_train(odec, tokens, h0, reward, scale, torch.float64)
    p_loss, p_logp, p_grads, p_gh0 = product_train(pdec, tokens, h0, reward, scale, chunk=chunk)
    pairs = {'logp': (p_logp, o_logp), 'grad_h0': (p_gh0, o_gh0)}
    for n, g in o_grads.items():
        if not n.startswith('latent_to_hidden'):
            pairs[n] = (p_grads[n], g)
    report = {'loss': abs(p_loss - o_loss) / (abs(o_loss) + 1e-30)}
    elem = {}
    for n, (a, b) in pairs.items():
        report[n] = rel_err(a, b)
        elem[n] = elem_err(a, b)
    if verbose:
        for k, v in report.items():
            print(f'  {k:40s} {v:.3e}  elem {elem.get(k, (0, 0))[0]:.3e}')
    bad = {k: v for k, v in report.items() if not (v <= tol)}
    assert not bad, f'parity failures (max-norm rel tol {tol}): {bad}'
    miss = {k: v for k, v in elem.items() if not (v[0] <= elem_tol)}
    noise = {}
    if miss:
        f_loss, f_logp, f_grads, f_gh0 = oracle_train(odec, tokens, h0, reward, scale, torch.float32)
        f32 = {'logp': f_logp, 'grad_h0': f_gh0, **f_grads}
        for k in miss:
            a = np.asarray(pairs[k][0], dtype=np.float64).reshape(-1)
            b = np.asarray(pairs[k][1], dtype=np.float64).reshape(-1)
            c = np.asarray(f32[k], dtype=np.float64).reshape(-1)
            den = np.abs(b) + 1e-2 * np.abs(b).max() + 1e-300
            e_hip, e_f32 = np.abs(a - b) / den, np.abs(c - b) / den
            over = e_hip > elem_tol
            real = over & (e_hip > 3.0 * np.maximum(e_f32, elem_tol / 3.0))
            assert not real.any(), (f'{k}: element-wise miss beyond fp32 summation noise at flat index '
                                    f'{int(np.argmax(real))}: hip {e_hip[real].max():.3e} vs fp32 oracle '
                                    f'{e_f32[real].max():.3e}')
            noise[k] = dict(worst=float(e_hip.max()), n=int(over.sum()), fp32_oracle_worst=float(e_f32.max()))
    report['elem'] = {k: v[0] for k, v in elem.items()}
    report['noise'] = noise
    assert allow_noise or not noise, f'element-wise bar missed at a BASELINE configuration (fp32 noise or not): {noise}'

    # parameters the loss cannot reach must stay without gradient
    assert all(not n.startswith('latent_to_hidden') for n in p_grads), p_grads.keys()
    return report


def check_greedy(params, B, gen_len, seed=0, gap_tol=1e-4):
    """Greedy decode: exact tokens.  A row may differ from the fp32 oracle only if, at its
    first differing position, the fp64 oracle's top-2 logit gap is below gap_tol (near tie)."""
    from paccmann_generator_amd import Generator, GreedySearch as PG
    odec, pdec = make_pair(params, seed=seed)
    torch.manual_seed(seed + 1)
    latent = torch.randn(1, B, params['latent_dim'])
    ogen = OracleGenerator(odec)
    ref = ogen.generate_raw(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len, GreedySearch())
    gen = Generator(None, pdec)
    tokens, lens = gen.generate_tokens(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len, PG(),
                                       early_exit=False)
    tokens = tokens.cpu()
    lens = lens.cpu()
    mism = (tokens != ref).any(dim=1).nonzero().flatten().tolist()
    near_ties = 0
    if mism:
        P64 = FM.params_from_decoder(odec, np.float64)
        h0 = odec.latent_to_hidden(latent.repeat(params['n_layers'], 1, 1)).detach().numpy().astype(np.float64)
        for b in mism:
            t = int((tokens[b] != ref[b]).nonzero()[0])
            # replay the product's own prefix in fp64 and look at the gap at position t
            gap = _gap_at(P64, h0[:, b:b + 1], tokens[b, :t].numpy(), START)
            assert gap < gap_tol, f'row {b} differs at t={t} with fp64 top-2 gap {gap}'
            near_ties += 1
    # lengths = tokens before the first stop
    for b in range(B):
        r = tokens[b].tolist()
        n = r.index(STOP) if STOP in r else gen_len
        assert int(lens[b]) == n, (b, int(lens[b]), n)
    return dict(rows=B, mismatched_rows=len(mism), near_ties=near_ties)


def _gap_at(P, h0, prefix, start):
    L = h0.shape[0]
    gtab = FM.gate_table(P)
    h = [h0[l].copy() for l in range(L)]
    S = np.zeros((1, P['D'], P['wp'].shape[0]))
    tok = np.array([start])
    for t in range(len(prefix) + 1):
        h, S, _ = FM.forward_step(P, gtab, tok, h, S)
        if t < len(prefix):
            tok = np.array([int(prefix[t])])
    lg = np.sort(FM.logits_of(P, h[L - 1])[0])
    return float(lg[-1] - lg[-2])


def check_sampling(params, B, gen_len, temperature=0.8, seed=0, philox_seed=1234, eps=2e-5):
    """Philox sampling: replay the product's tokens teacher-forced through the fp32 oracle and
    require every sampled token to be the inverse-CDF choice for the same uniform, except
    within eps of a CDF boundary.  Returns the token histogram for a chi-square check."""
    from paccmann_generator_amd import Generator, SamplingSearch as PS
    odec, pdec = make_pair(params, seed=seed)
    torch.manual_seed(seed + 1)
    latent = torch.randn(1, B, params['latent_dim'])
    gen = Generator(None, pdec)
    gen.seed(philox_seed)
    tokens, lens = gen.generate_tokens(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len,
                                       PS(temperature), early_exit=False)
    tokens = tokens.cpu().numpy()
    V = params['vocab_size']
    P = FM.params_from_decoder(odec, np.float32)
    L = params['n_layers']
    h0 = odec.latent_to_hidden(latent.repeat(L, 1, 1)).detach().numpy().astype(np.float32)
    gtab = FM.gate_table(P)
    h = [h0[l].copy() for l in range(L)]
    S = np.zeros((B, P['D'], P['wp'].shape[0]), dtype=np.float32)
    tok = np.full((B,), START, dtype=np.int64)
    exact = 0
    boundary = 0
    expected = np.zeros(V)
    for t in range(gen_len):
        h, S, _ = FM.forward_step(P, gtab, tok, h, S)
        logits = FM.logits_of(P, h[L - 1]).astype(np.float32)
        for b in range(B):
            u = uniform(philox_seed, b, t)
            choice, margin, probs = sample_row(logits[b], np.float32(temperature), u)
            expected += probs
            if choice == tokens[b, t]:
                exact += 1
            else:
                assert margin < eps, f'row {b} step {t}: got {tokens[b, t]}, oracle {choice}, margin {margin}'
                boundary += 1
        tok = tokens[:, t].astype(np.int64)
    hist = np.bincount(tokens.reshape(-1), minlength=V).astype(np.float64)
    return dict(exact=exact, boundary=boundary, hist=hist, expected=expected)


def check_step(params, B, seed=0, tol=1e-5):
    """decoder(tok, hidden, stack) single-step protocol vs the oracle module."""
    odec, pdec = make_pair(params, seed=seed)
    L, H = params['n_layers'], params['rnn_cell_size']
    torch.manual_seed(seed + 5)
    tok = torch.randint(0, params['vocab_size'], (1, B))
    hidden = torch.tanh(torch.randn(L, B, H))
    stack = torch.tanh(torch.randn(B, params['stack_depth'], params['stack_width']))
    with torch.no_grad():
        o_out, o_h, o_s = odec(tok, hidden, stack)
    p_out, p_h, p_s = pdec(tok.cuda(), hidden.cuda(), stack.cuda())
    rep = dict(out=rel_err(p_out.cpu(), o_out), hidden=rel_err(p_h.cpu(), o_h), stack=rel_err(p_s.cpu(), o_s))
    bad = {k: v for k, v in rep.items() if not (v <= tol)}
    assert not bad, bad
    return rep


def run_smoke():
    """__graft_entry__.smoke(): one tiny decode + REINFORCE step on cuda:0 vs the oracle."""
    assert torch.cuda.is_available(), 'smoke() needs a GPU'
    rep = check_train(TINY, B=20, T=9, seed=0)
    g = check_greedy(TINY, B=20, gen_len=12, seed=0)
    print('smoke ok: max rel err', max(v for v in rep.values() if isinstance(v, float)), 'greedy', g)
