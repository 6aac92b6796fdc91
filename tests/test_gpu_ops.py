"""torch.library operators on the GPU: opcheck (schema, fake tensors, autograd registration, AOT
dispatch), the chunked training pair (PMG_FLAG_ACCUMULATE / PMG_FLAG_REUSE_PACK) against the
single-call one, and BASELINE.json configs[3] (2 layers, hidden 1024) against the oracle at a
length that crosses every stack regime and, at its full size B=16384 / T=258, through
size-independent properties."""
import pytest
import torch

from tests import parity as P

pytestmark = pytest.mark.gpu


def _inputs(params, B, T, seed=0):
    _, pdec = P.make_pair(params, seed=seed)
    tokens = P.synthetic_tokens(T, B, params['vocab_size'], seed=seed + 3).cuda()
    torch.manual_seed(seed + 1)
    h0 = torch.tanh(torch.randn(params['n_layers'], B, params['rnn_cell_size'], device='cuda'))
    reward = torch.rand(B, device='cuda')
    return pdec, tokens, h0, reward


def test_opcheck_all_operators(gpu):
    from paccmann_generator_amd import ops
    params = dict(P.TINY, n_layers=2)
    pdec, tokens, h0, reward = _inputs(params, 19, 7)
    w = pdec.weights()
    flat = w.flat()
    D = params['stack_depth']
    h0g = h0.clone().requires_grad_(True)
    torch.library.opcheck(torch.ops.pmg.reinforce_loss, (tokens, h0g, reward, 1.0 / 19, flat, D, 0, 0, True))
    torch.library.opcheck(torch.ops.pmg.reinforce_loss, (tokens, h0g, reward, 1.0 / 19, flat, D, 0, 16, True))
    nog = [p.detach() for p in flat]
    torch.library.opcheck(torch.ops.pmg.reinforce_loss, (tokens, h0, reward, 1.0, nog, D, 0, 0, False))
    torch.library.opcheck(torch.ops.pmg.logprob_fwd, (tokens, h0, nog, D, 0))
    logp, ws = torch.ops.pmg.logprob_fwd(tokens, h0, nog, D, 0)
    # reinforce_bwd writes the scratch regions of `ws` (declared in its schema); torch's
    # functionalisation does not take an operator that both mutates an input and returns a Tensor[],
    # so the AOT-dispatch leg is run on the fused pmg::reinforce_loss only
    torch.library.opcheck(torch.ops.pmg.reinforce_bwd, (tokens, reward, 1.0 / 19, ws, nog, D, 0),
                          test_utils=('test_schema', 'test_faketensor', 'test_autograd_registration'))
    latent = torch.randn(19, params['latent_dim'], device='cuda')
    l2h = [w.t['w_l2h'].detach(), w.t['b_l2h'].detach()]
    torch.library.opcheck(torch.ops.pmg.decode, (latent, nog, *l2h, D, 0, P.START, P.STOP, 11, 0.8, 5, 0, 0, True))
    stack = torch.tanh(torch.randn(19, D, params['stack_width'], device='cuda'))
    torch.library.opcheck(torch.ops.pmg.step, (tokens[0], h0, stack, nog, D, 0, True))
    torch.library.opcheck(torch.ops.pmg.step, (tokens[0], h0, None, nog, D, 0, False))
    tok, lens = torch.ops.pmg.decode(latent, nog, *l2h, D, 0, P.START, P.STOP, 11, 0.8, 5, 0, 0, True)
    table = ops.TokenTable({i: 'ab'[i % 2] * (1 + i % 3) for i in range(params['vocab_size'])}, 'cuda')
    out_off, out_bytes = table.buffers(19, 11)
    torch.library.opcheck(torch.ops.pmg.detokenize, (tok, lens, table.tab_off, table.tab_bytes, table.max_tok_bytes,
                                                     out_off, out_bytes))


def test_registered_operator_drives_the_parity_check(gpu):
    """tests/parity.check_train goes through ops.reinforce_loss == torch.ops.pmg.reinforce_loss; the
    two-operator form (logprob_fwd + reinforce_bwd) gives the same numbers."""
    from paccmann_generator_amd import ops
    P.check_train(P.TINY, B=20, T=9)
    pdec, tokens, h0, reward = _inputs(P.TINY, 20, 9)
    w = pdec.weights()
    flat = [p.detach() for p in w.flat()]
    D = P.TINY['stack_depth']
    loss, logp, gh0, grads = torch.ops.pmg.reinforce_loss(tokens, h0, reward, 0.05, flat, D, 0, 0, True)
    logp2, ws = torch.ops.pmg.logprob_fwd(tokens, h0, flat, D, 0)
    loss2, gh02, grads2 = torch.ops.pmg.reinforce_bwd(tokens, reward, 0.05, ws, flat, D, 0)
    assert torch.equal(logp, logp2) and torch.equal(gh0, gh02) and float(loss) == float(loss2)
    assert all(torch.equal(a, b) for a, b in zip(grads, grads2))
    # the autograd formula scales the gradients formed in the forward pass by d(loss)
    h0g = h0.clone().requires_grad_(True)
    l3, _ = ops.reinforce_loss(w, tokens, h0g, reward, 0.05)
    (3.0 * l3).backward()
    assert torch.allclose(h0g.grad, 3.0 * gh0, rtol=1e-6, atol=0)
    assert torch.allclose(pdec.output_layer.weight.grad, 3.0 * grads[-2], rtol=1e-6, atol=0)
    with pytest.raises(Exception):       # out-of-range ids are refused on the host
        ops.reinforce_loss(w, tokens.clamp(min=0) + 1000, h0, reward, 0.05)


@pytest.mark.parametrize('params,B,T,chunk', [(P.TINY, 50, 9, 16), (dict(P.TINY, n_layers=2), 45, 6, 32),
                                              (P.CFG4, 40, 20, 16)], ids=['tiny', 'tiny_l2', 'cfg4shape'])
def test_chunked_accumulate_equals_single_call(gpu, params, B, T, chunk):
    """One call over B sequences == chunks of `chunk` sequences with PMG_FLAG_ACCUMULATE on chunks
    2..n (gradients AND loss accumulate; packed weights reused)."""
    from paccmann_generator_amd import ops
    pdec, tokens, h0, reward = _inputs(params, B, T)
    flat = [p.detach() for p in pdec.weights().flat()]
    D = params['stack_depth']
    assert len(ops.plan_chunks(pdec.weights().dims, B, T, chunk=chunk)) >= 2
    l1, lp1, g1, gs1 = torch.ops.pmg.reinforce_loss(tokens, h0, reward, 1.0 / B, flat, D, 0, 0, True)
    l2, lp2, g2, gs2 = torch.ops.pmg.reinforce_loss(tokens, h0, reward, 1.0 / B, flat, D, 0, chunk, True)
    assert torch.equal(lp1, lp2) and torch.equal(g1, g2)        # per-sequence results: bit for bit
    assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1))
    for a, b in zip(gs1, gs2):                                  # sums over the batch: another order
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-12


def test_cfg4_shape_full_stack_regimes_vs_oracle(gpu):
    """configs[3] model shape (2 layers, hidden 1024, stack 50x50) at T = 130 > 2 * stack depth:
    rows still zero, all rows live, rows already dead.  logp + every gradient vs the fp64 oracle,
    once in one call and once chunked."""
    P.check_train(P.CFG4, B=18, T=130)
    P.check_train(P.CFG4, B=34, T=60, chunk=16, seed=1)


def test_cfg4_shape_greedy_exact(gpu):
    r = P.check_greedy(P.CFG4, 20, 110)
    assert r['mismatched_rows'] == 0


def test_cfg4_full_size_properties(gpu):
    """configs[3] at its stated size (B=16384, T=256+2, 2 layers, hidden 1024): the training pair
    runs as chunks (80 GB of history per 4096 sequences).  Properties that need no oracle: finite,
    run-to-run bit-identical, independent of the chunking (4 x 4096 vs 8 x 2048), loss equals the
    reward-weighted sum of the log-probs, log-probs of a slice equal the rows of the whole."""
    from paccmann_generator_amd import ops
    from paccmann_generator_amd.reinforce import pad_rows_on_device
    _, pdec = P.make_pair(P.CFG4)
    B, T = 16384, 256
    torch.manual_seed(1)
    latent = torch.randn(B, P.CFG4['latent_dim'], device='cuda')
    w = pdec.weights()
    tok, lens = ops.decode(w, latent, P.START, P.STOP, T, 0.8, seed=1234, early_exit=False)
    a, al = ops.decode(w, latent[5000:9000].contiguous(), P.START, P.STOP, T, 0.8, seed=1234, seq_index_base=5000,
                       early_exit=False)
    assert torch.equal(tok[5000:9000], a) and torch.equal(lens[5000:9000], al)
    padded = pad_rows_on_device(tok, lens, P.STOP, t_max=T + 2)
    del tok, a
    with torch.no_grad():
        h0 = pdec.latent_to_hidden(latent.unsqueeze(0).repeat(2, 1, 1)).contiguous()
    torch.manual_seed(2)
    R = torch.rand(B, device='cuda')
    flat = [p.detach() for p in w.flat()]
    D = P.CFG4['stack_depth']
    plan = ops.plan_chunks(w.dims, B, T + 2)
    assert len(plan) >= 4

    def run(chunk):
        loss, logp, gh0, grads = torch.ops.pmg.reinforce_loss(padded, h0, R, 1.0 / B, flat, D, 0, chunk, True)
        return float(loss), logp, gh0, torch.cat([g.reshape(-1) for g in grads])

    l1, lp1, gh1, g1 = run(0)
    l2, lp2, gh2, g2 = run(0)
    assert torch.isfinite(g1).all() and torch.isfinite(lp1).all() and torch.isfinite(gh1).all()
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(lp1, lp2) and torch.equal(gh1, gh2)
    del lp2, gh2, g2
    l3, lp3, gh3, g3 = run(2048)
    assert torch.equal(lp1, lp3) and torch.equal(gh1, gh3)
    assert abs(l1 - l3) <= 1e-5 * abs(l1)
    assert float((g1 - g3).abs().max()) <= 1e-5 * float(g1.abs().max())
    ref = -float((lp1.double() * R.double().unsqueeze(0)).sum()) / B
    assert abs(l1 - ref) <= 1e-5 * abs(ref)
    sl = slice(4096 + 16, 4096 + 16 + 512)
    _, lps, _, _ = torch.ops.pmg.reinforce_loss(padded[:, sl].contiguous(), h0[:, sl].contiguous(), R[sl].contiguous(),
                                                1.0 / B, flat, D, 0, 0, False)
    assert torch.equal(lps, lp1[:, sl])
    torch.cuda.empty_cache()
