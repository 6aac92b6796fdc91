"""CPU tests of the oracle itself (no GPU, no product compute calls)."""
import numpy as np
import torch

from oracle import fused_model as FM
from oracle import reinforce_loop as RL
from oracle.philox import philox4x32_10, sample_row, uniform
from oracle.stack_gru import GreedySearch, OracleGenerator, SamplingSearch, build_decoder

SMALL = dict(embedding_size=8, rnn_cell_size=16, vocab_size=11, stack_width=5, stack_depth=6, latent_dim=12)


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_uniform_open_interval():
    us = np.array([uniform(7, s, t) for s in range(50) for t in range(20)])
    assert us.min() > 0.0 and us.max() < 1.0
    assert abs(us.mean() - 0.5) < 0.05


def test_sampler_matches_distribution():
    rng = np.random.default_rng(0)
    logits = rng.normal(size=23).astype(np.float32)
    counts = np.zeros(23)
    n = 4000
    for i in range(n):
        c, _, probs = sample_row(logits, 0.8, uniform(1, i, 0))
        counts[c] += 1
    chi2 = ((counts - n * probs) ** 2 / (n * probs)).sum()
    assert chi2 < 60.0, chi2  # 22 dof, p ~ 1e-5


def test_stepwise_matches_whole_sequence_gru():
    """The step-wise decoder is torch's nn.GRU: feeding a sequence at once gives the same h."""
    dec = build_decoder(dict(SMALL, n_layers=2), seed=0, dtype=torch.float64)
    torch.manual_seed(1)
    x = torch.randn(5, 3, SMALL['embedding_size'] + SMALL['stack_width'], dtype=torch.float64)
    h = torch.randn(2, 3, 16, dtype=torch.float64)
    whole, hw = dec.gru(x, h)
    hs = h
    outs = []
    for t in range(5):
        o, hs = dec.gru(x[t:t + 1], hs)
        outs.append(o)
    assert torch.allclose(torch.cat(outs), whole, atol=1e-12)
    assert torch.allclose(hs, hw, atol=1e-12)


def test_stack_push_pop_semantics():
    """controls are read as (push, pop, no-op): a pure push shifts down and writes the top."""
    dec = build_decoder(SMALL, seed=0)
    prev = torch.arange(2 * 6 * 5, dtype=torch.float32).reshape(2, 6, 5)
    val = torch.full((2, 1, 5), -1.0)
    push = dec.stack_augmentation(val, prev, torch.tensor([[1.0, 0, 0]] * 2))
    assert torch.equal(push[:, 0], val[:, 0]) and torch.equal(push[:, 1:], prev[:, :-1])
    pop = dec.stack_augmentation(val, prev, torch.tensor([[0, 1.0, 0]] * 2))
    assert torch.equal(pop[:, :-1], prev[:, 1:]) and torch.equal(pop[:, -1], torch.zeros(2, 5))
    noop = dec.stack_augmentation(val, prev, torch.tensor([[0, 0, 1.0]] * 2))
    assert torch.equal(noop, prev)


def _grad_check(L):
    params = dict(SMALL, n_layers=L)
    dec = build_decoder(params, seed=0, dtype=torch.float64)
    B, T = 5, 7
    torch.manual_seed(1)
    tokens = torch.randint(0, 11, (T, B))
    tokens[-2:, 0] = 0
    h0 = torch.randn(L, B, 16, dtype=torch.float64, requires_grad=True)
    R = torch.rand(B, dtype=torch.float64)
    loss = RL.reinforce_loss(dec, tokens, h0, R[:, None])
    loss.backward()
    out = FM.forward_backward(FM.params_from_decoder(dec), tokens.numpy(), h0.detach().numpy(), R.numpy(), 1.0 / B)
    assert abs(out['loss'] - loss.item()) < 1e-12

    def close(a, b):
        assert np.abs(a - b.numpy()).max() <= 1e-12 * max(1.0, np.abs(b.numpy()).max())

    G = out['grads']
    close(out['grad_h0'], h0.grad)
    close(G['emb'], dec.embedding.weight.grad)
    for l in range(L):
        close(G['w_ih'][l], getattr(dec.gru, f'weight_ih_l{l}').grad)
        close(G['w_hh'][l], getattr(dec.gru, f'weight_hh_l{l}').grad)
        close(G['b_ih'][l], getattr(dec.gru, f'bias_ih_l{l}').grad)
        close(G['b_hh'][l], getattr(dec.gru, f'bias_hh_l{l}').grad)
    close(G['wc'], dec.stack_controls_layer.weight.grad)
    close(G['bc'], dec.stack_controls_layer.bias.grad)
    close(G['wp'], dec.stack_input_layer.weight.grad)
    close(G['bp'], dec.stack_input_layer.bias.grad)
    close(G['wo'], dec.output_layer.weight.grad)
    close(G['bo'], dec.output_layer.bias.grad)
    lp = RL.teacher_forced_logprobs(dec, tokens, h0.detach())
    close(out['logp'], lp.detach())


def test_hand_bptt_matches_autograd_1_layer():
    _grad_check(1)


def test_hand_bptt_matches_autograd_2_layers():
    _grad_check(2)


def test_padding_row_has_no_embedding_gradient():
    dec = build_decoder(SMALL, seed=0, dtype=torch.float64)
    tokens = torch.zeros(4, 3, dtype=torch.long)
    h0 = torch.randn(1, 3, 16, dtype=torch.float64)
    RL.reinforce_loss(dec, tokens, h0, torch.ones(3, 1, dtype=torch.float64)).backward()
    assert torch.all(dec.embedding.weight.grad[0] == 0)


def test_generate_crops_like_the_harness_expects():
    """Start token stripped, cut before the first end token (consumed at reinforce.py:172-187)."""
    dec = build_decoder(SMALL, seed=0)
    gen = OracleGenerator(dec)
    torch.manual_seed(0)
    latent = torch.randn(1, 6, 12)
    raw = gen.generate_raw(latent, torch.tensor([2]), torch.tensor([3]), 9, GreedySearch())
    rows = list(gen.generate(latent, torch.tensor([2]), torch.tensor([3]), 9, GreedySearch()))
    assert len(rows) == 6
    for r, full in zip(rows, raw):
        f = full.tolist()
        n = f.index(3) if 3 in f else 9
        assert r.tolist() == f[:n]
    greedy_np = FM.greedy_decode(FM.params_from_decoder(dec, np.float32),
                                 dec.latent_to_hidden(latent).detach().numpy(), 2, 9)
    assert (greedy_np == raw.numpy()).all()


def test_policy_gradient_step_runs_and_updates_only_decoder():
    params = dict(SMALL)
    dec = build_decoder(params, seed=0)
    gen = OracleGenerator(dec)
    opt = RL.make_optimizer(dec, {})
    torch.manual_seed(0)
    latent = torch.randn(1, 8, 12)
    mols = list(gen.generate(latent, torch.tensor([2]), torch.tensor([3]), 9, SamplingSearch(0.8)))
    nums = RL.append_stops(mols, 3)
    valid_idx = [0, 2, 3, 5, 7]
    before = dec.output_layer.weight.detach().clone()
    r, loss = RL.policy_gradient_step(dec, opt, latent, [nums[i] for i in valid_idx], valid_idx,
                                      np.random.default_rng(0).random(5), lrps=1, clip_grad=0.1)
    assert np.isfinite(loss) and 0 <= float(r) <= 1
    assert not torch.equal(before, dec.output_layer.weight)
