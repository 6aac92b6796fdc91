"""The CPU oracle (both forms) against the committed golden vectors.  No GPU, no product compute."""
import numpy as np
import pytest
import torch

from oracle import fused_model as FM
from oracle import reinforce_loop as RL
from oracle.stack_gru import GreedySearch, OracleGenerator, OracleStackGRUDecoder
from tests import golden_util as GU
from tests.parity import START, STOP, rel_err


def _decoder(params, g, dtype=torch.float32):
    dec = OracleStackGRUDecoder(params, dtype=dtype)
    dec.load_state_dict(GU.state_dict(g, dtype))
    return dec.eval()


@pytest.mark.parametrize('name', GU.CASES)
def test_torch_oracle_reproduces_golden(name):
    params, g = GU.load(name)
    dec = _decoder(params, g)
    latent = torch.from_numpy(g['dec.latent'])
    gen_len = g['dec.greedy_tokens'].shape[1]
    raw = OracleGenerator(dec).generate_raw(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len,
                                            GreedySearch())
    assert (raw.numpy() == g['dec.greedy_tokens']).all()
    assert (g['dec.greedy_tokens'] == g['dec.greedy_tokens_fp64']).all()
    tokens = torch.from_numpy(g['train.tokens'])
    h0 = torch.from_numpy(g['train.h0']).requires_grad_(True)
    reward = torch.from_numpy(g['train.reward'])
    loss = RL.reinforce_loss(dec, tokens, h0, reward[:, None])
    loss.backward()
    assert abs(loss.item() - float(g['train.loss'])) <= 1e-5 * abs(float(g['train.loss']))
    with torch.no_grad():
        logp = RL.teacher_forced_logprobs(dec, tokens, h0.detach())
    assert rel_err(logp.numpy(), g['train.logp']) < 1e-5
    assert rel_err(h0.grad.numpy(), g['train.grad_h0']) < 1e-5
    for n, ref in GU.grads(g).items():
        got = dict(dec.named_parameters())[n].grad
        assert rel_err(got.numpy(), ref) < 1e-5, n


@pytest.mark.parametrize('name', ['tiny_l1', 'tiny_l2'])
def test_hand_bptt_model_reproduces_golden(name):
    """oracle/fused_model.py (the kernels' line-by-line CPU twin) in float64."""
    params, g = GU.load(name)
    dec = _decoder(params, g, torch.float64)
    P = FM.params_from_decoder(dec)
    out = FM.forward_backward(P, g['train.tokens'], g['train.h0'].astype(np.float64),
                              g['train.reward'].astype(np.float64), float(g['train.scale']))
    assert abs(out['loss'] - float(g['train.loss'])) < 1e-10
    assert rel_err(out['logp'], g['train.logp']) < 1e-10
    assert rel_err(out['grad_h0'], g['train.grad_h0']) < 1e-10
    G, ref = out['grads'], GU.grads(g)
    L = params['n_layers']
    pairs = [('embedding.weight', G['emb']), ('stack_controls_layer.weight', G['wc']),
             ('stack_controls_layer.bias', G['bc']), ('stack_input_layer.weight', G['wp']),
             ('stack_input_layer.bias', G['bp']), ('output_layer.weight', G['wo']), ('output_layer.bias', G['bo'])]
    for l in range(L):
        pairs += [(f'gru.weight_ih_l{l}', G['w_ih'][l]), (f'gru.weight_hh_l{l}', G['w_hh'][l]),
                  (f'gru.bias_ih_l{l}', G['b_ih'][l]), (f'gru.bias_hh_l{l}', G['b_hh'][l])]
    for n, got in pairs:
        assert rel_err(got, ref[n]) < 1e-10, n
    assert not any(n.startswith('latent_to_hidden') for n in ref)
    # greedy decode of the numpy model in fp32 equals the golden tokens
    P32 = FM.params_from_decoder(_decoder(params, g), np.float32)
    dec32 = _decoder(params, g)
    latent = torch.from_numpy(g['dec.latent'])
    h0 = dec32.latent_to_hidden(latent.repeat(L, 1, 1)).detach().numpy()
    toks = FM.greedy_decode(P32, h0, START, g['dec.greedy_tokens'].shape[1])
    assert (toks == g['dec.greedy_tokens']).all()


def test_golden_fp32_noise_floor_is_far_below_tolerance():
    """The 1e-4 parity bar must not be within reach of plain fp32 round-off."""
    for name in GU.CASES:
        _, g = GU.load(name)
        assert float(g['train.fp32_autograd_max_rel_err']) < 1e-5


@pytest.mark.parametrize('name', GU.COMPACT)
def test_compact_cases_are_reproduced_by_the_fp32_oracle(name):
    """configs[1] / configs[3] model shapes at T=130: weights from the seed (checksum pinned), greedy
    tokens, log-probs, grad_h0 in full, parameter gradients as samples + statistics."""
    params, g = GU.load(name)
    dec = GU.compact_decoder(params, g)
    latent = torch.from_numpy(g['dec.latent'])
    gen_len = g['dec.greedy_tokens'].shape[1]
    raw = OracleGenerator(dec).generate_raw(latent, torch.tensor([START]), torch.tensor([STOP]), gen_len,
                                            GreedySearch())
    assert (raw.numpy() == g['dec.greedy_tokens']).all()
    assert (g['dec.greedy_tokens'] == g['dec.greedy_tokens_fp64']).all()
    tokens = torch.from_numpy(g['train.tokens'])
    h0 = torch.from_numpy(g['train.h0']).requires_grad_(True)
    reward = torch.from_numpy(g['train.reward'])
    loss = RL.reinforce_loss(dec, tokens, h0, reward[:, None])
    loss.backward()
    assert abs(loss.item() - float(g['train.loss'])) <= 1e-5 * abs(float(g['train.loss']))
    assert rel_err(h0.grad.numpy(), g['train.grad_h0']) < 1e-5
    named = dict(dec.named_parameters())
    ref = GU.compact_grads(g)
    assert set(ref) == {n for n, p in named.items() if p.grad is not None}
    for n, (samples, stat) in ref.items():
        GU.check_compact_grad(n, named[n].grad.numpy(), samples, stat, 1e-5)
    assert float(g['train.fp32_autograd_max_rel_err']) < 1e-5
