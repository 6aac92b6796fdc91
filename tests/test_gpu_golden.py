"""HIP path (through the C ABI) against the committed golden vectors (tests/golden/*.npz).

Bars: greedy tokens bit-exact; sampled tokens exact except within 2e-5 of a CDF boundary;
log-probs, loss and every gradient within 1e-4 relative of the float64 golden values."""
import numpy as np
import pytest
import torch

from tests import golden_util as GU
from tests.parity import START, STOP, rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('forward_path')]
TOL = 1e-4


def _product(params, g):
    from paccmann_generator_amd import StackGRUDecoder
    dec = StackGRUDecoder(params)
    dec.load_state_dict(GU.state_dict(g))
    return dec.cuda().eval()


@pytest.mark.parametrize('name', GU.CASES)
def test_greedy_tokens_bit_exact(gpu, name):
    from paccmann_generator_amd import Generator, GreedySearch
    params, g = GU.load(name)
    gen = Generator(None, _product(params, g))
    ref = g['dec.greedy_tokens']
    tokens, lens = gen.generate_tokens(torch.from_numpy(g['dec.latent']), torch.tensor([START]),
                                       torch.tensor([STOP]), ref.shape[1], GreedySearch(), early_exit=False)
    assert (tokens.cpu().numpy() == ref).all()
    for b, row in enumerate(ref.tolist()):
        assert int(lens[b]) == (row.index(STOP) if STOP in row else len(row))


@pytest.mark.parametrize('name', ['tiny_l1', 'tiny_l2'])
def test_sampled_tokens_match_golden_draws(gpu, name):
    from paccmann_generator_amd import Generator, SamplingSearch
    params, g = GU.load(name)
    gen = Generator(None, _product(params, g))
    gen.seed(int(g['dec.philox_seed']))
    ref, margin = g['dec.sampled_tokens'], g['dec.sampled_margin']
    tokens, _ = gen.generate_tokens(torch.from_numpy(g['dec.latent']), torch.tensor([START]), torch.tensor([STOP]),
                                    ref.shape[1], SamplingSearch(float(g['dec.temperature'])), early_exit=False)
    got = tokens.cpu().numpy()
    for b in range(ref.shape[0]):
        diff = np.nonzero(got[b] != ref[b])[0]
        if diff.size:  # a row may leave the golden path only at a draw that sits on a CDF boundary
            assert margin[b, diff[0]] < 2e-5, (b, int(diff[0]), float(margin[b, diff[0]]))
    assert (got != ref).any(axis=1).sum() <= 1


@pytest.mark.parametrize('name', GU.CASES)
def test_logprobs_and_gradients(gpu, name):
    from paccmann_generator_amd import ops
    params, g = GU.load(name)
    dec = _product(params, g)
    tokens = torch.from_numpy(g['train.tokens']).cuda()
    h0 = torch.from_numpy(g['train.h0']).cuda().requires_grad_(True)
    reward = torch.from_numpy(g['train.reward']).cuda()
    loss, logp = ops.reinforce_loss(dec.weights(), tokens, h0, reward, float(g['train.scale']))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g['train.loss'])) <= TOL * abs(float(g['train.loss']))
    assert rel_err(logp.cpu().numpy(), g['train.logp']) < TOL
    assert rel_err(h0.grad.cpu().numpy(), g['train.grad_h0']) < TOL
    got = {n: p.grad for n, p in dec.named_parameters() if p.grad is not None}
    ref = GU.grads(g)
    assert set(got) == set(ref)
    for n, r in ref.items():
        assert rel_err(got[n].cpu().numpy(), r) < TOL, n


@pytest.mark.parametrize('name', GU.COMPACT)
def test_compact_cases_cfg2_cfg4_shapes(gpu, name):
    """configs[1] (hidden 512) and configs[3] (2 layers, hidden 1024) model shapes at T=130 against
    frozen vectors: greedy tokens bit-exact, log-probs / grad_h0 in full, every parameter gradient
    through 4096 samples + its L2 norm and sum."""
    from paccmann_generator_amd import Generator, GreedySearch, StackGRUDecoder, ops
    params, g = GU.load(name)
    odec = GU.compact_decoder(params, g)
    dec = StackGRUDecoder(params)
    dec.load_state_dict(odec.state_dict())
    dec = dec.cuda().eval()
    ref = g['dec.greedy_tokens']
    tokens, lens = Generator(None, dec).generate_tokens(torch.from_numpy(g['dec.latent']), torch.tensor([START]),
                                                        torch.tensor([STOP]), ref.shape[1], GreedySearch(),
                                                        early_exit=False)
    assert (tokens.cpu().numpy() == ref).all()
    tok = torch.from_numpy(g['train.tokens']).cuda()
    h0 = torch.from_numpy(g['train.h0']).cuda().requires_grad_(True)
    reward = torch.from_numpy(g['train.reward']).cuda()
    loss, logp = ops.reinforce_loss(dec.weights(), tok, h0, reward, float(g['train.scale']))
    loss.backward()
    assert abs(loss.item() - float(g['train.loss'])) <= TOL * abs(float(g['train.loss']))
    assert rel_err(logp.cpu().numpy(), g['train.logp']) < TOL
    assert rel_err(h0.grad.cpu().numpy(), g['train.grad_h0']) < TOL
    got = {n: p.grad for n, p in dec.named_parameters() if p.grad is not None}
    refg = GU.compact_grads(g)
    assert set(got) == set(refg)
    for n, (samples, stat) in refg.items():
        GU.check_compact_grad(n, got[n].cpu().numpy(), samples, stat, TOL)
