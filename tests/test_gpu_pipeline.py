"""Host scoring pipelined against the GPU work of one policy_gradient step (SURVEY.md section 8 f1,
second half; Reinforce._policy_gradient_pipelined).  The reference runs decode -> validity -> reward
-> teacher-forced loop + backward strictly one after the other (reinforce.py:162-211,
reinforce_omics.py:295-341); with ``micro_batches=k`` the valid molecules are cut into k slices: their
teacher-forced passes run side by side on k streams while the host computes the rewards slice by slice,
and a slice's backward pass is queued the moment its rewards exist.  The result must be that of the
serial step: the same molecules, the same (mean reward, loss), gradients within 1e-6."""
import copy
import logging

import numpy as np
import pandas as pd
import pytest
import torch

from tests import parity as P
from tests.test_gpu_harness import ALPHABET, FakeEncoder, FakeLanguage, reward_fn

pytestmark = pytest.mark.gpu
PARAMS = dict(P.TINY, latent_dim=32)
HP = dict(generate_len=14, temperature=0.9, clip_grad=0.1, learning_rate=1e-3, save_untrained=False)
BATCH = 100       # about 58 valid molecules: four slices of 16 rows at k = 4


class TableLanguage(FakeLanguage):
    index_to_token = {i: ALPHABET[i % len(ALPHABET)] for i in range(PARAMS['vocab_size'])}


def _build(k, language, seed=0):
    from paccmann_generator_amd import Generator, ReinforceOmic
    _, pdec = P.make_pair(PARAMS, seed=seed)
    torch.manual_seed(5)
    enc = FakeEncoder(9, 16).cuda()
    gen = Generator(enc, pdec)
    gen._associate_language(language())
    gen.seed(33)
    rng = np.random.RandomState(0)
    df = pd.DataFrame({'cell_line': ['cl0'], 'gene_expression': [pd.Series(rng.randn(9).astype(np.float32))]})
    state = dict(cursor=0, seen=[], reward_sizes=[])

    def is_valid_batch(smiles):
        # called per step (serial) or per slice in row order: the cursor is the first row of the call
        base = state['cursor']
        state['cursor'] += len(smiles)
        state['seen'].extend(smiles)
        return [len(s) > 0 and (s.count('C') + s.count('N') + base + i) % 3 != 0 and not (13 <= base + i < 26)
                for i, s in enumerate(smiles)]

    def reward(smiles, target):
        state['reward_sizes'].append(len(smiles))
        return reward_fn(smiles, target)

    learner = ReinforceOmic(gen, enc, None, df, dict(HP), 'm', logging.getLogger('pipe'), reward_fn=reward,
                            is_valid_batch=is_valid_batch, micro_batches=k)
    return learner, state


def _two_steps(k, language):
    learner, state = _build(k, language)
    out = []
    for _ in range(2):      # the second step also carries the encoder gradient over (never zeroed)
        state['cursor'] = 0
        r, l = learner.policy_gradient('cl0', 1, BATCH)
        out.append((float(r), float(l)))
    torch.cuda.synchronize()
    g_dec = torch.cat([p.grad.reshape(-1) for p in learner.generator.decoder.parameters() if p.grad is not None])
    g_enc = torch.cat([p.grad.reshape(-1) for p in learner.encoder.mu.parameters()])
    w_dec = torch.cat([p.detach().reshape(-1) for p in learner.generator.decoder.parameters()])
    assert all(p.grad is None for p in learner.encoder.lv.parameters())
    return out, g_dec.cpu().double(), g_enc.cpu().double(), w_dec.cpu().double(), state


@pytest.mark.parametrize('language', [TableLanguage, FakeLanguage], ids=['device_detok', 'host_rows'])
def test_micro_batches_equal_the_serial_step(gpu, language):
    ref, g_dec, g_enc, w_dec, st1 = _two_steps(1, language)
    n_valid = st1['reward_sizes'][0]
    assert len(st1['reward_sizes']) == 2 and 48 < n_valid <= 64
    for k in (2, 4):
        out, gd, ge, wd, st = _two_steps(k, language)
        assert st['seen'] == st1['seen'], k                      # identical molecules, in order
        sizes = st['reward_sizes'][:len(st['reward_sizes']) // 2]      # first step: one reward call per slice
        assert len(sizes) == k and sum(sizes) == n_valid and all(n % 16 == 0 for n in sizes[:-1]), sizes
        for (r, l), (r1, l1) in zip(out, ref):
            assert r == r1                                       # same rewards, same order, same mean
            assert abs(l - l1) <= 1e-6 * abs(l1), (k, l, l1)
        assert float((gd - g_dec).abs().max()) <= 1e-6 * float(g_dec.abs().max()), k
        assert float((ge - g_enc).abs().max()) <= 1e-6 * float(g_enc.abs().max()), k
        assert float((wd - w_dec).abs().max()) <= 1e-2 * HP["learning_rate"], k


def test_pipelined_step_matches_oracle(gpu):
    """The pipelined step against the CPU restatement of reinforce_omics.py:276-342 (not only against
    the serial HIP step): parameters after one update."""
    from oracle import reinforce_loop as RL
    learner, state = _build(4, TableLanguage)
    odec, _ = P.make_pair(PARAMS, seed=0)
    enc_cpu = copy.deepcopy(learner.encoder).cpu()
    with torch.no_grad():
        latent = learner.encode_cell_line('cl0', BATCH)
    learner.generator.seed(33)
    smiles, nums, valid_idx = learner.get_smiles_from_latent(latent)
    rewards = reward_fn(smiles, 'cl0')
    learner.generator.seed(33)
    state['cursor'] = 0
    rew, loss = learner.policy_gradient('cl0', 1, BATCH)
    x = torch.unsqueeze(torch.Tensor(learner.gep_df.iloc[0].gene_expression.values), 0)
    opt = RL.make_optimizer(odec, dict(learning_rate=HP['learning_rate']))
    mu, _ = enc_cpu(x)
    o_rew, o_loss = RL.policy_gradient_step(odec, opt, torch.unsqueeze(mu.repeat(BATCH, 1), 0), [n.cpu() for n in nums],
                                            valid_idx, rewards, lrps=2, clip_grad=HP['clip_grad'],
                                            extra_clip_params=list(enc_cpu.parameters()))
    assert abs(float(rew) - float(o_rew)) < 1e-6
    assert abs(loss - o_loss) <= 1e-4 * abs(o_loss)
    for (n, po), (_, pp) in zip(odec.named_parameters(), learner.generator.decoder.named_parameters()):
        assert float((pp.detach().cpu() - po.detach()).abs().max()) < 0.02 * HP['learning_rate'], n
    for (n, po), (_, pp) in zip(enc_cpu.named_parameters(), learner.encoder.named_parameters()):
        if po.grad is not None:
            assert P.rel_err(pp.grad.cpu(), po.grad) < 1e-4, n


def test_no_valid_molecule_and_reward_errors(gpu):
    learner, state = _build(3, TableLanguage)
    learner.is_valid_batch = lambda smiles: [False] * len(smiles)
    before = torch.cat([p.detach().reshape(-1).clone() for p in learner.generator.decoder.parameters()])
    r, l = learner.policy_gradient('cl0', 1, BATCH)
    assert (float(r), l) == (0.0, 0.0)
    assert torch.equal(before, torch.cat([p.detach().reshape(-1) for p in learner.generator.decoder.parameters()]))

    calls = []

    def boom(smiles, target):      # the second slice's reward fails while the first slice's backward is in flight
        calls.append(len(smiles))
        if len(calls) == 2:
            raise ValueError('reward backend died')
        return reward_fn(smiles, target)
    learner.is_valid_batch = lambda smiles: [True] * len(smiles)
    learner.reward_fn = boom
    with pytest.raises(ValueError, match='reward backend died'):
        learner.policy_gradient('cl0', 1, BATCH)
    torch.cuda.synchronize()
    # and the learner is usable afterwards
    learner.reward_fn = reward_fn
    r, l = learner.policy_gradient('cl0', 1, BATCH)
    assert np.isfinite(l) and l != 0.0
