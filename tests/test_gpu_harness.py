"""Harness level: the product's ReinforceOmic / ReinforceProtein / ReinforceMultiModalSets drive a
whole policy_gradient step (sample -> validity -> reward -> teacher-forced loss -> backward -> clip
-> Adam) on the GPU and must land on the parameters the CPU oracle's restatement of
reinforce_omics.py:276-342 reaches from the same sampled molecules.  Fakes stand in for what is out
of scope (SMILES language, encoder network, RDKit validity, reward), as the reference's own tests do
with FakeOrganDB (drug_evaluators/tests/test_organdb.py:9-21)."""
import copy
import logging
import os

import numpy as np
import pandas as pd
import pytest
import torch
import torch.nn as nn

from oracle import reinforce_loop as RL
from tests import parity as P

pytestmark = pytest.mark.gpu
ALPHABET = 'xx^$CNOSFclnos()=#123456[]+-@H.'


class FakeLanguage:
    start_index, stop_index = P.START, P.STOP

    def token_indexes_to_smiles(self, idx):
        return ''.join(ALPHABET[i % len(ALPHABET)] for i in idx)

    def selfies_to_smiles(self, s):
        return s


class FakeEncoder(nn.Module):
    """encoder(x) -> (mu, logvar); latent_size = Z/2 exercises the latent.repeat(1,1,2) branch
    (reinforce.py:160-161, reinforce_omics.py:310-312)."""

    def __init__(self, n_in, latent_size):
        super().__init__()
        self.latent_size = latent_size
        self.mu = nn.Linear(n_in, latent_size)
        self.lv = nn.Linear(n_in, latent_size)

    def forward(self, x):
        return self.mu(x), self.lv(x)

    def save(self, path, *a, **k):
        torch.save(self.state_dict(), path)


def is_valid(s):
    return len(s) > 0 and (s.count('C') + s.count('N')) % 3 != 0


def reward_fn(smiles, target):
    return np.array([min(1.0, (s.count('C') + 1) / 6.0) for s in smiles], dtype=np.float32)


PARAMS = dict(P.TINY, latent_dim=32)
HP = dict(generate_len=14, temperature=0.9, clip_grad=0.1, learning_rate=1e-3, batch_size=40, epochs=2, steps=3,
          eval_batch_size=24)


def _setup(tmp_path, cls_name='omic'):
    from paccmann_generator_amd import Generator, ReinforceOmic, ReinforceProtein
    odec, pdec = P.make_pair(PARAMS, seed=0)
    torch.manual_seed(5)
    enc_cpu = FakeEncoder(9, 16)
    enc_gpu = copy.deepcopy(enc_cpu).cuda()
    gen = Generator(enc_gpu, pdec)
    gen._associate_language(FakeLanguage())
    hp = dict(HP, model_folder=str(tmp_path / 'biased_models'))
    log = logging.getLogger('test')
    rng = np.random.RandomState(0)
    if cls_name == 'omic':
        df = pd.DataFrame({'cell_line': ['cl0', 'cl1', 'cl2'],
                           'gene_expression': [pd.Series(rng.randn(9).astype(np.float32)) for _ in range(3)]})
        learner = ReinforceOmic(gen, enc_gpu, None, df, hp, 'm', log, is_valid=is_valid, reward_fn=reward_fn)
    else:
        df = pd.DataFrame(rng.randn(3, 9).astype(np.float32), index=['p0', 'p1', 'p2'])
        learner = ReinforceProtein(gen, enc_gpu, None, df, hp, 'm', log, is_valid=is_valid, reward_fn=reward_fn)
    return learner, odec, enc_cpu, df


def _oracle_step(odec, enc_cpu, x, B, nums, valid_idx, rewards):
    opt = RL.make_optimizer(odec, dict(learning_rate=HP['learning_rate']))
    mu, _ = enc_cpu(x)
    latent = torch.unsqueeze(mu.repeat(B, 1), 0)
    return RL.policy_gradient_step(odec, opt, latent, [n.cpu() for n in nums], valid_idx, rewards, lrps=2,
                                   clip_grad=HP['clip_grad'], extra_clip_params=list(enc_cpu.parameters()))


@pytest.mark.parametrize('cls_name', ['omic', 'protein'])
def test_policy_gradient_step_matches_oracle(gpu, tmp_path, cls_name):
    learner, odec, enc_cpu, df = _setup(tmp_path, cls_name)
    B = HP['batch_size']
    target = 'cl1' if cls_name == 'omic' else 'p1'
    # the molecules this step will sample (Philox: same seed + offset + weights -> same tokens)
    learner.generator.seed(7)
    with torch.no_grad():
        latent = learner.encode_cell_line(target, B) if cls_name == 'omic' else learner.encode_protein(target, B)
    smiles, nums, valid_idx = learner.get_smiles_from_latent(latent)
    assert 0 < len(valid_idx) < B                      # some molecules were dropped as invalid
    assert all(n.tolist()[-2:] == [P.STOP, P.STOP] for n in nums)   # reinforce.py:177-184
    rewards = reward_fn(smiles, target)
    learner.generator.seed(7)
    rew, loss = learner.policy_gradient(target, 1, B)
    if cls_name == 'omic':
        x = torch.unsqueeze(torch.Tensor(df[df['cell_line'] == target].iloc[0].gene_expression.values), 0)
    else:
        x = torch.unsqueeze(torch.Tensor(df.loc[target].values), 0)
    o_rew, o_loss = _oracle_step(odec, enc_cpu, x, B, nums, valid_idx, rewards)
    assert abs(float(rew) - float(o_rew)) < 1e-6
    assert abs(loss - o_loss) <= 1e-4 * abs(o_loss)
    pdec = learner.generator.decoder
    for (n, po), (_, pp) in zip(odec.named_parameters(), pdec.named_parameters()):
        # one Adam step of size lr: compare the parameters after the update
        assert float((pp.detach().cpu() - po.detach()).abs().max()) < 0.02 * HP['learning_rate'], n
    # the quirk of reinforce_omics.py:336-341: encoder gradients exist (autograd reaches the encoder
    # through latent_to_hidden), take part in the clip norm, and the encoder is never stepped
    for (n, po), (_, pp) in zip(enc_cpu.named_parameters(), learner.encoder.named_parameters()):
        assert torch.equal(pp.detach().cpu(), po.detach()), n
        if po.grad is None:      # logvar branch: reparameterize returns mu only (reinforce.py:110)
            assert pp.grad is None, n
        else:
            assert P.rel_err(pp.grad.cpu(), po.grad) < 1e-4, n


def test_driver_loop_checkpoints_and_csv(gpu, tmp_path):
    from paccmann_generator_amd import driver
    learner, odec, enc_cpu, df = _setup(tmp_path)
    before = copy.deepcopy(learner.generator.decoder.state_dict())
    lines = []
    rewards, losses, rows = driver.train(learner, ['cl0', 'cl1'], ['cl2'], HP, rng=np.random.RandomState(1),
                                         log=lines.append)
    n_steps = HP['epochs'] * (HP['steps'] - 1)          # range(1, steps), train_paccmann_rl.py:187
    assert len(rewards) == len(losses) == len(lines) == n_steps
    assert all(np.isfinite(losses)) and all(0.0 <= r <= 1.0 for r in rewards)
    w = os.path.join(learner.model_path, 'weights')
    for e in (1, 2):
        assert os.path.exists(os.path.join(w, f'gen_{e}.pt')) and os.path.exists(os.path.join(w, f'enc_{e}.pt'))
    assert os.path.exists(os.path.join(w, 'generator_epoch_0.pt'))      # reinforce.py:66 untrained snapshot
    res = os.path.join(learner.model_path, 'results')
    ev = pd.read_csv(os.path.join(res, 'loss_reward_evolution.csv'))
    assert len(ev) == n_steps and list(ev.columns[1:]) == ['loss', 'rewards']
    gen_csv = pd.read_csv(os.path.join(res, 'generated.csv'))
    assert set(gen_csv['mode']) == {'train', 'test'} and len(gen_csv) == len(rows)
    assert list(gen_csv.columns[1:]) == ['cell_line', 'SMILES', 'IC50', 'mode']     # train_paccmann_rl.py:260-269
    after = learner.generator.decoder.state_dict()
    assert any(not torch.equal(before[k].cpu(), after[k].cpu()) for k in before)
    # checkpoints are plain state_dicts that load back into a fresh decoder (format unchanged)
    from paccmann_generator_amd import Generator, StackGRUDecoder
    g2 = Generator(None, StackGRUDecoder(PARAMS))
    sd = torch.load(os.path.join(w, 'gen_2.pt'), weights_only=True)
    assert any(k.startswith('encoder.') for k in sd) and any(k.startswith('decoder.') for k in sd)
    g2.load(os.path.join(w, 'gen_2.pt'))      # built without the encoder: its entries are skipped
    assert torch.equal(g2.decoder.output_layer.weight, after['output_layer.weight'].cpu())


def test_sets_variant_uses_mean_latent_and_no_clipping(gpu, tmp_path):
    from paccmann_generator_amd import Generator, ReinforceMultiModalSets
    odec, pdec = P.make_pair(PARAMS, seed=0)

    class SetEncoder(nn.Module):
        latent_size = 16

        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(6, 16), nn.Linear(6, 16)

        def forward(self, s):      # s [1, B, 6] -> two [1, B, 16] tensors (cell state, read vector)
            return self.a(s), torch.tanh(self.b(s))

    torch.manual_seed(3)
    enc = SetEncoder().cuda()
    gen = Generator(enc, pdec)
    gen._associate_language(FakeLanguage())
    torch.manual_seed(4)
    base = torch.randn(1, 1, 6)

    def set_fn(cell_line, protein, batch_size):
        return base.repeat(1, batch_size, 1)

    hp = dict(HP, model_folder=str(tmp_path / 'bm'), clip_grad=0.1)
    learner = ReinforceMultiModalSets(gen, enc, hp, 's', logging.getLogger('t'), set_fn=set_fn, is_valid=is_valid,
                                      reward_fn=lambda smiles, protein, cell: reward_fn(smiles, protein))
    assert learner.grad_clipping is None                # reinforce_sets.py:499-501: no clip
    learner.generator.seed(11)
    with torch.no_grad():
        latent = learner.encode_set('c', 'p', 32)
    a, b = enc(base.repeat(1, 32, 1).cuda())
    assert torch.allclose(latent, (a + b) / 2)          # reinforce_sets.py:464
    smiles, nums, valid_idx = learner.get_smiles_from_latent(latent)
    rewards = reward_fn(smiles, 'p')
    learner.generator.seed(11)
    rew, loss = learner.policy_gradient('c', 'p', 1, 32)
    enc_cpu = copy.deepcopy(enc).cpu()
    for p in enc_cpu.parameters():
        p.grad = None
    opt = RL.make_optimizer(odec, dict(learning_rate=HP['learning_rate']))
    lat_cpu = torch.mean(torch.stack(list(enc_cpu(base.repeat(1, 32, 1)))), dim=0)
    _, o_loss = RL.policy_gradient_step(odec, opt, lat_cpu, [n.cpu() for n in nums], valid_idx, rewards, lrps=2,
                                        clip_grad=None)
    assert abs(loss - o_loss) <= 1e-4 * abs(o_loss)
    for (n, po), (_, pp) in zip(odec.named_parameters(), learner.generator.decoder.named_parameters()):
        assert float((pp.detach().cpu() - po.detach()).abs().max()) < 0.02 * HP['learning_rate'], n


def test_affinity_driver_preset_and_protein_embedding_columns(gpu, tmp_path):
    """examples/affinity/train_conditional_generator.py:232-320: checkpoints every 10th epoch, one
    held-out protein evaluated as 'eval' and never trained on, generated.csv with the columns
    protein / SMILES / Binding probability / mode / Tox21.  The protein frame has the reference's
    layout: a 'Sequence' column next to the embedding columns '0'..'767' (reinforce_proteins.py:124-128)."""
    from paccmann_generator_amd import Generator, ReinforceProtein, driver
    _, pdec = P.make_pair(PARAMS, seed=0)
    torch.manual_seed(5)
    enc = FakeEncoder(768, 16).cuda()
    gen = Generator(enc, pdec)
    gen._associate_language(FakeLanguage())
    rng = np.random.RandomState(0)
    df = pd.DataFrame(rng.randn(3, 768).astype(np.float32), index=['p0', 'p1', 'p2'],
                      columns=[str(i) for i in range(768)])
    df.insert(0, 'Sequence', ['MKV', 'MAA', 'MGG'])
    hp = dict(HP, model_folder=str(tmp_path / 'bm'), epochs=10, steps=2, batch_size=24, eval_batch_size=16)
    seen = []

    def reward(smiles, target):
        seen.append(target)
        return reward_fn(smiles, target)

    learner = ReinforceProtein(gen, enc, None, df, hp, 'aff', logging.getLogger('t'), is_valid=is_valid,
                               reward_fn=reward, score_fn=lambda smiles, t: reward_fn(smiles, t))
    with torch.no_grad():
        z = learner.encode_protein('p1', 4)
        mu, _ = enc(torch.tensor(df.loc['p1'][[str(i) for i in range(768)]].values.astype(np.float32)).cuda()[None])
    assert torch.allclose(z[0, 0], mu[0]) and z.shape == (1, 4, 16)
    rewards, losses, rows = driver.train(learner, list(df.index), ['p2'], hp, rng=np.random.RandomState(1),
                                         log=lambda s: None, preset='affinity', tox_fn=lambda s: 1.0)
    assert len(rewards) == 10 and 'p2' not in seen          # range(1, steps) per epoch; held-out never trained on
    w = os.path.join(learner.model_path, 'weights')
    assert os.path.exists(os.path.join(w, 'gen_10.pt')) and not os.path.exists(os.path.join(w, 'gen_9.pt'))
    gen_csv = pd.read_csv(os.path.join(learner.model_path, 'results', 'generated.csv'))
    assert list(gen_csv.columns[1:]) == ['protein', 'SMILES', 'Binding probability', 'mode', 'Tox21']
    assert set(gen_csv['mode']) <= {'eval'} and set(gen_csv['protein']) <= {'p2'}
    assert (gen_csv['Binding probability'] > 0.5).all()
    # ratios.csv (train_conditional_generator.py:283-289,317-320): one row per epoch, percentages of ALL
    # evaluated molecules, one decimal
    ratios = pd.read_csv(os.path.join(learner.model_path, 'results', 'ratios.csv'))
    assert list(ratios.columns[1:]) == ['efficacy_ratio', 'tox_ratio'] and len(ratios) == 10
    assert ((ratios['tox_ratio'] == 100.0) | ratios['tox_ratio'].isna()).all()
    eff = ratios['efficacy_ratio'].dropna()
    assert ((eff >= 0) & (eff <= 100)).all() and np.allclose(eff * 10, np.round(eff * 10))
