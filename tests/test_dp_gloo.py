"""The N>1 path on CPU: two gloo ranks shard a batch, each forms the REINFORCE gradient of
its slice with scale 1/B_valid_global, exchange them with the product's dp module, and
must end up with the gradient (and, after clip + Adam, the parameters) of the unsharded
batch.  The per-rank gradient here comes from the CPU oracle (the HIP operator is covered
by the -m gpu tests); what this test pins is the sharding, scaling and the exchange."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import reinforce_loop as RL
from oracle.stack_gru import build_decoder
from tests.parity import synthetic_tokens

PARAMS = dict(embedding_size=8, rnn_cell_size=16, vocab_size=11, stack_width=5, stack_depth=6, latent_dim=12,
              n_layers=2)
B, T = 11, 7   # odd on purpose: shards of 6 and 5


def _inputs():
    tokens = synthetic_tokens(T, B, PARAMS['vocab_size'], seed=3)
    torch.manual_seed(1)
    latent = torch.randn(1, B, PARAMS['latent_dim'])
    torch.manual_seed(2)
    reward = torch.rand(B)
    valid = [0, 1, 2, 4, 5, 6, 8, 9, 10]   # rows 3 and 7 were "invalid molecules"
    return tokens, latent, reward, valid


def _step(dec, opt, tokens, latent, reward, rows, scale, group=None, use_dp=False):
    """reinforce_omics.py:289-341 on `rows`, mean replaced by scale*sum."""
    from paccmann_generator_amd import dp
    opt.zero_grad()
    hidden = RL.initial_hidden(dec, latent, rows, 1)
    loss = RL.reinforce_loss(dec, tokens[:, rows], hidden, reward[rows][:, None]) * (scale * len(rows))
    loss.backward()
    if use_dp:
        n = dp.all_reduce_gradients_(dec.parameters(), group=group)
        assert n == sum(p.numel() for p in dec.parameters())
    torch.nn.utils.clip_grad_norm_(list(dec.parameters()), 0.1)
    grads = torch.cat([p.grad.reshape(-1) for p in dec.parameters()]).clone()
    opt.step()
    return grads, torch.cat([p.detach().reshape(-1) for p in dec.parameters()]).clone(), float(loss.item())


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from paccmann_generator_amd import dp
        torch.set_num_threads(1)
        tokens, latent, reward, valid = _inputs()
        lo, hi = dp.shard_range(B, rank, world)
        mine = [r for r in valid if lo <= r < hi]
        n_glob = int(round(dp.global_sum(len(mine), 'cpu')))
        assert n_glob == len(valid)
        dec = build_decoder(PARAMS, seed=0)
        opt = RL.make_optimizer(dec, dict(learning_rate=1e-3))
        g, p, loss = _step(dec, opt, tokens, latent, reward, mine, 1.0 / n_glob, use_dp=True)
        loss_sum = dp.global_sum(loss, 'cpu')
        np.savez(os.path.join(out, f'rank{rank}.npz'), g=g.numpy(), p=p.numpy(), loss=loss_sum, lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from paccmann_generator_amd import dp
    for n in (1, 7, 16, 4096, 32768 + 3):
        for world in (1, 2, 3, 8):
            edges = [dp.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gradient_exchange_equals_unsharded(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    assert (int(r0['lo']), int(r0['hi']), int(r1['lo']), int(r1['hi'])) == (0, 6, 6, 11)
    # replicas stay identical without a broadcast
    assert np.array_equal(r0['g'], r1['g']) and np.array_equal(r0['p'], r1['p'])
    # and equal the single-process step over all valid rows
    tokens, latent, reward, valid = _inputs()
    dec = build_decoder(PARAMS, seed=0)
    opt = RL.make_optimizer(dec, dict(learning_rate=1e-3))
    g, p, loss = _step(dec, opt, tokens, latent, reward, valid, 1.0 / len(valid))
    assert np.abs(r0['g'] - g.numpy()).max() <= 1e-5 * np.abs(g.numpy()).max()
    assert np.abs(r0['p'] - p.numpy()).max() <= 1e-6
    assert abs(float(r0['loss']) - loss) <= 1e-5 * abs(loss)


# ---------------------------------------------------------------------------------------------
# dp.data_parallel_step: the whole data-parallel update (the function Reinforce.learn runs under a
# process group), with an encoder whose gradients carry over between steps, a rank without any
# valid molecule in one step, and a step in which no rank has one.
# ---------------------------------------------------------------------------------------------
N_STEPS = 4
#            step 0                 step 1: rank 1 (rows 6..10) empty   step 2: nobody   step 3
VALID = [[0, 1, 2, 4, 5, 6, 8, 9, 10], [0, 2, 3, 5],                    [],              [1, 4, 7, 8, 10]]


class _Enc(torch.nn.Module):
    """encoder(x) -> (mu, logvar); logvar never reaches the loss (reinforce.py:110)."""

    def __init__(self):
        super().__init__()
        self.mu = torch.nn.Linear(5, PARAMS['latent_dim'])
        self.lv = torch.nn.Linear(5, PARAMS['latent_dim'])

    def forward(self, x):
        return self.mu(x), self.lv(x)


def _models():
    dec = build_decoder(PARAMS, seed=0)
    torch.manual_seed(9)
    enc = _Enc()
    opt = RL.make_optimizer(dec, dict(learning_rate=1e-2))
    return dec, enc, opt


def _latent(enc, n):
    torch.manual_seed(4)
    x = torch.randn(1, 5)
    mu, _ = enc(x)
    return torch.unsqueeze(mu.repeat(n, 1), 0)          # reinforce_omics.py:89-94


def _reference_run():
    """Single process, the reference's own order of operations (reinforce_omics.py:289-341): the
    optimizer zeroes decoder gradients only, encoder gradients pile up, clip sees both."""
    tokens, _, reward, _ = _inputs()
    dec, enc, opt = _models()
    out = []
    for s in range(N_STEPS):
        rows = VALID[s]
        if not rows:
            out.append((0.0, 0.0))
            continue
        opt.zero_grad()
        latent = _latent(enc, B)
        hidden = RL.initial_hidden(dec, latent, rows, 1)
        loss = RL.reinforce_loss(dec, tokens[:, rows], hidden, reward[rows][:, None])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(dec.parameters()) + list(enc.parameters()), 0.1)
        opt.step()
        out.append((float(reward[rows].mean()), float(loss.item())))
    return dec, enc, out


def _dp_worker(rank, world, port, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from paccmann_generator_amd import dp
        torch.set_num_threads(1)
        tokens, _, reward, _ = _inputs()
        dec, enc, opt = _models()
        lo, hi = dp.shard_range(B, rank, world)
        res = []
        for s in range(N_STEPS):
            mine = [r for r in VALID[s] if lo <= r < hi]
            local = [r - lo for r in mine]

            def backward():
                latent = _latent(enc, hi - lo)            # this rank encodes only its shard
                hidden = RL.initial_hidden(dec, latent, local, 1)
                loss = RL.reinforce_loss(dec, tokens[:, mine], hidden, reward[mine][:, None]) * len(mine)
                loss.backward()
                return loss.item()

            r, l, n = dp.data_parallel_step(list(dec.parameters()), list(enc.parameters()), opt, backward,
                                            len(mine), float(reward[mine].sum()) if mine else 0.0, 0.1)
            assert int(round(float(n))) == len(VALID[s])
            res.append((float(r), float(l)))
        np.savez(os.path.join(outdir, f'dp{rank}.npz'),
                 p=torch.cat([p.detach().reshape(-1) for p in dec.parameters()]).numpy(),
                 ge=torch.cat([p.grad.reshape(-1) for p in enc.mu.parameters()]).numpy(),
                 lv_none=all(p.grad is None for p in enc.lv.parameters()), res=np.array(res))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_with_encoder_empty_rank_and_empty_step(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'dp0.npz')
    r1 = np.load(tmp_path / 'dp1.npz')
    assert np.array_equal(r0['p'], r1['p']) and np.array_equal(r0['ge'], r1['ge'])   # replicas identical
    dec, enc, out = _reference_run()
    p = torch.cat([q.detach().reshape(-1) for q in dec.parameters()]).numpy()
    ge = torch.cat([q.grad.reshape(-1) for q in enc.mu.parameters()]).numpy()
    assert np.abs(r0['p'] - p).max() <= 2e-5, np.abs(r0['p'] - p).max()
    # encoder gradients carried over 3 updates: equal to the single-process sum, NOT multiplied by the world size
    assert np.abs(r0['ge'] - ge).max() <= 1e-4 * np.abs(ge).max()
    assert bool(r0['lv_none']) and all(q.grad is None for q in enc.lv.parameters())
    for (r, l), (rr, ll) in zip(r0['res'], out):
        assert abs(r - rr) <= 1e-5 * max(1.0, abs(rr)) and abs(l - ll) <= 1e-4 * max(1.0, abs(ll))
