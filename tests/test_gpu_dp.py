"""Data-parallel policy_gradient through the product's own classes on the GPU: two ranks (gloo
rendezvous, both on the one GPU of the test box; collectives staged through the host -- RCCL itself
needs one GPU per rank and is exercised by `bench.py --gpus N`) each run
``ReinforceOmic.policy_gradient(target, epoch, GLOBAL batch)`` with a process group and must

* decode exactly their contiguous shard of the molecules the unsharded run samples (Philox counters
  keyed by the global sequence index),
* survive a step in which one rank has no valid molecule and a step in which nobody has one,
* pad to the GLOBAL T_max (step 3: rank 0 keeps short molecules only, so its own longest valid
  molecule is shorter than rank 1's; the unsharded reference pads every row to the batch's longest),
* give the same result with the host scoring pipelined in micro-batches (micro_batches=2 per rank),
* end with the decoder parameters, encoder gradients (carried over, not multiplied by the world
  size) and (mean reward, loss) of the single-process run over the whole batch.
"""
import copy
import json
import logging
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch
import torch.multiprocessing as mp

from tests import parity as P
from tests.test_gpu_harness import ALPHABET, FakeEncoder, reward_fn

pytestmark = pytest.mark.gpu
PARAMS = dict(P.TINY, latent_dim=32)
HP = dict(generate_len=12, temperature=0.9, clip_grad=0.1, learning_rate=1e-3, save_untrained=False)
BATCH, STEPS, WORLD = 37, 4, 2        # shards of 19 and 18


class TableLanguage:
    """Exposes index_to_token, so get_smiles_from_latent takes the device detokeniser path."""
    start_index, stop_index = P.START, P.STOP
    index_to_token = {i: ALPHABET[i % len(ALPHABET)] for i in range(PARAMS['vocab_size'])}

    def token_indexes_to_smiles(self, idx):
        return ''.join(self.index_to_token[i] for i in idx)

    def selfies_to_smiles(self, s):
        return s


def _validity(step, first_row, smiles):
    """Deterministic in (step, global row, text).  Step 1: every row of rank 1's shard is invalid.
    Step 2: nothing is valid anywhere.  Step 3: rank 0's shard keeps molecules shorter than 6 only."""
    out = []
    for i, s in enumerate(smiles):
        row = first_row + i
        ok = len(s) > 0 and (s.count('C') + s.count('N') + row) % 3 != 0
        if step == 1 and row >= 19:
            ok = False
        if step == 2:
            ok = False
        if step == 3 and row < 19:
            ok = 0 < len(s) < 6
        out.append(ok)
    return out


def _build(process_group, micro_batches=1, shard_lo=0):
    from paccmann_generator_amd import Generator, ReinforceOmic
    _, pdec = P.make_pair(PARAMS, seed=0)
    torch.manual_seed(5)
    enc = FakeEncoder(9, 16).cuda()
    gen = Generator(enc, pdec)
    gen._associate_language(TableLanguage())
    gen.seed(21)
    rng = np.random.RandomState(0)
    df = pd.DataFrame({'cell_line': ['cl0', 'cl1'],
                       'gene_expression': [pd.Series(rng.randn(9).astype(np.float32)) for _ in range(2)]})
    state = dict(step=0, seen=[], cursor=0)

    def is_valid_batch(smiles):
        # called once per step (serial) or once per micro-batch, in row order: the cursor is the first row
        base = shard_lo + state['cursor']
        state['cursor'] += len(smiles)
        state['seen'][-1][1].extend(smiles)
        return _validity(state['step'], base, smiles)

    learner = ReinforceOmic(gen, enc, None, df, dict(HP), 'm', logging.getLogger('dp'), reward_fn=reward_fn,
                            is_valid_batch=is_valid_batch, process_group=process_group, micro_batches=micro_batches)
    return learner, state


def _run(learner, state):
    res = []
    for s in range(STEPS):
        state['step'], state['cursor'] = s, 0
        state['seen'].append([None, []])
        r, l = learner.policy_gradient('cl1', 1, BATCH)
        res.append((float(r), float(l)))
    dec = torch.cat([p.detach().reshape(-1) for p in learner.generator.decoder.parameters()]).cpu().numpy()
    genc = torch.cat([p.grad.reshape(-1) for p in learner.encoder.mu.parameters()]).cpu().numpy()
    return res, dec, genc


def _worker(rank, world, port, outdir, micro_batches):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from paccmann_generator_amd import dp
        learner, state = _build(dist.group.WORLD, micro_batches, dp.shard_range(BATCH, rank, world)[0])
        res, dec, genc = _run(learner, state)
        np.savez(os.path.join(outdir, f'r{rank}.npz'), res=np.array(res), dec=dec, genc=genc)
        with open(os.path.join(outdir, f'r{rank}.json'), 'w') as f:
            json.dump(state['seen'], f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('micro_batches', [1, 2])
def test_two_rank_policy_gradient_equals_unsharded(gpu, tmp_path, micro_batches):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), micro_batches), nprocs=WORLD, join=True)
    learner, state = _build(None)
    res, dec, genc = _run(learner, state)
    r0, r1 = np.load(tmp_path / 'r0.npz'), np.load(tmp_path / 'r1.npz')
    s0, s1 = json.load(open(tmp_path / 'r0.json')), json.load(open(tmp_path / 'r1.json'))
    # (i) the shards are the rows of the unsharded decode, step by step
    for st in range(STEPS):
        whole = state['seen'][st][1]
        assert len(whole) == BATCH and (len(s0[st][1]), len(s1[st][1])) == (19, 18)
        assert s0[st][1] + s1[st][1] == whole, st
    # (ii) replicas identical, no broadcast ever
    assert np.array_equal(r0['dec'], r1['dec']) and np.array_equal(r0['genc'], r1['genc'])
    # (iii) same update as the single process
    lr = HP['learning_rate']
    assert np.abs(r0['dec'] - dec).max() < 0.02 * lr, np.abs(r0['dec'] - dec).max()
    assert np.abs(r0['genc'] - genc).max() <= 1e-4 * np.abs(genc).max()
    for (a, b), (c, d) in zip(r0['res'], res):
        assert abs(a - c) <= 1e-5 * max(1.0, abs(c)) and abs(b - d) <= 1e-4 * max(1.0, abs(d))
    assert res[2] == (0.0, 0.0) and tuple(r0['res'][2]) == (0.0, 0.0)      # the step nobody could learn from
    assert res[1][1] != 0.0 and res[3][1] != 0.0
