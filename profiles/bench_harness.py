#!/usr/bin/env python
"""Wall-clock of one whole ReinforceOmic.policy_gradient call at cfg2 (B=4096, T=128) with the
out-of-scope host pieces stubbed (validity = a string test, reward = a constant-time array): what
the Python glue around the kernels costs on top of bench.py's device-resident step.  Compares a
language that exposes index_to_token (device detokeniser + device padding) with one that does not
(the reference's per-row host loop).

Second part (r03): the same call with a CALIBRATED host stub in place of the
out-of-scope RDKit work -- validity and reward each busy-loop 0.1 ms per molecule (0.2 ms per molecule
in all, the low end of what SELFIES -> SMILES + MolFromSmiles + evaluators cost), fanned out over
HostPool(16) -- serial (micro_batches=1, the reference's order) against pipelined (micro_batches=k).
The pools are forked before the first GPU call.  Prints one JSON line."""
import json
import logging
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paccmann_generator_amd import Generator, HostPool, ReinforceOmic, StackGRUDecoder  # noqa: E402

STUB_US = 100.0       # busy time per molecule and host stage


def _busy(us):
    t_end = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t_end:
        pass


def stub_valid(smiles):
    _busy(STUB_US)
    return len(smiles) % 11 != 0


def stub_reward(smiles):
    _busy(STUB_US)
    return 0.5

V = 60
TOK = {i: ('' if i < 4 else '[' + 'CNOSFPB'[i % 7] * (1 + i % 9) + ']') for i in range(V)}


class Lang:
    start_index, stop_index = 2, 3

    def __init__(self, table):
        if table:
            self.index_to_token = TOK

    def token_indexes_to_smiles(self, idx):
        return ''.join(TOK.get(i, '') for i in idx)

    def selfies_to_smiles(self, s):
        return s


class Enc(nn.Module):
    latent_size = 128

    def __init__(self):
        super().__init__()
        self.mu = nn.Linear(64, 128)

    def forward(self, x):
        return self.mu(x), self.mu(x)


def pipelined(B, out, pools):
    """Serial vs micro-batched policy_gradient with the calibrated host stub behind HostPool."""
    pv, pr = pools
    res = {'stub_us_per_molecule_and_stage': STUB_US, 'pool_workers': pv.workers}
    t0 = time.perf_counter()
    pv(['C' * 10] * B)
    res['host_validity_stage_ms'] = round((time.perf_counter() - t0) * 1e3, 2)     # B molecules through the pool
    for k in (1, 2, 4, 8):
        dec = StackGRUDecoder(dict(vocab_size=V)).cuda()
        enc = Enc().cuda()
        gen = Generator(enc, dec)
        gen._associate_language(Lang(True))
        gen.seed(1)
        df = pd.DataFrame({'cell_line': ['a'], 'gene_expression': [pd.Series(np.random.randn(64).astype(np.float32))]})
        hp = dict(generate_len=128, temperature=0.8, clip_grad=0.1, model_folder=tempfile.mkdtemp(), save_untrained=False)
        learner = ReinforceOmic(gen, enc, None, df, hp, 'm', None, is_valid_batch=pv,
                                reward_fn=lambda smiles, t: np.asarray(pr(smiles), dtype=np.float32), micro_batches=k)
        ts = []
        for i in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            learner.policy_gradient('a', 1, B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[f'micro_batches_{k}_ms'] = round(float(np.median(ts[2:])), 2)
    out['host_stub'] = res


def main():
    B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4096
    pools = (HostPool(stub_valid, workers=16, chunk=64), HostPool(stub_reward, workers=16, chunk=64))   # before any GPU call
    torch.manual_seed(0)
    out = {'batch': B, 'gen_len': 128}
    for name, table in (('device_detok_ms', True), ('host_loop_ms', False)):
        dec = StackGRUDecoder(dict(vocab_size=V)).cuda()
        enc = Enc().cuda()
        gen = Generator(enc, dec)
        gen._associate_language(Lang(table))
        gen.seed(1)
        df = pd.DataFrame({'cell_line': ['a'], 'gene_expression': [pd.Series(np.random.randn(64).astype(np.float32))]})
        hp = dict(generate_len=128, temperature=0.8, clip_grad=0.1, model_folder=tempfile.mkdtemp(), save_untrained=False)
        learner = ReinforceOmic(gen, enc, None, df, hp, 'm', None, is_valid=lambda s: len(s) % 11 != 0,
                                reward_fn=lambda smiles, t: np.full(len(smiles), 0.5, dtype=np.float32))
        ts = []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            learner.policy_gradient('a', 1, B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name] = round(float(np.median(ts[2:])), 2)
        if table:       # where the glue time goes (same calls as policy_gradient, timed one by one)
            def tm(fn):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                return r, round((time.perf_counter() - t0) * 1e3, 2)
            bd = {}
            with torch.no_grad():
                latent, bd['encode'] = tm(lambda: learner.encode_cell_line('a', B))
            (smiles, nums, valid), bd['get_smiles_from_latent'] = tm(lambda: learner.get_smiles_from_latent(latent))
            (tok, lens), bd['  decode'] = tm(lambda: learner.sample_tokens(latent))
            _, bd['  detokenize'] = tm(lambda: __import__('paccmann_generator_amd').ops.detokenize(
                learner._token_table(gen.smiles_language), tok, lens))
            _, bd['  validity_stub'] = tm(lambda: [len(x) % 11 != 0 for x in smiles])
            learner.get_smiles_from_latent(latent)
            latent2 = learner.encode_cell_line('a', B)
            smiles, nums, valid = learner.get_smiles_from_latent(latent2)
            (padded, _), bd['padded_valid'] = tm(lambda: learner._padded_valid(nums, valid))
            # median of three: the first call after the calls above may pay for a fresh workspace allocation
            # (a fresh encoder graph per call: learn() backpropagates through it)
            bd['learn'] = sorted(tm(lambda lat=learner.encode_cell_line('a', B): learner.learn(
                lat, padded, valid, np.full(len(nums), 0.5, np.float32), 2))[1] for _ in range(3))[1]
            out['breakdown_ms'] = bd
    pipelined(B, out, pools)
    for p in pools:
        p.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
