#!/usr/bin/env python
"""Wall-clock of one whole ReinforceOmic.policy_gradient call at cfg2 (B=4096, T=128) with the
out-of-scope host pieces stubbed (validity = a string test, reward = a constant-time array): what
the Python glue around the kernels costs on top of bench.py's device-resident step.  Compares a
language that exposes index_to_token (device detokeniser + device padding) with one that does not
(the reference's per-row host loop).  Prints one JSON line."""
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
from paccmann_generator_amd import Generator, ReinforceOmic, StackGRUDecoder  # noqa: E402

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


def main():
    B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4096
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
            padded, bd['padded_valid'] = tm(lambda: learner._padded_valid(nums, valid))
            _, bd['learn'] = tm(lambda: learner.learn(latent2, padded, valid, np.full(len(nums), 0.5, np.float32), 2))
            out['breakdown_ms'] = bd
    print(json.dumps(out))


if __name__ == '__main__':
    main()
