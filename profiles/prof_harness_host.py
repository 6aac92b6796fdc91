#!/usr/bin/env python
"""cProfile of ReinforceOmic.policy_gradient at cfg2 (host side only): where the Python glue goes."""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_harness import Enc, Lang, V  # noqa: E402
from paccmann_generator_amd import Generator, ReinforceOmic, StackGRUDecoder  # noqa: E402

B = 4096
torch.manual_seed(0)
dec = StackGRUDecoder(dict(vocab_size=V)).cuda()
enc = Enc().cuda()
gen = Generator(enc, dec)
gen._associate_language(Lang(True))
df = pd.DataFrame({'cell_line': ['a'], 'gene_expression': [pd.Series(np.random.randn(64).astype(np.float32))]})
hp = dict(generate_len=128, temperature=0.8, clip_grad=0.1, model_folder=tempfile.mkdtemp(), save_untrained=False)
learner = ReinforceOmic(gen, enc, None, df, hp, 'm', None, is_valid=lambda s: len(s) % 11 != 0,
                        reward_fn=lambda smiles, t: np.full(len(smiles), 0.5, dtype=np.float32))
for _ in range(3):
    learner.policy_gradient('a', 1, B)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    learner.policy_gradient('a', 1, B)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
