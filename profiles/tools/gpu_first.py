"""Bring-up script (not collected by pytest): run each parity check and print what it finds."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import parity as P

def run(name, fn):
    print('==', name, flush=True)
    try:
        print('   ', fn(), flush=True)
    except Exception:
        traceback.print_exc()
        sys.stdout.flush()

tiny2 = dict(P.TINY, n_layers=2)
run('step tiny', lambda: P.check_step(P.TINY, 20))
run('step tiny L2', lambda: P.check_step(tiny2, 20))
run('greedy tiny', lambda: P.check_greedy(P.TINY, 20, 12))
run('train tiny', lambda: P.check_train(P.TINY, 20, 9, verbose=True))
run('train tiny L2', lambda: P.check_train(tiny2, 20, 9, verbose=True))
run('sampling tiny', lambda: {k: v for k, v in P.check_sampling(P.TINY, 20, 12).items() if k in ('exact', 'boundary')})
run('greedy cfg1', lambda: P.check_greedy(P.CFG1, 64, 100))
run('train cfg1', lambda: P.check_train(P.CFG1, 64, 30, verbose=True))
