#!/usr/bin/env python
"""Times the detokenisation stage of reinforce.py:172-176 on one decoded cfg2 batch (B=4096, T=128):
the reference's per-row host loop (dictionary look-ups + join over row.tolist()) against
pmg_detokenize (+ the two device->host copies and the B string slices).  Prints one JSON line.

usage: python profiles/bench_detok.py [--batch 4096 --len 128]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paccmann_generator_amd import ops  # noqa: E402


def main():
    B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4096
    N = int(sys.argv[sys.argv.index('--len') + 1]) if '--len' in sys.argv else 128
    V = 60
    rng = np.random.RandomState(0)
    itt = {i: ('' if i < 4 else '[' + 'CNOSFPB'[i % 7] * (1 + i % 9) + ']') for i in range(V)}   # SELFIES-like, 3..11 B
    tokens = torch.from_numpy(rng.randint(4, V, size=(B, N)).astype(np.int64)).cuda()
    lens = torch.from_numpy(rng.randint(20, N - 1, size=B).astype(np.int32)).cuda()
    table = ops.TokenTable(itt, 'cuda')

    def host_loop():
        tc, lc = tokens.cpu(), lens.cpu().tolist()
        return [''.join(itt.get(i, '') for i in row[:n].tolist()) for row, n in zip(tc, lc)]

    def device():
        return ops.detokenize(table, tokens, lens)

    assert host_loop() == device()
    res = {}
    for name, fn in (('host_loop_ms', host_loop), ('device_ms', device)):
        fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(10):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) * 1e3)
        res[name] = round(float(np.median(t)), 3)
    # the three kernels alone, HIP events on the launch stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.detokenize_device(table, tokens, lens)
    e0.record()
    for _ in range(20):
        off, blob = ops.detokenize_device(table, tokens, lens)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 20
    total = int(off[-1].item())
    alg = int(lens.sum().item()) * 8 + B * 4 + (B + 1) * 8 * 3 + total    # tokens read + lens + offsets + text written
    res.update(kernels_ms=round(k_ms, 4), text_bytes=total, algorithmic_bytes=alg,
               achieved_GBps=round(alg / k_ms / 1e6, 1), batch=B, gen_len=N,
               speedup=round(res['host_loop_ms'] / res['device_ms'], 1))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
