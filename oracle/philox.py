"""Philox4x32-10 counter RNG and the inverse-CDF categorical draw, restated in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference samples with ``torch.multinomial`` inside paccmann_chemistry's
SamplingSearch (call site /root/reference/paccmann_generator/reinforce.py:169), whose
stream cannot be reproduced on a GPU; the build replaces it by a counter-based draw
(Philox4x32-10, Salmon et al., SC'11) from the same distribution
softmax(logits / temperature).  This file is the CPU twin of that draw
(paccmann_generator_amd/csrc/pmg_common.h: pmg_philox_first / pmg_uniform and the
sampling block of pmg_fwd.hip).  Philox itself is pinned by the Random123 known-answer
vectors in tests/test_oracle.py.
"""
from __future__ import annotations

import numpy as np

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 uint32, key: 2 uint32 -> 4 uint32 (python ints)."""
    c0, c1, c2, c3 = [int(x) & MASK for x in counter]
    k0, k1 = [int(x) & MASK for x in key]
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniform(seed: int, seq: int, step: int) -> np.float32:
    """(0,1) uniform with 24 random bits: counter = (seq lo, seq hi, step lo, step hi), key = seed."""
    x = philox4x32_10((seq & MASK, (seq >> 32) & MASK, step & MASK, (step >> 32) & MASK),
                      (seed & MASK, (seed >> 32) & MASK))[0]
    return np.float32((np.float32(x >> 8) + np.float32(0.5)) * np.float32(1.0 / 16777216.0))


def sample_row(logits, temperature, u):
    """Inverse-CDF draw, fp32, same operation order as the kernel (pmg_fwd.hip, "consume logits"):
    entries are taken in slots of 32 (one per lane), the running CDF of a slot is a 32-lane
    Hillis-Steele inclusive scan added to the carry of the previous slots, the draw is the first
    entry whose CDF exceeds u * total.

    Returns (choice, margin, probs): margin = distance of u to the nearest CDF boundary
    (fp64), used by the tests to excuse boundary-rounding differences."""
    f = np.float32
    logits = np.asarray(logits, dtype=np.float32)
    V = logits.shape[0]
    x = (logits / f(temperature)).astype(np.float32)
    m = x.max()
    e = np.exp((x - m).astype(np.float32)).astype(np.float32)
    cdf = np.zeros(V, dtype=np.float32)
    carry = f(0.0)
    for k in range((V + 31) // 32):
        seg = np.zeros(32, dtype=np.float32)
        n = min(32, V - 32 * k)
        seg[:n] = e[32 * k:32 * k + n]
        o = 1
        while o <= 16:
            nxt = seg.copy()
            nxt[o:] = (seg[o:] + seg[:-o]).astype(np.float32)
            seg = nxt
            o *= 2
        c = (carry + seg).astype(np.float32)
        cdf[32 * k:32 * k + n] = c[:n]
        carry = c[31]
    target = f(f(u) * carry)
    hit = np.nonzero(target < cdf)[0]
    choice = int(hit[0]) if hit.size else V - 1
    e64 = e.astype(np.float64)
    cdf64 = np.cumsum(e64) / e64.sum()
    margin = float(np.min(np.abs(cdf64 - float(u))))
    return int(choice), margin, e64 / e64.sum()
