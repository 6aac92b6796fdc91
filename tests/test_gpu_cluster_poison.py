"""The bounded cluster barriers end to end: a clustered launch whose members cannot all be resident must
neither hang nor return plausible numbers (include/pmg.h, pmg_decode).  The launch goes to a stream masked down
to eight CUs (hipExtStreamCreateWithCUMask), where a cluster of sixteen workgroups of more than half a CU's LDS
each can never assemble: the resident members give up after about a second, raise the status word, every other
workgroup leaves on its first poll, and the outputs are poisoned."""
import ctypes as C
import time

import pytest
import torch

from tests import parity as P

pytestmark = pytest.mark.gpu


def _masked_stream(n_cus: int):
    hip = C.CDLL('libamdhip64.so')
    mask = (C.c_uint32 * 8)(*([(1 << n_cus) - 1] + [0] * 7))
    handle = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(8), mask)
    if rc != 0 or not handle.value:
        pytest.skip(f'hipExtStreamCreateWithCUMask unavailable (rc={rc})')
    return hip, handle


@pytest.mark.timeout(60)
def test_cluster_that_cannot_assemble_gives_up_and_poisons(gpu):
    from paccmann_generator_amd import _ffi, ops
    from paccmann_generator_amd.reinforce import StopRows, _loss_value
    cfg = dict(P.TINY, rnn_cell_size=256)
    _, pdec = P.make_pair(cfg, seed=0)
    w = pdec.weights()
    B, T = 20, 6
    torch.manual_seed(0)
    latent = torch.randn(1, B, cfg['latent_dim'], device='cuda')
    ops.USE_CLUSTER = True
    tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.0, seed=1)       # sanity: the same call on all CUs
    assert int(lens.min()) >= 0
    hip, handle = _masked_stream(8)
    try:
        stream = torch.cuda.ExternalStream(handle.value)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.0, seed=1)
            toks = P.synthetic_tokens(T, B, cfg['vocab_size'], seed=1).cuda()
            h = torch.tanh(torch.randn(1, B, 256, device='cuda')).requires_grad_(True)
            loss, logp = ops.reinforce_loss(w, toks, h, torch.rand(B, device='cuda'), 1.0 / B)
        stream.synchronize()
        assert time.perf_counter() - t0 < 30.0
        assert bool((lens == -1).all())
        with pytest.raises(_ffi.PmgError, match='cluster barrier'):
            StopRows(tokens, lens, 3)[0]
        assert bool(torch.isnan(logp).all())
        with pytest.raises(_ffi.PmgError, match='NaN'):
            _loss_value(loss)
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(handle)
    # and the device is fine afterwards
    tokens, lens = ops.decode(w, latent, 2, 3, T, temperature=0.0, seed=1)
    assert int(lens.min()) >= 0


@pytest.mark.timeout(120)
def test_clustered_launches_from_two_streams_take_turns(gpu):
    """Two host threads, two streams, clustered launches of 256 workgroups each: left alone they keep each other's
    cluster members off the CUs (100x slower, profiles/tools/concurrent_clusters.py before the guard).  The operators
    make them take turns on the device (ops._ClusterTurn): same results as the serial runs, and about the serial time."""
    import threading
    from paccmann_generator_amd import ops
    ops.USE_CLUSTER = True
    cfg = dict(P.CFG1, rnn_cell_size=512)
    _, pdec = P.make_pair(cfg, seed=0)
    w = pdec.weights()
    B, T, reps = 768, 30, 6
    lat = [torch.randn(1, B, cfg['latent_dim'], device='cuda') for _ in range(2)]
    ref = [ops.decode(w, l, 2, 3, T, temperature=0.9, seed=5 + i, early_exit=False) for i, l in enumerate(lat)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(2):
            ops.decode(w, lat[i], 2, 3, T, temperature=0.9, seed=5 + i, early_exit=False)
    torch.cuda.synchronize()
    serial = time.perf_counter() - t0
    streams = [torch.cuda.Stream() for _ in range(2)]
    out, errors = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(reps):
                    out[i] = ops.decode(w, lat[i], 2, 3, T, temperature=0.9, seed=5 + i, early_exit=False)
            streams[i].synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize()
    both = time.perf_counter() - t0
    assert not errors, errors
    for i in range(2):
        assert torch.equal(out[i][0], ref[i][0]) and torch.equal(out[i][1], ref[i][1])
    assert both < 5.0 * serial + 0.05, (both, serial)
