"""pmg_decode_noalloc (the entry point that allocates nothing: static device scratch inside the
library) against pmg_decode, through the C ABI: bit-exact tokens and lengths on two calls in a row,
the second with a smaller input; oversize shapes are rejected; neither torch's allocator nor the
driver (hipMemGetInfo, which also sees a hipMalloc made inside the library) sees an allocation during
the calls; two calls captured in a HIP graph replay bit-exactly.  Collected last (see the file name): a failure here must not hide other tests."""
import ctypes as C

import pytest
import torch

from tests import parity as P

pytestmark = pytest.mark.gpu


def _decode_both(pdec, latent, gen_len, temperature, seed, base, first_on_device=False):
    from paccmann_generator_amd import _ffi, ops
    w = pdec.weights()
    B = latent.shape[0]
    ref_tok, ref_len = ops.decode(w, latent, P.START, P.STOP, gen_len, temperature, seed=seed, offset=5,
                                  seq_index_base=base, early_exit=True)
    shape = w.shape(B, gen_len)
    tokens = torch.full((B, gen_len), -7, dtype=torch.int64, device=latent.device)
    lens = torch.full((B,), -7, dtype=torch.int32, device=latent.device)
    wstruct, keep = w.struct(True)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before = torch.cuda.mem_get_info()[0]      # hipMemGetInfo: sees hipMalloc/hipFree by anybody
    rc = _ffi.lib.pmg_decode_noalloc(C.byref(shape), C.byref(wstruct), _ffi._ptr(latent), P.START, P.STOP, gen_len,
                                     float(temperature), seed, 5, base, 0, _ffi._ptr(tokens), _ffi._ptr(lens),
                                     _ffi.current_stream())
    _ffi.check(rc, 'pmg_decode_noalloc')
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if not first_on_device:     # the first call on a device may load the code object / look the symbol up
        assert torch.cuda.mem_get_info()[0] == free_before
    del keep
    return (ref_tok, ref_len), (tokens, lens)


def test_noalloc_decode_equals_decode_twice_second_smaller(gpu):
    from paccmann_generator_amd import _ffi
    assert _ffi.lib.pmg_noalloc_scratch_bytes() == 32 << 20
    _, pdec = P.make_pair(P.CFG1, seed=0)
    torch.manual_seed(3)
    latent = torch.randn(200, P.CFG1['latent_dim'], device='cuda')
    (rt, rl), (t, l) = _decode_both(pdec, latent, 60, 0.8, seed=11, base=0, first_on_device=True)
    assert torch.equal(rt, t) and torch.equal(rl, l)
    # second call: fewer sequences, fewer steps, greedy; the scratch still holds the first call's state
    small = latent[:37].contiguous()
    (rt2, rl2), (t2, l2) = _decode_both(pdec, small, 23, 0.0, seed=11, base=0)
    assert torch.equal(rt2, t2) and torch.equal(rl2, l2)
    # and a third, back at the first size, must reproduce the first result
    (_, _), (t3, l3) = _decode_both(pdec, latent, 60, 0.8, seed=11, base=0)
    assert torch.equal(t3, t) and torch.equal(l3, l)


def test_noalloc_rejects_shapes_beyond_the_static_scratch(gpu):
    from paccmann_generator_amd import _ffi
    _, pdec = P.make_pair(P.CFG1, seed=0)
    w = pdec.weights()
    B, gen_len = 4096, 16          # 256 row groups x 2 stack slabs of 160 KB: far beyond 32 MiB
    shape = w.shape(B, gen_len)
    assert _ffi.workspace_bytes(shape, _ffi.PMG_MODE_DECODE) > _ffi.lib.pmg_noalloc_scratch_bytes()
    latent = torch.zeros(B, P.CFG1['latent_dim'], device='cuda')
    tokens = torch.zeros(B, gen_len, dtype=torch.int64, device='cuda')
    lens = torch.zeros(B, dtype=torch.int32, device='cuda')
    wstruct, keep = w.struct(True)
    rc = _ffi.lib.pmg_decode_noalloc(C.byref(shape), C.byref(wstruct), _ffi._ptr(latent), P.START, P.STOP, gen_len,
                                     1.0, 0, 0, 0, 0, _ffi._ptr(tokens), _ffi._ptr(lens), _ffi.current_stream())
    assert rc < 0 and b'scratch' in _ffi.lib.pmg_strerror(rc)
    with pytest.raises(_ffi.PmgError):
        _ffi.check(rc, 'pmg_decode_noalloc')


def test_noalloc_decode_replays_from_a_hip_graph(gpu):
    """The use the entry point is for: two pmg_decode_noalloc calls (different Philox offsets) captured
    in one graph after a warm-up call, replayed twice, against pmg_decode."""
    from paccmann_generator_amd import _ffi, ops
    _, pdec = P.make_pair(P.CFG1, seed=0)
    w = pdec.weights()
    B, gen_len = 96, 40
    torch.manual_seed(4)
    latent = torch.randn(B, P.CFG1['latent_dim'], device='cuda')
    shape = w.shape(B, gen_len)
    wstruct, keep = w.struct(True)
    outs = [(torch.full((B, gen_len), -7, dtype=torch.int64, device='cuda'),
             torch.full((B,), -7, dtype=torch.int32, device='cuda')) for _ in range(2)]

    def call(offset, tokens, lens):
        rc = _ffi.lib.pmg_decode_noalloc(C.byref(shape), C.byref(wstruct), _ffi._ptr(latent), P.START, P.STOP, gen_len,
                                         0.9, 17, offset, 0, 0, _ffi._ptr(tokens), _ffi._ptr(lens), _ffi.current_stream())
        _ffi.check(rc, 'pmg_decode_noalloc')

    call(0, *outs[0])                      # warm-up outside the capture: symbol lookup, kernel attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(3, *outs[0])
        call(3 + gen_len, *outs[1])
    refs = [ops.decode(w, latent, P.START, P.STOP, gen_len, 0.9, seed=17, offset=o, early_exit=True) for o in (3, 3 + gen_len)]
    for _ in range(2):
        for t, l in outs:
            t.fill_(-7)
            l.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        for (t, l), (rt, rl) in zip(outs, refs):
            assert torch.equal(t, rt) and torch.equal(l, rl)
    assert not torch.equal(outs[0][0], outs[1][0])
    del keep
