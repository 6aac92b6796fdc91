"""The operators are registered with torch.library (namespace ``pmg``): schemas, fake (meta)
implementations and the chunk planner, checked without a GPU.  No compute call is made -- the only
real implementation is the HIP one and CPU tensors are refused."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode

from tests import parity as P


def _decoder(params):
    from paccmann_generator_amd import StackGRUDecoder
    torch.manual_seed(0)
    return StackGRUDecoder(params)


def test_operators_are_registered_with_schemas():
    from paccmann_generator_amd import ops  # noqa: F401  (registers)
    for name in ('decode', 'step', 'logprob_fwd', 'reinforce_bwd', 'reinforce_loss', 'detokenize'):
        op = getattr(torch.ops.pmg, name)
        schema = str(op.default._schema)
        assert schema.startswith(f'pmg::{name}(')
    s = str(torch.ops.pmg.reinforce_loss.default._schema)
    assert 'Tensor[] params' in s and s.endswith('-> (Tensor, Tensor, Tensor, Tensor[])')
    assert 'Tensor(a' in str(torch.ops.pmg.detokenize.default._schema)      # mutates its output buffers


def test_fake_implementations_give_the_output_shapes():
    from paccmann_generator_amd import ops
    dec = _decoder(dict(P.TINY, n_layers=2))
    w = dec.weights()
    L, H, V, D, Wd = 2, P.TINY['rnn_cell_size'], P.TINY['vocab_size'], P.TINY['stack_depth'], P.TINY['stack_width']
    B, T = 21, 9
    with FakeTensorMode(allow_non_fake_inputs=True):
        params = [torch.empty_like(p) for p in w.flat()]
        tokens = torch.empty(T, B, dtype=torch.long)
        h0 = torch.empty(L, B, H)
        reward = torch.empty(B)
        loss, logp, gh0, grads = torch.ops.pmg.reinforce_loss(tokens, h0, reward, 0.5, params, D, 0, 0, True)
        assert loss.shape == () and logp.shape == (T - 1, B) and gh0.shape == (L, B, H)
        assert [g.shape for g in grads] == [p.shape for p in params]
        loss, logp, gh0, grads = torch.ops.pmg.reinforce_loss(tokens, h0, reward, 0.5, params, D, 0, 0, False)
        assert grads == [] and gh0.numel() == 0
        lp, ws = torch.ops.pmg.logprob_fwd(tokens, h0, params, D, 0)
        assert lp.shape == (T - 1, B) and ws.dtype == torch.uint8
        l1, g1, gs = torch.ops.pmg.reinforce_bwd(tokens, reward, 1.0, ws, params, D, 0)
        assert l1.shape == (1,) and g1.shape == (L, B, H) and len(gs) == len(params)
        latent = torch.empty(B, P.TINY['latent_dim'])
        tok, lens = torch.ops.pmg.decode(latent, params, torch.empty(H, 32), torch.empty(H), D, 0, 2, 3, 14, 0.8, 1, 0,
                                         0, True)
        assert tok.shape == (B, 14) and tok.dtype == torch.int64 and lens.dtype == torch.int32
        h, s, lg = torch.ops.pmg.step(tokens[0], h0, None, params, D, 0, True)
        assert h.shape == h0.shape and s.shape == (B, D, Wd) and lg.shape == (B, V)


def test_cpu_tensors_are_refused_by_the_real_implementation():
    from paccmann_generator_amd import _ffi, ops
    dec = _decoder(P.TINY)
    w = dec.weights()
    with pytest.raises(_ffi.PmgError):
        ops.reinforce_loss(w, torch.zeros(4, 3, dtype=torch.long), torch.zeros(1, 3, 128), torch.zeros(3), 1.0)
    with pytest.raises(_ffi.PmgError):          # ids are validated on the host before any launch
        ops.check_token_range(torch.tensor([[1, 99]]), 23)


def test_chunk_plan_covers_the_batch_in_row_group_multiples():
    from paccmann_generator_amd import ops
    d4 = _decoder(P.CFG4).weights().dims
    chunks = ops.plan_chunks(d4, 16384, 258)          # BASELINE.json configs[3]
    assert chunks[0][0] == 0 and chunks[-1][1] == 16384
    assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    assert all((hi - lo) % 16 == 0 for lo, hi in chunks) and len(chunks) >= 4
    from paccmann_generator_amd import _ffi
    big = max(hi - lo for lo, hi in chunks)
    assert _ffi.workspace_bytes(ops._shape(d4, big, 258), _ffi.PMG_MODE_TRAIN) <= ops.WORKSPACE_BUDGET_BYTES
    d2 = _decoder(P.CFG2).weights().dims
    assert ops.plan_chunks(d2, 4096, 130) == [(0, 4096)]                     # configs[1]: one call
    assert ops.plan_chunks(d2, 100, 130, chunk=40) == [(0, 48), (48, 96), (96, 100)]
    # a smaller budget (less memory free on the device: a co-tenant, a second rank on the GPU) gives more,
    # smaller chunks, each of which really fits -- checked against the library's own size, not the estimate
    for budget in (6 << 30, 3 << 30, 1 << 30):
        plan = ops.plan_chunks(d2, 4096, 130, budget=budget)
        assert plan[0][0] == 0 and plan[-1][1] == 4096 and len(plan) >= 3
        for lo, hi in plan:
            assert _ffi.workspace_bytes(ops._shape(d2, hi - lo, 130), _ffi.PMG_MODE_TRAIN) <= budget
    assert ops.workspace_budget() <= ops.WORKSPACE_BUDGET_BYTES


def test_workspace_budget_follows_free_device_memory(monkeypatch):
    from paccmann_generator_amd import ops
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'is_initialized', lambda: True)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda device=None: (10 << 30, 288 << 30))
    monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda device=None: 6 << 30)
    monkeypatch.setattr(torch.cuda, 'memory_allocated', lambda device=None: 2 << 30)
    assert ops.workspace_budget() == int(0.9 * (14 << 30))          # free + cached-but-unused
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda device=None: (280 << 30, 288 << 30))
    assert ops.workspace_budget() == ops.WORKSPACE_BUDGET_BYTES     # capped
