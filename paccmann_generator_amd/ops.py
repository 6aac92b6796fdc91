"""Operators over the C ABI, registered with ``torch.library`` under the namespace ``pmg``.

These are the pieces of the reference loop that move to the GPU: ``generator.generate``
(reinforce.py:162-170), the teacher-forced loop (reinforce_omics.py:319-332),
``rl_loss.backward()`` (:335), the single decoder step of the protocol (:323-325) and the
per-molecule detokenisation (reinforce.py:172-176).

    torch.ops.pmg.decode          -> pmg_decode
    torch.ops.pmg.step            -> pmg_step
    torch.ops.pmg.logprob_fwd     -> pmg_logprob_fwd          (returns the history workspace)
    torch.ops.pmg.reinforce_bwd   -> pmg_reinforce_bwd        (consumes that workspace)
    torch.ops.pmg.reinforce_loss  -> both, chunked over the batch; the differentiable operator
    torch.ops.pmg.detokenize      -> pmg_detokenize

Every operator has a schema, a fake (meta) implementation and only a CUDA implementation: there is
no CPU path.  Decoder parameters travel as one ``Tensor[]`` in the order of
``DecoderWeights.NAMES`` (per-layer entries expanded): 4 L + 7 tensors; all model dimensions except
the stack depth are read off their shapes.  Workspaces are private to a call (torch's caching
allocator makes that cheap), so calls on different streams or threads never share scratch.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _ffi

# Largest workspace one kernel call of the training pair may take; larger batches run as chunks whose
# gradients accumulate (PMG_FLAG_ACCUMULATE).  At most 96 GiB of the 288 GB of HBM -- configs[3]
# (2 layers, hidden 1024, T=258) needs ~20 GB per 1024 sequences, so B=16384 runs as 4 chunks of 4096 --
# and at most WORKSPACE_FREE_FRACTION of what is free on the device at call time (workspace_budget),
# so that a co-tenant, a second rank on the same GPU or a caller's large resident tensors get smaller
# chunks instead of an out-of-memory error.
WORKSPACE_BUDGET_BYTES = 96 << 30
WORKSPACE_FREE_FRACTION = 0.9


def workspace_budget(device=None) -> int:
    """min(WORKSPACE_BUDGET_BYTES, WORKSPACE_FREE_FRACTION x free bytes), free = what the driver
    reports (hipMemGetInfo) plus the blocks torch's caching allocator holds but has not handed out."""
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return WORKSPACE_BUDGET_BYTES
    free, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    return int(min(WORKSPACE_BUDGET_BYTES, WORKSPACE_FREE_FRACTION * (free + cached)))


# Small batches run on clustered kernels (include/pmg.h, pmg_uses_cluster).  False forces the
# batch-partitioned kernels everywhere (A/B measurements; processes that share the GPU with another
# one that launches clustered kernels).
USE_CLUSTER = True


def _cluster_flag() -> int:
    return 0 if USE_CLUSTER else _ffi.PMG_FLAG_NO_CLUSTER


class _ClusterTurn:
    """Clustered launches need every member of a cluster resident at once (include/pmg.h): two of them running
    side by side on different streams of one device keep each other's members off the CUs and crawl (measured 100x
    slower at 256 workgroups each, profiles/tools/concurrent_clusters.py) or, past a second, give up.  The operators
    therefore take turns per device: a launch that may be clustered waits (on the device, not the host) for the
    previous one issued through this module from ANOTHER stream, and leaves an event for the next.  Launches with
    PMG_FLAG_NO_CLUSTER (ops.TrainSlice, USE_CLUSTER = False) do not take part."""
    _lock = threading.Lock()
    _last = {}          # device index -> (event, stream handle)

    def __init__(self, device, flags: int):
        self.idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        self.active = not (flags & _ffi.PMG_FLAG_NO_CLUSTER)

    def __enter__(self):
        if self.active:
            self._lock.acquire()
            stream = torch.cuda.current_stream(self.idx)
            last = self._last.get(self.idx)
            if last is not None and last[1] != stream.cuda_stream:
                stream.wait_event(last[0])
        return self

    def __exit__(self, *exc):
        if self.active:
            try:
                stream = torch.cuda.current_stream(self.idx)
                last = self._last.get(self.idx)
                # one event per device, recorded again after every launch: a wait captures the record current at
                # the time of the wait_event call, so recording over it later does not disturb earlier waiters
                ev = last[0] if last is not None else torch.cuda.Event()
                ev.record(stream)
                self._last[self.idx] = (ev, stream.cuda_stream)
            finally:
                self._lock.release()
        return False


def release_workspaces() -> None:
    """Kept for callers of the r01 interface: workspaces are per call now; this only returns
    torch's cached blocks to the driver."""
    torch.cuda.empty_cache()


def _alloc_ws(shape, mode: int, device) -> Tensor:
    return torch.empty(_ffi.workspace_bytes(shape, mode), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------------- parameters
class DecoderWeights:
    """Views of a decoder's parameters in the order/naming of ``pmg_weights``."""

    NAMES = ('embedding', 'w_ih', 'w_hh', 'b_ih', 'b_hh', 'w_ctrl', 'b_ctrl', 'w_push', 'b_push',
             'w_out', 'b_out')

    def __init__(self, decoder):
        L = decoder.n_layers
        g = decoder.gru
        self.n_layers = L
        self.t = dict(
            embedding=decoder.embedding.weight,
            w_ih=[getattr(g, f'weight_ih_l{l}') for l in range(L)],
            w_hh=[getattr(g, f'weight_hh_l{l}') for l in range(L)],
            b_ih=[getattr(g, f'bias_ih_l{l}') for l in range(L)],
            b_hh=[getattr(g, f'bias_hh_l{l}') for l in range(L)],
            w_ctrl=decoder.stack_controls_layer.weight, b_ctrl=decoder.stack_controls_layer.bias,
            w_push=decoder.stack_input_layer.weight, b_push=decoder.stack_input_layer.bias,
            w_out=decoder.output_layer.weight, b_out=decoder.output_layer.bias,
            w_l2h=decoder.latent_to_hidden.weight, b_l2h=decoder.latent_to_hidden.bias,
        )
        self.dims = dict(
            H=decoder.rnn_cell_size, E=decoder.embedding_size, W=decoder.stack_width, D=decoder.stack_depth,
            V=decoder.vocab_size, L=L, Z=decoder.latent_dim,
            pad_idx=decoder.embedding.padding_idx,
        )

    @property
    def pad_idx(self) -> int:
        p = self.dims['pad_idx']
        return -1 if p is None else int(p)

    def flat(self) -> List[Tensor]:
        """Parameters in the fixed order of the operators' ``params`` list (latent_to_hidden excluded)."""
        out = []
        for n in self.NAMES:
            v = self.t[n]
            out.extend(v if isinstance(v, list) else [v])
        return out

    @staticmethod
    def unflatten(flat: Sequence[Tensor], L: int) -> dict:
        it = iter(flat)
        d = {}
        for n in DecoderWeights.NAMES:
            if n in ('w_ih', 'w_hh', 'b_ih', 'b_hh'):
                d[n] = [next(it) for _ in range(L)]
            else:
                d[n] = next(it)
        return d

    def shape(self, B: int, T: int):
        d = self.dims
        return _ffi.make_shape(B, T, d['H'], d['E'], d['W'], d['D'], d['V'], d['L'], d['Z'], d['pad_idx'])

    def struct(self, with_l2h: bool = True):
        t = {k: ([x.detach() for x in v] if isinstance(v, list) else v.detach()) for k, v in self.t.items()}
        if not with_l2h:
            t.pop('w_l2h')
            t.pop('b_l2h')
        return _ffi.pack_weight_struct(_ffi.PmgWeights, t, self.n_layers)


def _dims_of(params: Sequence[Tensor], stack_depth: int, pad_idx: int, Z: int = 32) -> dict:
    """Model dimensions from the shapes of the flat parameter list."""
    n = len(params)
    if n < 11 or (n - 7) % 4:
        raise _ffi.PmgError(f'pmg: expected 4*L+7 decoder parameters, got {n}')
    L = (n - 7) // 4
    V, E = params[0].shape
    H = params[1 + L].shape[1]                 # w_hh[0]: [3H, H]
    W = params[1].shape[1] - E                 # w_ih[0]: [3H, E+W]
    return dict(H=int(H), E=int(E), W=int(W), D=int(stack_depth), V=int(V), L=int(L), Z=int(Z), pad_idx=int(pad_idx))


def _shape(d: dict, B: int, T: int):
    return _ffi.make_shape(B, T, d['H'], d['E'], d['W'], d['D'], d['V'], d['L'], d['Z'], d['pad_idx'])


def _wstruct(params: Sequence[Tensor], L: int, w_l2h: Optional[Tensor] = None, b_l2h: Optional[Tensor] = None):
    t = DecoderWeights.unflatten([p.detach() for p in params], L)
    if w_l2h is not None:
        t['w_l2h'], t['b_l2h'] = w_l2h.detach(), b_l2h.detach()
    return _ffi.pack_weight_struct(_ffi.PmgWeights, t, L)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _ffi.PmgError('pmg: tensors must live on the GPU (there is no CPU path)')


# --------------------------------------------------------------------------------- pmg::decode
@torch.library.custom_op('pmg::decode', mutates_args=())
def _decode_op(latent: Tensor, params: Sequence[Tensor], w_l2h: Tensor, b_l2h: Tensor, stack_depth: int,
               pad_idx: int, start_tok: int, stop_tok: int, gen_len: int, temperature: float, seed: int,
               offset: int, seq_index_base: int, early_exit: bool) -> Tuple[Tensor, Tensor]:
    _need_cuda(latent, *params)
    latent = latent.detach().to(torch.float32).contiguous()
    B = latent.shape[0]
    d = _dims_of(params, stack_depth, pad_idx, Z=latent.shape[1])
    shape = _shape(d, B, gen_len)
    ws = _alloc_ws(shape, _ffi.PMG_MODE_DECODE, latent.device)
    tokens = torch.empty(B, gen_len, dtype=torch.int64, device=latent.device)
    lens = torch.empty(B, dtype=torch.int32, device=latent.device)
    wstruct, keep = _wstruct(params, d['L'], w_l2h, b_l2h)
    flags = (0 if early_exit else _ffi.PMG_FLAG_NO_EARLY_EXIT) | _cluster_flag()
    with _ClusterTurn(latent.device, flags):
        rc = _ffi.lib.pmg_decode(C.byref(shape), C.byref(wstruct), _ffi._ptr(latent), int(start_tok), int(stop_tok),
                                 int(gen_len), float(temperature), int(seed), int(offset), int(seq_index_base),
                                 flags, _ffi._ptr(tokens), _ffi._ptr(lens), _ffi._ptr(ws), _ffi.current_stream())
    _ffi.check(rc, 'pmg_decode')
    del keep
    return tokens, lens


@_decode_op.register_fake
def _(latent, params, w_l2h, b_l2h, stack_depth, pad_idx, start_tok, stop_tok, gen_len, temperature, seed, offset,
      seq_index_base, early_exit):
    B = latent.shape[0]
    return latent.new_empty((B, gen_len), dtype=torch.int64), latent.new_empty((B,), dtype=torch.int32)


@torch.no_grad()
def decode(weights: DecoderWeights, latent: Tensor, start_tok: int, stop_tok: int, gen_len: int,
           temperature: float = 1.0, seed: int = 0, offset: int = 0, seq_index_base: int = 0,
           early_exit: bool = True):
    """Returns (tokens [B,gen_len] int64, lengths [B] int32).  latent: [B,Z] or [1,B,Z]."""
    if latent.dim() == 3:
        latent = latent[0]
    return _decode_op(latent, weights.flat(), weights.t['w_l2h'], weights.t['b_l2h'], weights.dims['D'],
                      weights.pad_idx, int(start_tok), int(stop_tok), int(gen_len), float(temperature), int(seed),
                      int(offset), int(seq_index_base), bool(early_exit))


# --------------------------------------------------------------------------------- pmg::step
@torch.library.custom_op('pmg::step', mutates_args=())
def _step_op(tok: Tensor, hidden: Tensor, stack: Optional[Tensor], params: Sequence[Tensor], stack_depth: int,
             pad_idx: int, want_logits: bool) -> Tuple[Tensor, Tensor, Tensor]:
    _need_cuda(tok, hidden, stack, *params)
    tok = tok.reshape(-1).to(torch.int64).contiguous()
    hidden = hidden.detach().to(torch.float32).contiguous()
    B = tok.shape[0]
    d = _dims_of(params, stack_depth, pad_idx)
    shape = _shape(d, B, 1)
    ws = _alloc_ws(shape, _ffi.PMG_MODE_DECODE, hidden.device)
    h_out = torch.empty_like(hidden)
    s_out = torch.empty(B, d['D'], d['W'], dtype=torch.float32, device=hidden.device)
    logits = torch.empty(B, d['V'] if want_logits else 0, dtype=torch.float32, device=hidden.device)
    s_in = None if stack is None else stack.detach().to(torch.float32).contiguous()
    wstruct, keep = _wstruct(params, d['L'])
    rc = _ffi.lib.pmg_step(C.byref(shape), C.byref(wstruct), _ffi._ptr(tok), _ffi._ptr(hidden), _ffi._ptr(s_in),
                           _ffi._ptr(h_out), _ffi._ptr(s_out), _ffi._ptr(logits) if want_logits else None,
                           _ffi._ptr(ws), _ffi.current_stream())
    _ffi.check(rc, 'pmg_step')
    del keep
    return h_out, s_out, logits


@_step_op.register_fake
def _(tok, hidden, stack, params, stack_depth, pad_idx, want_logits):
    B = tok.numel()
    V, E = params[0].shape
    W = params[1].shape[1] - E
    return (torch.empty_like(hidden), hidden.new_empty((B, stack_depth, W)),
            hidden.new_empty((B, V if want_logits else 0)))


@torch.no_grad()
def step(weights: DecoderWeights, tok: Tensor, hidden: Tensor, stack: Optional[Tensor],
         want_logits: bool = True):
    """One decoder step.  tok [B] or [1,B]; hidden [L,B,H]; stack [B,D,W] or None (zeros)."""
    h_out, s_out, logits = _step_op(tok, hidden, stack, weights.flat(), weights.dims['D'], weights.pad_idx,
                                    bool(want_logits))
    return h_out, s_out, (logits if want_logits else None)


# --------------------------------------------------------------------------------- training pair
def _check_train_inputs(tokens: Tensor, h0: Tensor, d: dict):
    _need_cuda(tokens, h0)
    if tokens.dim() != 2 or h0.dim() != 3:
        raise _ffi.PmgError('pmg: tokens must be [T,B] and h0 [L,B,H]')
    T, B = tokens.shape
    if h0.shape != (d['L'], B, d['H']):
        raise _ffi.PmgError(f'pmg: h0 must be [{d["L"]},{B},{d["H"]}], got {tuple(h0.shape)}')
    return T, B


def _fwd_call(shape, wstruct, tokens, h0, logp, ws, flags=0):
    flags = flags | _cluster_flag()
    with _ClusterTurn(h0.device, flags):
        rc = _ffi.lib.pmg_logprob_fwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(h0),
                                       _ffi._ptr(logp), None, flags, _ffi._ptr(ws), _ffi.current_stream())
    _ffi.check(rc, 'pmg_logprob_fwd')


def _bwd_call(shape, wstruct, tokens, reward, scale, flags, gstruct, grad_h0, loss, ws):
    with _ClusterTurn(grad_h0.device, flags):
        rc = _ffi.lib.pmg_reinforce_bwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(reward),
                                        float(scale), flags, C.byref(gstruct), _ffi._ptr(grad_h0), _ffi._ptr(loss),
                                        _ffi._ptr(ws), _ffi.current_stream())
    _ffi.check(rc, 'pmg_reinforce_bwd')


@torch.library.custom_op('pmg::logprob_fwd', mutates_args=())
def _logprob_fwd_op(tokens: Tensor, h0: Tensor, params: Sequence[Tensor], stack_depth: int,
                    pad_idx: int) -> Tuple[Tensor, Tensor]:
    """(logp [T-1,B], ws): ``ws`` holds the trajectory pmg::reinforce_bwd differentiates."""
    d = _dims_of(params, stack_depth, pad_idx)
    T, B = _check_train_inputs(tokens, h0, d)
    tokens = tokens.to(torch.int64).contiguous()
    h0c = h0.detach().to(torch.float32).contiguous()
    shape = _shape(d, B, T)
    # returned to the caller, so zero-filled: the operator stays a pure function of its inputs (the
    # regions the backward will write, and alignment gaps, would otherwise hold stale memory)
    ws = torch.zeros(_ffi.workspace_bytes(shape, _ffi.PMG_MODE_TRAIN), dtype=torch.uint8, device=h0c.device)
    wstruct, keep = _wstruct(params, d['L'])
    logp = torch.empty(T - 1, B, dtype=torch.float32, device=h0c.device)
    _fwd_call(shape, wstruct, tokens, h0c, logp, ws)
    del keep
    return logp, ws


@_logprob_fwd_op.register_fake
def _(tokens, h0, params, stack_depth, pad_idx):
    T, B = tokens.shape
    d = _dims_of(params, stack_depth, pad_idx)
    need = _ffi.workspace_bytes(_shape(d, int(B), int(T)), _ffi.PMG_MODE_TRAIN)   # host-only size query
    return h0.new_empty((T - 1, B)), h0.new_empty((need,), dtype=torch.uint8)


def _grad_struct(grads: Sequence[Tensor], L: int):
    return _ffi.pack_weight_struct(_ffi.PmgGrads, DecoderWeights.unflatten(list(grads), L), L)


@torch.library.custom_op('pmg::reinforce_bwd', mutates_args=('ws',))   # scratch regions of ws are written
def _reinforce_bwd_op(tokens: Tensor, reward: Tensor, scale: float, ws: Tensor, params: Sequence[Tensor],
                      stack_depth: int, pad_idx: int) -> Tuple[Tensor, Tensor, List[Tensor]]:
    """(loss [1], grad_h0 [L,B,H], one gradient per parameter) from the trajectory in ``ws``."""
    _need_cuda(tokens, reward, ws)
    d = _dims_of(params, stack_depth, pad_idx)
    T, B = tokens.shape
    tokens = tokens.to(torch.int64).contiguous()
    reward = reward.detach().to(torch.float32).reshape(-1).contiguous()
    shape = _shape(d, B, T)
    if ws.numel() < _ffi.workspace_bytes(shape, _ffi.PMG_MODE_TRAIN):
        raise _ffi.PmgError('pmg::reinforce_bwd: workspace does not belong to this shape')
    wstruct, keep = _wstruct(params, d['L'])
    grads = [torch.empty_like(p, dtype=torch.float32) for p in params]
    gstruct, keep2 = _grad_struct(grads, d['L'])
    grad_h0 = torch.empty(d['L'], B, d['H'], dtype=torch.float32, device=tokens.device)
    loss = torch.empty(1, dtype=torch.float32, device=tokens.device)
    _bwd_call(shape, wstruct, tokens, reward, scale, _cluster_flag(), gstruct, grad_h0, loss, ws)
    del keep, keep2
    return loss, grad_h0, grads


@_reinforce_bwd_op.register_fake
def _(tokens, reward, scale, ws, params, stack_depth, pad_idx):
    T, B = tokens.shape
    L = (len(params) - 7) // 4
    H = params[1 + L].shape[1]
    return (reward.new_empty((1,), dtype=torch.float32), reward.new_empty((L, B, H), dtype=torch.float32),
            [torch.empty_like(p) for p in params])


def plan_chunks(d: dict, B: int, T: int, budget: Optional[int] = None, chunk: int = 0,
                device=None) -> List[Tuple[int, int]]:
    """[lo, hi) slices of the batch, each a whole number of 16-sequence row groups (but the last),
    whose training workspace fits ``budget`` bytes (default: workspace_budget(device), derived from
    the memory free on the device right now).  ``chunk`` > 0 fixes the slice size instead."""
    if chunk <= 0:
        need = lambda n: _ffi.workspace_bytes(_shape(d, n, T), _ffi.PMG_MODE_TRAIN)      # noqa: E731
        if budget is None:
            # the whole batch within what the driver alone reports free (a lower bound of the budget: no walk
            # through the caching allocator's statistics, 0.1 ms of host time per call): one chunk
            if torch.cuda.is_available() and torch.cuda.is_initialized() and \
                    need(B) <= min(WORKSPACE_BUDGET_BYTES, WORKSPACE_FREE_FRACTION * torch.cuda.mem_get_info(device)[0]):
                return [(0, B)]
            budget = workspace_budget(device)
        budget = int(budget)
        per16 = need(32) - need(16)
        fixed = need(16) - per16
        groups = max(1, int((budget - fixed) // max(1, per16)))
        chunk = min(groups * 16, -(-B // 16) * 16)
        if chunk < B:    # balance: same number of chunks, equal sizes
            n = -(-B // chunk)
            chunk = -(-(-(-B // n)) // 16) * 16
        # the estimate is linear in the row groups; the split-K partial buffers of the weight-gradient
        # GEMM need not be: check the real size and give row groups back until it fits
        while chunk > 16 and need(min(chunk, B)) > budget:
            chunk -= 16
    else:
        chunk = -(-int(chunk) // 16) * 16
    return [(lo, min(B, lo + chunk)) for lo in range(0, B, chunk)]


@torch.library.custom_op('pmg::reinforce_loss', mutates_args=())
def _reinforce_loss_op(tokens: Tensor, h0: Tensor, reward: Tensor, scale: float, params: Sequence[Tensor],
                       stack_depth: int, pad_idx: int, chunk: int, with_grads: bool
                       ) -> Tuple[Tensor, Tensor, Tensor, List[Tensor]]:
    """loss = -scale * sum_{p,b} reward[b] * logp[p,b]  (reinforce_omics.py:322-332 with the mean over
    the valid molecules written as scale * sum), its log-probs, and -- with ``with_grads`` -- the
    gradient of the loss w.r.t. h0 and every parameter, formed in the same pass: per chunk of the
    batch pmg_logprob_fwd then pmg_reinforce_bwd on one shared workspace, chunks after the first with
    PMG_FLAG_ACCUMULATE.  Returns (loss [], logp [T-1,B], grad_h0 [L,B,H], grads)."""
    d = _dims_of(params, stack_depth, pad_idx)
    T, B = _check_train_inputs(tokens, h0, d)
    _need_cuda(reward, *params)
    dev = h0.device
    tokens = tokens.to(torch.int64).contiguous()
    h0c = h0.detach().to(torch.float32).contiguous()
    reward = reward.detach().to(torch.float32).reshape(-1).contiguous()
    if reward.numel() != B:
        raise _ffi.PmgError('pmg: one reward per sequence required')
    L, H = d['L'], d['H']
    chunks = plan_chunks(d, B, T, chunk=chunk, device=dev)
    wstruct, keep = _wstruct(params, L)
    logp = torch.empty(T - 1, B, dtype=torch.float32, device=dev)
    if with_grads:
        grads = [torch.empty_like(p, dtype=torch.float32) for p in params]
        gstruct, keep2 = _grad_struct(grads, L)
        grad_h0 = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
    else:
        grads, grad_h0 = [], torch.empty(0, dtype=torch.float32, device=dev)
    ws = _alloc_ws(_shape(d, chunks[0][1] - chunks[0][0], T), _ffi.PMG_MODE_TRAIN, dev)
    st = _ffi.current_stream()
    for i, (lo, hi) in enumerate(chunks):
        whole = len(chunks) == 1
        shape = _shape(d, hi - lo, T)
        tok_c = tokens if whole else tokens[:, lo:hi].contiguous()
        h0_c = h0c if whole else h0c[:, lo:hi].contiguous()
        lp_c = logp if whole else torch.empty(T - 1, hi - lo, dtype=torch.float32, device=dev)
        # the packed weights of chunk 0 are still in the workspace for the later chunks
        _fwd_call(shape, wstruct, tok_c, h0_c, lp_c, ws, _ffi.PMG_FLAG_REUSE_PACK if i > 0 else 0)
        if not whole:
            logp[:, lo:hi] = lp_c
        if with_grads:
            r_c = reward if whole else reward[lo:hi].contiguous()
            gh_c = grad_h0 if whole else torch.empty(L, hi - lo, H, dtype=torch.float32, device=dev)
            _bwd_call(shape, wstruct, tok_c, r_c, scale, (_ffi.PMG_FLAG_ACCUMULATE if i > 0 else 0) | _cluster_flag(),
                      gstruct, gh_c, loss, ws)
            if not whole:
                grad_h0[:, lo:hi] = gh_c
    if not with_grads:
        loss = -(logp * reward.unsqueeze(0)).sum().reshape(1) * float(scale)
    del keep
    return loss.reshape(()), logp, grad_h0, grads


@_reinforce_loss_op.register_fake
def _(tokens, h0, reward, scale, params, stack_depth, pad_idx, chunk, with_grads):
    T, B = tokens.shape
    f32 = dict(dtype=torch.float32)
    if with_grads:
        return (h0.new_empty((), **f32), h0.new_empty((T - 1, B), **f32), h0.new_empty(h0.shape, **f32),
                [torch.empty_like(p) for p in params])
    return h0.new_empty((), **f32), h0.new_empty((T - 1, B), **f32), h0.new_empty((0,), **f32), []


def _reinforce_setup(ctx, inputs, output):
    tokens, h0, reward, scale, params, stack_depth, pad_idx, chunk, with_grads = inputs
    _, _, grad_h0, grads = output
    ctx.with_grads = bool(with_grads)
    # logp, grad_h0 and grads are results, not differentiable outputs: no zero tensors for their missing gradients
    ctx.set_materialize_grads(False)
    ctx.n_params = len(params)
    ctx.h0_dtype = h0.dtype
    ctx.save_for_backward(grad_h0, *grads)


def _reinforce_backward(ctx, g_loss, g_logp, g_gh0, g_grads):
    """The gradients were formed in the forward pass (for d loss = 1): scale them by the incoming
    gradient of the scalar loss.  logp, grad_h0 and grads are results, not differentiable outputs."""
    if not ctx.with_grads:
        raise RuntimeError('pmg::reinforce_loss was called with with_grads=False: no gradient was formed')
    if g_loss is None:
        return (None,) * 9
    grad_h0, *grads = ctx.saved_tensors
    gl = g_loss.to(torch.float32)
    scaled = list(torch._foreach_mul(grads, gl)) if grads else []      # one multi-tensor launch, not 4L+7
    return (None, (grad_h0 * gl).to(ctx.h0_dtype), None, None, scaled, None, None, None, None)


_reinforce_loss_op.register_autograd(_reinforce_backward, setup_context=_reinforce_setup)


class TrainSlice:
    """The training pair of ONE slice of a batch with the two halves callable apart, on the stream
    that is current when they are called: construction runs pmg_logprob_fwd (it needs the tokens and
    h0 only), ``backward(reward, scale)`` runs pmg_reinforce_bwd once the rewards exist.  Workspace and
    gradient buffers are the slice's own, so slices on different streams share nothing.  No autograd:
    the caller sums ``grads`` over its slices and feeds ``grad_h0`` to the graph that produced h0.
    The rows of a slice must fit one workspace (it is one chunk)."""

    def __init__(self, weights: DecoderWeights, tokens: Tensor, h0: Tensor):
        self.params = [p.detach() for p in weights.flat()]
        d = self.d = _dims_of(self.params, weights.dims['D'], weights.pad_idx)
        self.tokens = tokens.to(torch.int64).contiguous()
        self.h0 = h0.detach().to(torch.float32).contiguous()
        T, B = _check_train_inputs(self.tokens, self.h0, d)
        self.shape = _shape(d, B, T)
        dev = self.h0.device
        self.ws = _alloc_ws(self.shape, _ffi.PMG_MODE_TRAIN, dev)
        self.wstruct, self._keep = _wstruct(self.params, d['L'])
        self.logp = torch.empty(T - 1, B, dtype=torch.float32, device=dev)
        # slices of a pipelined step run side by side on several streams: never the clustered kernels, whose
        # cluster barriers need every member resident (include/pmg.h, PMG_FLAG_NO_CLUSTER)
        _fwd_call(self.shape, self.wstruct, self.tokens, self.h0, self.logp, self.ws, _ffi.PMG_FLAG_NO_CLUSTER)
        self.grads = self.grad_h0 = self.loss = None

    def backward(self, reward: Tensor, scale: float) -> None:
        d, dev = self.d, self.h0.device
        B = self.h0.shape[1]
        self.reward = reward.detach().to(torch.float32).reshape(-1).contiguous()
        if self.reward.numel() != B:
            raise _ffi.PmgError('pmg: one reward per sequence required')
        self.grads = [torch.empty_like(p, dtype=torch.float32) for p in self.params]
        gstruct, keep = _grad_struct(self.grads, d['L'])
        self.grad_h0 = torch.empty(d['L'], B, d['H'], dtype=torch.float32, device=dev)
        self.loss = torch.empty(1, dtype=torch.float32, device=dev)
        rc = _ffi.lib.pmg_reinforce_bwd(C.byref(self.shape), C.byref(self.wstruct), _ffi._ptr(self.tokens),
                                        _ffi._ptr(self.reward), float(scale), _ffi.PMG_FLAG_NO_CLUSTER,
                                        C.byref(gstruct),
                                        _ffi._ptr(self.grad_h0), _ffi._ptr(self.loss), _ffi._ptr(self.ws),
                                        _ffi.current_stream())
        _ffi.check(rc, 'pmg_reinforce_bwd')
        del keep


def check_token_range(tokens: Tensor, V: int) -> None:
    """Host-side validation of token ids (one device->host round trip): the kernels only clamp."""
    if tokens.numel() == 0:
        return
    lo, hi = int(tokens.min()), int(tokens.max())
    if lo < 0 or hi >= V:
        raise _ffi.PmgError(f'pmg: token ids must lie in [0, {V}), got [{lo}, {hi}]')


def reinforce_loss(weights: DecoderWeights, tokens: Tensor, h0: Tensor, reward: Tensor, scale: float,
                   chunk: int = 0, trusted_tokens: bool = False):
    """Returns (loss, logp[T-1,B]).  tokens [T,B] (pad_sequence layout), h0 [L,B,H], reward [B].

    Differentiable w.r.t. h0 and the decoder parameters.  ``chunk`` > 0 fixes the number of sequences
    per kernel call; 0 lets the workspace budget decide.  ``trusted_tokens`` skips the id range check
    (for rows cut from pmg_decode's own output)."""
    if not trusted_tokens:
        check_token_range(tokens, weights.dims['V'])
    params = weights.flat()
    with_grads = torch.is_grad_enabled() and (h0.requires_grad or any(p.requires_grad for p in params))
    loss, logp, _, _ = _reinforce_loss_op(tokens, h0, reward, float(scale), params, weights.dims['D'],
                                          weights.pad_idx, int(chunk), bool(with_grads))
    return loss, logp.detach()      # the log-probs are a result, not a differentiable output


@torch.no_grad()
def teacher_forced_logprobs(weights: DecoderWeights, tokens: Tensor, h0: Tensor, chunk: int = 0) -> Tensor:
    """logp [T-1,B] without forming gradients."""
    B = tokens.shape[1]
    reward = torch.zeros(B, dtype=torch.float32, device=h0.device)
    _, logp = reinforce_loss(weights, tokens, h0, reward, 1.0, chunk=chunk)
    return logp


# --------------------------------------------------------------------------------- pmg::detokenize
@torch.library.custom_op('pmg::detokenize', mutates_args=('out_off', 'out_bytes'))
def _detokenize_op(tokens: Tensor, lens: Tensor, tab_off: Tensor, tab_bytes: Tensor, max_tok_bytes: int,
                   out_off: Tensor, out_bytes: Tensor) -> None:
    _need_cuda(tokens, lens, tab_off, tab_bytes, out_off, out_bytes)
    B, N = tokens.shape
    rc = _ffi.lib.pmg_detokenize(_ffi._ptr(tokens), N, _ffi._ptr(lens), B, N, _ffi._ptr(tab_off),
                                 _ffi._ptr(tab_bytes), tab_off.numel() - 1, int(max_tok_bytes), _ffi._ptr(out_off),
                                 _ffi._ptr(out_bytes), out_bytes.numel(), _ffi.current_stream())
    _ffi.check(rc, 'pmg_detokenize')


class TokenTable:
    """index -> token string table of a SMILES/SELFIES language, resident on the device in the
    layout pmg_detokenize reads (tab_off [V+1] int32, tab_bytes uint8, UTF-8).

    ``index_to_token``: dict {int: str} (pytoda's ``SMILESLanguage.index_to_token``) or a list.
    ``drop``: indexes whose text is dropped (e.g. pad/start/stop), like entries missing from the dict."""

    def __init__(self, index_to_token, device, drop=()):
        items = dict(enumerate(index_to_token)) if isinstance(index_to_token, (list, tuple)) \
            else {int(k): v for k, v in dict(index_to_token).items()}
        drop = set(int(i) for i in drop)
        self.V = (max(items) + 1) if items else 1
        enc = [b'' if (i in drop or i not in items) else str(items[i]).encode('utf-8') for i in range(self.V)]
        off = [0]
        for e in enc:
            off.append(off[-1] + len(e))
        self.max_tok_bytes = max(1, max((len(e) for e in enc), default=1))
        self.ascii = all(e.isascii() for e in enc)
        blob = b''.join(enc) or b'\0'
        self.tab_off = torch.tensor(off, dtype=torch.int32, device=device)
        self.tab_bytes = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        self.device = torch.device(device)
        self._out: Dict[tuple, tuple] = {}

    def buffers(self, B: int, max_len: int):
        cap = B * max_len * self.max_tok_bytes
        key = (B, max_len)
        buf = self._out.get(key)
        if buf is None:
            self._out.clear()
            buf = (torch.empty(B + 1, dtype=torch.int64, device=self.device),
                   torch.empty(cap, dtype=torch.uint8, device=self.device))
            self._out[key] = buf
        return buf


@torch.no_grad()
def detokenize_device(table: TokenTable, tokens: Tensor, lens: Tensor):
    """Stream-ordered, no synchronisation: (out_off [B+1] int64, out_bytes uint8) on the device;
    row b is out_bytes[out_off[b]:out_off[b+1]].  The buffers belong to ``table`` and are reused by
    the next call with the same (B, max_len)."""
    if tokens.dtype != torch.int64 or lens.dtype != torch.int32:
        raise _ffi.PmgError('pmg_detokenize: tokens must be int64 and lens int32')
    tokens = tokens.contiguous()
    lens = lens.contiguous()
    B, N = tokens.shape
    if lens.numel() != B:
        raise _ffi.PmgError('pmg_detokenize: one length per row required')
    out_off, out_bytes = table.buffers(B, N)
    _detokenize_op(tokens, lens, table.tab_off, table.tab_bytes, table.max_tok_bytes, out_off, out_bytes)
    return out_off, out_bytes


def detokenize(table: TokenTable, tokens: Tensor, lens: Tensor) -> List[str]:
    """Batched twin of ``[lang.token_indexes_to_smiles(row[:n].tolist()) for row, n in ...]``
    (reinforce.py:172-176): one device pass, two device->host copies, B string slices."""
    if tokens.shape[0] == 0:
        return []
    out_off, out_bytes = detokenize_device(table, tokens, lens)
    off = out_off.cpu().tolist()
    blob = out_bytes[:off[-1]].cpu().numpy().tobytes()
    if table.ascii:
        text = blob.decode('ascii')
        return [text[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    return [blob[off[i]:off[i + 1]].decode('utf-8') for i in range(len(off) - 1)]
