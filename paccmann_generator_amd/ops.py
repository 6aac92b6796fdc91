"""Operators over the C ABI: sampling decode, teacher-forced log-probs, REINFORCE loss.

These are the three pieces of the reference loop that move to the GPU:
``generator.generate`` (reinforce.py:162-170), the teacher-forced loop
(reinforce_omics.py:319-332) and ``rl_loss.backward()`` (:335).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _ffi

_WS: Dict[tuple, torch.Tensor] = {}


def _workspace(shape, mode: int, device, private: bool = False) -> torch.Tensor:
    need = _ffi.workspace_bytes(shape, mode)
    if private:
        return torch.empty(need, dtype=torch.uint8, device=device)
    key = (str(device), mode)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = None
        _WS.pop(key, None)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def release_workspaces() -> None:
    _WS.clear()


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

    def flat(self) -> List[torch.Tensor]:
        """Parameters in the fixed order used by ReinforceLossFn (latent_to_hidden excluded)."""
        out = []
        for n in self.NAMES:
            v = self.t[n]
            out.extend(v if isinstance(v, list) else [v])
        return out

    @staticmethod
    def unflatten(flat: List[torch.Tensor], L: int) -> dict:
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


@torch.no_grad()
def decode(weights: DecoderWeights, latent: torch.Tensor, start_tok: int, stop_tok: int, gen_len: int,
           temperature: float = 1.0, seed: int = 0, offset: int = 0, seq_index_base: int = 0,
           early_exit: bool = True):
    """Returns (tokens [B,gen_len] int64, lengths [B] int32).  latent: [B,Z] or [1,B,Z]."""
    if latent.dim() == 3:
        latent = latent[0]
    latent = latent.detach().to(torch.float32).contiguous()
    B = latent.shape[0]
    shape = weights.shape(B, gen_len)
    ws = _workspace(shape, _ffi.PMG_MODE_DECODE, latent.device)
    tokens = torch.empty(B, gen_len, dtype=torch.int64, device=latent.device)
    lens = torch.empty(B, dtype=torch.int32, device=latent.device)
    wstruct, keep = weights.struct(True)
    flags = 0 if early_exit else _ffi.PMG_FLAG_NO_EARLY_EXIT
    rc = _ffi.lib.pmg_decode(C.byref(shape), C.byref(wstruct), _ffi._ptr(latent), int(start_tok), int(stop_tok),
                             int(gen_len), float(temperature), int(seed), int(offset), int(seq_index_base),
                             flags, _ffi._ptr(tokens), _ffi._ptr(lens), _ffi._ptr(ws), _ffi.current_stream())
    _ffi.check(rc, 'pmg_decode')
    del keep
    return tokens, lens


@torch.no_grad()
def step(weights: DecoderWeights, tok: torch.Tensor, hidden: torch.Tensor, stack: Optional[torch.Tensor],
         want_logits: bool = True):
    """One decoder step.  tok [B] or [1,B]; hidden [L,B,H]; stack [B,D,W] or None (zeros)."""
    tok = tok.reshape(-1).to(torch.int64).contiguous()
    hidden = hidden.detach().to(torch.float32).contiguous()
    B = tok.shape[0]
    d = weights.dims
    shape = weights.shape(B, 1)
    ws = _workspace(shape, _ffi.PMG_MODE_DECODE, hidden.device)
    h_out = torch.empty_like(hidden)
    s_out = torch.empty(B, d['D'], d['W'], dtype=torch.float32, device=hidden.device)
    logits = torch.empty(B, d['V'], dtype=torch.float32, device=hidden.device) if want_logits else None
    s_in = None if stack is None else stack.detach().to(torch.float32).contiguous()
    wstruct, keep = weights.struct(False)
    rc = _ffi.lib.pmg_step(C.byref(shape), C.byref(wstruct), _ffi._ptr(tok), _ffi._ptr(hidden), _ffi._ptr(s_in),
                           _ffi._ptr(h_out), _ffi._ptr(s_out), _ffi._ptr(logits), _ffi._ptr(ws),
                           _ffi.current_stream())
    _ffi.check(rc, 'pmg_step')
    del keep
    return h_out, s_out, logits


class ReinforceLossFn(torch.autograd.Function):
    """loss = -scale * sum_{p,b} reward[b] * logp[p,b]  (reinforce_omics.py:322-332 with
    mean over the valid molecules = scale * sum), differentiable w.r.t. the decoder
    parameters and h0.  Forward = pmg_logprob_fwd, backward = pmg_reinforce_bwd."""

    @staticmethod
    def forward(ctx, tokens, h0, reward, scale, dims, *params):
        L = dims['L']
        tokens = tokens.to(torch.int64).contiguous()
        h0c = h0.detach().to(torch.float32).contiguous()
        reward = reward.detach().to(torch.float32).reshape(-1).contiguous()
        T, B = tokens.shape
        shape = _ffi.make_shape(B, T, dims['H'], dims['E'], dims['W'], dims['D'], dims['V'], L, dims['Z'],
                                dims['pad_idx'])
        ws = _workspace(shape, _ffi.PMG_MODE_TRAIN, h0c.device, private=True)
        wt = DecoderWeights.unflatten([p.detach() for p in params], L)
        wstruct, keep = _ffi.pack_weight_struct(_ffi.PmgWeights, wt, L)
        logp = torch.empty(T - 1, B, dtype=torch.float32, device=h0c.device)
        rc = _ffi.lib.pmg_logprob_fwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(h0c),
                                      _ffi._ptr(logp), None, _ffi._ptr(ws), _ffi.current_stream())
        _ffi.check(rc, 'pmg_logprob_fwd')
        ctx.shape = shape
        ctx.ws = ws
        ctx.L = L
        ctx.scale = float(scale)
        ctx.save_for_backward(tokens, reward, *[p.detach() for p in params])
        ctx.mark_non_differentiable(logp)
        loss = -(logp * reward.unsqueeze(0)).sum() * float(scale)
        return loss, logp

    @staticmethod
    def backward(ctx, grad_loss, _grad_logp):
        tokens, reward, *params = ctx.saved_tensors
        L = ctx.L
        wt = DecoderWeights.unflatten(list(params), L)
        wstruct, keep = _ffi.pack_weight_struct(_ffi.PmgWeights, wt, L)
        grads = [torch.empty_like(p) for p in params]
        gt = DecoderWeights.unflatten(grads, L)
        gstruct, keep2 = _ffi.pack_weight_struct(_ffi.PmgGrads, gt, L)
        shape = ctx.shape
        grad_h0 = torch.empty(L, shape.B, shape.H, dtype=torch.float32, device=tokens.device)
        loss = torch.empty(1, dtype=torch.float32, device=tokens.device)
        rc = _ffi.lib.pmg_reinforce_bwd(C.byref(shape), C.byref(wstruct), _ffi._ptr(tokens), _ffi._ptr(reward),
                                        ctx.scale, 0, C.byref(gstruct), _ffi._ptr(grad_h0), _ffi._ptr(loss),
                                        _ffi._ptr(ctx.ws), _ffi.current_stream())
        _ffi.check(rc, 'pmg_reinforce_bwd')
        ctx.ws = None
        ctx.bwd_loss = loss
        gl = grad_loss.to(torch.float32)
        grads = [g * gl for g in grads]
        return (None, grad_h0 * gl, None, None, None, *grads)


def reinforce_loss(weights: DecoderWeights, tokens: torch.Tensor, h0: torch.Tensor, reward: torch.Tensor,
                   scale: float):
    """Returns (loss, logp[T-1,B]).  tokens [T,B] (pad_sequence layout), h0 [L,B,H], reward [B]."""
    return ReinforceLossFn.apply(tokens, h0, reward, scale, weights.dims, *weights.flat())


@torch.no_grad()
def teacher_forced_logprobs(weights: DecoderWeights, tokens: torch.Tensor, h0: torch.Tensor) -> torch.Tensor:
    """logp [T-1,B] without keeping a graph."""
    T, B = tokens.shape
    reward = torch.zeros(B, dtype=torch.float32, device=h0.device)
    _, logp = ReinforceLossFn.apply(tokens, h0, reward, 1.0, weights.dims, *[p.detach() for p in weights.flat()])
    return logp


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
def detokenize_device(table: TokenTable, tokens: torch.Tensor, lens: torch.Tensor):
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
    rc = _ffi.lib.pmg_detokenize(_ffi._ptr(tokens), N, _ffi._ptr(lens), B, N, _ffi._ptr(table.tab_off),
                                 _ffi._ptr(table.tab_bytes), table.V, table.max_tok_bytes, _ffi._ptr(out_off),
                                 _ffi._ptr(out_bytes), out_bytes.numel(), _ffi.current_stream())
    _ffi.check(rc, 'pmg_detokenize')
    return out_off, out_bytes


def detokenize(table: TokenTable, tokens: torch.Tensor, lens: torch.Tensor) -> List[str]:
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
