"""ctypes binding of the C ABI in include/pmg.h (lib/libpmg.so, built by __graft_entry__.build()).

PyTorch is used for device memory and streams only: every call passes raw device
pointers (``tensor.data_ptr()``) and the current HIP stream.  There is no CPU
fallback: if the shared library is missing the import of the product package fails.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

PMG_MAX_LAYERS = 4
PMG_MODE_DECODE = 0
PMG_MODE_TRAIN = 1
PMG_FLAG_NO_EARLY_EXIT = 1
PMG_FLAG_ACCUMULATE = 2
PMG_FLAG_REUSE_PACK = 16
PMG_FLAG_NO_CLUSTER = 32
PMG_ABI_VERSION = 3

_HERE = os.path.dirname(os.path.abspath(__file__))
# PMG_LIB selects a tuning variant built by __graft_entry__.build(extra_flags=..., libname=...)
LIB_PATH = os.environ.get('PMG_LIB') or os.path.join(_HERE, 'lib', 'libpmg.so')


class PmgShape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'T', 'H', 'E', 'W', 'D', 'V', 'L', 'Z', 'pad_idx')]


_F = C.c_void_p  # const float*


class PmgWeights(C.Structure):
    _fields_ = [
        ('embedding', _F),
        ('w_ih', _F * PMG_MAX_LAYERS), ('w_hh', _F * PMG_MAX_LAYERS),
        ('b_ih', _F * PMG_MAX_LAYERS), ('b_hh', _F * PMG_MAX_LAYERS),
        ('w_ctrl', _F), ('b_ctrl', _F), ('w_push', _F), ('b_push', _F),
        ('w_out', _F), ('b_out', _F), ('w_l2h', _F), ('b_l2h', _F),
    ]


class PmgGrads(C.Structure):
    _fields_ = [
        ('embedding', _F),
        ('w_ih', _F * PMG_MAX_LAYERS), ('w_hh', _F * PMG_MAX_LAYERS),
        ('b_ih', _F * PMG_MAX_LAYERS), ('b_hh', _F * PMG_MAX_LAYERS),
        ('w_ctrl', _F), ('b_ctrl', _F), ('w_push', _F), ('b_push', _F),
        ('w_out', _F), ('b_out', _F),
    ]


EXPORTS = (
    'pmg_abi_version', 'pmg_strerror', 'pmg_check_shape', 'pmg_workspace_bytes',
    'pmg_decode', 'pmg_logprob_fwd', 'pmg_reinforce_bwd', 'pmg_step', 'pmg_detokenize',
    'pmg_decode_noalloc', 'pmg_noalloc_scratch_bytes', 'pmg_reinforce_bptt', 'pmg_reinforce_wgrad', 'pmg_uses_cluster',
)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build the HIP library first '
            f'(python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f'{LIB_PATH} does not export {name}')
    lib.pmg_abi_version.restype = C.c_int
    lib.pmg_strerror.restype = C.c_char_p
    lib.pmg_strerror.argtypes = [C.c_int]
    lib.pmg_check_shape.argtypes = [C.POINTER(PmgShape), C.c_int]
    lib.pmg_uses_cluster.argtypes = [C.POINTER(PmgShape), C.c_int, C.c_uint32]
    lib.pmg_workspace_bytes.argtypes = [C.POINTER(PmgShape), C.c_int, C.POINTER(C.c_size_t)]
    lib.pmg_decode.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
        C.c_uint64, C.c_uint64, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pmg_logprob_fwd.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint32, C.c_void_p, C.c_void_p]
    lib.pmg_reinforce_bwd.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_void_p, C.c_float, C.c_uint32,
        C.POINTER(PmgGrads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pmg_reinforce_bptt.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p,
        C.c_void_p, C.c_void_p]
    lib.pmg_reinforce_wgrad.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_uint32, C.POINTER(PmgGrads), C.c_void_p,
        C.c_void_p, C.c_void_p]
    lib.pmg_step.argtypes = [
        C.POINTER(PmgShape), C.POINTER(PmgWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pmg_detokenize.argtypes = [
        C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pmg_decode_noalloc.argtypes = lib.pmg_decode.argtypes[:-2] + [C.c_void_p]
    lib.pmg_noalloc_scratch_bytes.argtypes = []
    for name in EXPORTS[2:]:
        getattr(lib, name).restype = C.c_int
    lib.pmg_noalloc_scratch_bytes.restype = C.c_size_t
    if lib.pmg_abi_version() != PMG_ABI_VERSION:
        raise ImportError('libpmg.so ABI version mismatch')
    return lib


lib = _load()


class PmgError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise PmgError(f'{what}: {lib.pmg_strerror(rc).decode()} (code {rc})')
    raise PmgError(f'{what}: HIP error {rc}')


def make_shape(B, T, H, E, W, D, V, L, Z, pad_idx) -> PmgShape:
    return PmgShape(int(B), int(T), int(H), int(E), int(W), int(D), int(V), int(L), int(Z),
                    -1 if pad_idx is None else int(pad_idx))


def workspace_bytes(shape: PmgShape, mode: int) -> int:
    out = C.c_size_t(0)
    check(lib.pmg_workspace_bytes(C.byref(shape), mode, C.byref(out)), 'pmg_workspace_bytes')
    return int(out.value)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise PmgError('pmg: tensors must live on the GPU (there is no CPU path)')
    if not t.is_contiguous():
        raise PmgError('pmg: tensors must be contiguous')
    return C.c_void_p(t.data_ptr())


def _f32(t):
    if t.dtype != torch.float32:
        raise PmgError('pmg: fp32 tensors required')
    return t


def pack_weight_struct(cls, tensors: dict, n_layers: int):
    """tensors: names as in pmg_weights; per-layer entries are lists."""
    s = cls()
    keep = []
    for name, _ in cls._fields_:
        v = tensors.get(name)
        if v is None:
            continue
        if isinstance(v, (list, tuple)):
            arr = getattr(s, name)
            for l in range(n_layers):
                t = _f32(v[l]).contiguous()
                keep.append(t)
                arr[l] = t.data_ptr()
        else:
            t = _f32(v).contiguous()
            keep.append(t)
            setattr(s, name, t.data_ptr())
    return s, keep


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
