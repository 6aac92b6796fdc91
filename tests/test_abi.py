"""The C-ABI library loads on a machine without a GPU and exports every entry point that
include/pmg.h declares; the host-only entry points (shape check, workspace size, error
text) behave as documented.  No compute call is made."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'pmg.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pmg_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted([
        'pmg_abi_version', 'pmg_strerror', 'pmg_check_shape', 'pmg_workspace_bytes', 'pmg_decode',
        'pmg_logprob_fwd', 'pmg_reinforce_bwd', 'pmg_step', 'pmg_detokenize', 'pmg_decode_noalloc',
        'pmg_noalloc_scratch_bytes', 'pmg_reinforce_bptt', 'pmg_reinforce_wgrad', 'pmg_uses_cluster'])


def test_library_exports_every_declared_symbol():
    from paccmann_generator_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH)
    lib = C.CDLL(_ffi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted(_ffi.EXPORTS) == declared_functions()
    assert lib.pmg_abi_version() == _ffi.PMG_ABI_VERSION


def test_header_compiles_as_plain_c(tmp_path):
    """The boundary is a C ABI: the header must be valid C99 without HIP or torch."""
    import subprocess
    src = tmp_path / 't.c'
    src.write_text('#include "pmg.h"\nint main(void){pmg_shape s; pmg_weights w; pmg_grads g; (void)s;(void)w;(void)g;'
                   'return sizeof(s) == 40 ? 0 : 1;}\n')
    exe = tmp_path / 't'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o',
                           str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_ctypes_structs_match_the_header_layout():
    from paccmann_generator_amd import _ffi
    assert C.sizeof(_ffi.PmgShape) == 40
    ptr = C.sizeof(C.c_void_p)
    assert C.sizeof(_ffi.PmgWeights) == ptr * (1 + 4 * _ffi.PMG_MAX_LAYERS + 8)
    assert C.sizeof(_ffi.PmgGrads) == ptr * (1 + 4 * _ffi.PMG_MAX_LAYERS + 6)


def test_shape_checks_and_error_text():
    from paccmann_generator_amd import _ffi
    ok = _ffi.make_shape(4096, 130, 512, 128, 50, 50, 60, 1, 256, 0)
    assert _ffi.lib.pmg_check_shape(C.byref(ok), _ffi.PMG_MODE_TRAIN) == 0
    assert _ffi.lib.pmg_check_shape(C.byref(ok), _ffi.PMG_MODE_DECODE) == 0
    assert _ffi.lib.pmg_check_shape(C.byref(ok), 7) < 0
    bad = [
        _ffi.make_shape(8, 4, 100, 16, 10, 7, 23, 1, 32, 0),     # H not supported
        _ffi.make_shape(8, 4, 128, 16, 65, 7, 23, 1, 32, 0),     # W > 64
        _ffi.make_shape(8, 4, 128, 16, 10, 7, 1025, 1, 32, 0),   # V > 1024
        _ffi.make_shape(8, 4, 128, 16, 10, 7, 23, 5, 32, 0),     # L > 4
        _ffi.make_shape(0, 4, 128, 16, 10, 7, 23, 1, 32, 0),     # empty batch
        _ffi.make_shape(8, 1, 128, 16, 10, 7, 23, 1, 32, 0),     # train needs T >= 2
    ]
    codes = set()
    for s in bad:
        rc = _ffi.lib.pmg_check_shape(C.byref(s), _ffi.PMG_MODE_TRAIN)
        assert rc < 0
        codes.add(rc)
        assert len(_ffi.lib.pmg_strerror(rc)) > 3
        with pytest.raises(_ffi.PmgError):
            _ffi.workspace_bytes(s, _ffi.PMG_MODE_TRAIN)
    assert len(codes) >= 5
    assert _ffi.lib.pmg_check_shape(None, 0) < 0
    # the envelope of StackGRUDecoder(params) (train_paccmann_rl.py:93-97): every multiple of 128 up to
    # 1024 as hidden size, vocabularies of a few hundred tokens
    for H in range(128, 1025, 128):
        for V in (23, 60, 300, 575, 1024):
            s = _ffi.make_shape(64, 102, H, 128, 50, 50, V, 1, 256, 0)
            assert _ffi.lib.pmg_check_shape(C.byref(s), _ffi.PMG_MODE_TRAIN) == 0, (H, V)
            assert _ffi.lib.pmg_check_shape(C.byref(s), _ffi.PMG_MODE_DECODE) == 0, (H, V)
    # ... up to what one CU's LDS can hold: 4 layers of hidden 1024 next to a 1024-token vocabulary cannot
    s = _ffi.make_shape(64, 102, 1024, 128, 50, 50, 1024, 4, 256, 0)
    rc = _ffi.lib.pmg_check_shape(C.byref(s), _ffi.PMG_MODE_TRAIN)
    assert rc < 0 and b'LDS' in _ffi.lib.pmg_strerror(rc)


def test_workspace_size_scales_with_batch_and_time():
    from paccmann_generator_amd import _ffi
    a = _ffi.workspace_bytes(_ffi.make_shape(4096, 130, 512, 128, 50, 50, 60, 1, 256, 0), _ffi.PMG_MODE_TRAIN)
    b = _ffi.workspace_bytes(_ffi.make_shape(2048, 130, 512, 128, 50, 50, 60, 1, 256, 0), _ffi.PMG_MODE_TRAIN)
    d = _ffi.workspace_bytes(_ffi.make_shape(4096, 128, 512, 128, 50, 50, 60, 1, 256, 0), _ffi.PMG_MODE_DECODE)
    assert 10e9 < a < 24e9 and 0.45 < b / a < 0.6
    assert d < 0.2e9   # decode keeps two stack slabs per row group, no history


def test_cpu_tensors_are_rejected_not_routed_elsewhere():
    import torch
    from paccmann_generator_amd import _ffi
    with pytest.raises(_ffi.PmgError):
        _ffi._ptr(torch.zeros(4))


def test_default_library_has_no_ablation_switches():
    """The shipped libpmg.so reads no environment variable: the phase-skipping bits of the timing
    experiments exist only in -DPMG_ABLATE / -DPMG_PROFILE variant builds."""
    from paccmann_generator_amd import _ffi
    default = os.path.join(ROOT, 'paccmann_generator_amd', 'lib', 'libpmg.so')
    blob = open(default, 'rb').read()
    assert b'PMG_DEBUG' not in blob
    import subprocess
    syms = subprocess.run(['nm', '-D', '--undefined-only', default], capture_output=True, text=True).stdout
    assert 'getenv' not in syms
    assert os.path.samefile(_ffi.LIB_PATH, default) or os.environ.get('PMG_LIB')


def test_cluster_path_selection_by_shape():
    """Which launches take the clustered small-batch kernels (include/pmg.h, pmg_uses_cluster)."""
    from paccmann_generator_amd import _ffi
    def uses(B, H, L=1, V=60, flags=0, mode=_ffi.PMG_MODE_DECODE, T=100):
        s = _ffi.make_shape(B, T, H, 128, 50, 50, V, L, 256, 0)
        return _ffi.lib.pmg_uses_cluster(C.byref(s), mode, flags)
    TR = _ffi.PMG_MODE_TRAIN
    assert uses(64, 256) == 1 and uses(64, 256, mode=TR, T=102) == 3      # BASELINE configs[0]: forward and BPTT
    # 16 clusters of 16 CUs fill the chip at hidden 256; above that a cluster serves several row groups, up to
    # four (beyond, the batch-partitioned kernels are as fast); the BPTT's owned sequences fit the LDS up to three
    assert uses(256, 256) == 1 and uses(257, 256) == 1 and uses(1024, 256) == 1 and uses(1025, 256) == 0
    assert uses(768, 256, mode=TR) == 3 and uses(769, 256, mode=TR) == 1 and uses(1025, 256, mode=TR) == 0
    # conditional_generator.json:12 at hidden 512: 8 clusters of 32 CUs, up to six row groups each (LDS)
    assert uses(128, 512) == 1 and uses(768, 512) == 1 and uses(769, 512) == 0
    assert uses(128, 512, mode=TR) == 3 and uses(768, 512, mode=TR) == 3 and uses(769, 512, mode=TR) == 0
    assert uses(512, 128) == 1 and uses(513, 128) == 0          # hidden 128: one row group per cluster only
    assert uses(96, 384) == 1 and uses(896, 384) == 1 and uses(897, 384) == 0
    assert uses(512, 384, mode=TR) == 3 and uses(513, 384, mode=TR) == 1
    assert uses(4096, 512) == 0 and uses(4096, 512, mode=TR) == 0          # configs[1]: batch-partitioned
    assert uses(64, 256, L=2) == 0 and uses(64, 640) == 0 and uses(64, 256, V=300) == 0
    assert uses(64, 256, flags=_ffi.PMG_FLAG_NO_CLUSTER) == 0
    assert uses(64, 256, mode=TR, flags=_ffi.PMG_FLAG_NO_CLUSTER) == 0
    assert uses(64, 100) < 0
