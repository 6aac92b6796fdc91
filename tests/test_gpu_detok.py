"""pmg_detokenize through the C ABI against oracle/detok.py: byte-exact packed text and offsets,
ragged / empty rows, ids without a table entry, multi-byte UTF-8, row counts that are not a multiple
of the 8 rows a workgroup takes and that need several passes of the offset scan, the full decode
batch; and Reinforce.get_smiles_from_latent giving the same result through the device detokeniser
(language with index_to_token) as through the reference's per-row host loop."""
import ctypes as C
import logging

import numpy as np
import pytest
import torch

from oracle import detok as OD
from tests import parity as P

pytestmark = pytest.mark.gpu

SELFIES = ['<PAD>', '<UNK>', '<START>', '<STOP>', '[C]', '[N]', '[O]', '[=C]', '[Branch1_1]', '[Ring1]', '[#N]',
           '[Expl=Ring2]', '[S@@Hexpl]', 'C', 'c', '(', ')', '1', '', 'Cl', 'Br', '[nH]', 'µ', '→x', '[F]']


def _run(table, tokens, lens):
    from paccmann_generator_amd import ops
    off, blob = ops.detokenize_device(table, tokens, lens)
    torch.cuda.synchronize()
    off = off.cpu().tolist()
    return off, blob[:off[-1]].cpu().numpy().tobytes()


@pytest.mark.parametrize('B,N', [(1, 1), (7, 5), (8, 64), (37, 65), (300, 130), (4096, 128), (70001, 9)])
def test_packed_bytes_match_oracle(gpu, B, N):
    from paccmann_generator_amd import ops
    rng = np.random.RandomState(B + N)
    itt = {i: t for i, t in enumerate(SELFIES) if i != 6}          # id 6 has no entry
    table = ops.TokenTable(itt, 'cuda', drop=(0, 2, 3))
    tok = rng.randint(-2, len(SELFIES) + 3, size=(B, N)).astype(np.int64)   # some ids outside [0,V)
    lens = rng.randint(0, N + 1, size=B).astype(np.int32)
    lens[0] = N
    if B > 2:
        lens[1] = 0
        lens[2] = N + 5                                            # clamped to N
    off, blob = _run(table, torch.from_numpy(tok).cuda(), torch.from_numpy(lens).cuda())
    want = OD.detokenize_rows(itt, tok.tolist(), lens.tolist(), drop=(0, 2, 3))
    w_off, w_blob = OD.pack(want)
    assert off == w_off
    assert blob == w_blob
    assert ops.detokenize(table, torch.from_numpy(tok).cuda(), torch.from_numpy(lens).cuda()) == want


def test_rejects_bad_arguments(gpu):
    from paccmann_generator_amd import _ffi, ops
    table = ops.TokenTable(SELFIES, 'cuda')
    tok = torch.zeros(4, 8, dtype=torch.int64, device='cuda')
    lens = torch.zeros(4, dtype=torch.int32, device='cuda')
    off, blob = table.buffers(4, 8)
    args = lambda cap, b=4: (_ffi._ptr(tok), 8, _ffi._ptr(lens), b, 8, _ffi._ptr(table.tab_off),
                             _ffi._ptr(table.tab_bytes), table.V, table.max_tok_bytes, _ffi._ptr(off),
                             _ffi._ptr(blob), cap, _ffi.current_stream())
    assert _ffi.lib.pmg_detokenize(*args(blob.numel())) == 0
    assert _ffi.lib.pmg_detokenize(*args(blob.numel() - 1)) < 0          # worst case would not fit
    assert _ffi.lib.pmg_detokenize(*args(blob.numel(), 0)) < 0           # empty batch
    a = list(args(blob.numel()))
    a[0] = None
    assert _ffi.lib.pmg_detokenize(*a) < 0
    with pytest.raises(_ffi.PmgError):
        ops.detokenize(table, tok.to(torch.int32), lens)
    torch.cuda.synchronize()


class _Lang:
    start_index, stop_index = P.START, P.STOP

    def __init__(self, with_table):
        self._t = {i: ('' if i < 4 else SELFIES[4 + (i % (len(SELFIES) - 4))]) for i in range(P.TINY['vocab_size'])}
        if with_table:
            self.index_to_token = self._t
        self.calls = 0

    def token_indexes_to_smiles(self, idx):
        self.calls += 1
        return ''.join(self._t.get(i, '') for i in idx)

    def selfies_to_smiles(self, s):
        return s


def test_get_smiles_from_latent_device_path_equals_host_loop(gpu, tmp_path):
    from paccmann_generator_amd import Generator, Reinforce
    params = dict(P.TINY, latent_dim=32)
    _, pdec = P.make_pair(params, seed=0)
    hp = dict(generate_len=21, temperature=1.1, model_folder=str(tmp_path / 'm'), save_untrained=False)
    res = []
    for with_table in (True, False):
        gen = Generator(None, pdec)
        lang = _Lang(with_table)
        gen._associate_language(lang)
        gen.seed(3)
        learner = Reinforce(gen, None, hp, 'm', logging.getLogger('t'), is_valid=lambda s: len(s) % 3 != 0)
        torch.manual_seed(0)
        latent = torch.randn(1, 53, 32, device='cuda')
        smiles, nums, valid = learner.get_smiles_from_latent(latent)
        res.append((smiles, [n.tolist() for n in nums], valid))
        assert (lang.calls == 0) == with_table            # the device path never calls the per-row method
        assert all(n.dtype == torch.int64 for n in nums)
    assert res[0] == res[1]
    assert 0 < len(res[0][2]) < 53
