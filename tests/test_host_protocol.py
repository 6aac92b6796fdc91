"""Host side of the drop-in protocol, no GPU needed: parameter names and
checkpoint round trips, the iterable generate() returns, the device twin of `+2 x stop` +
pad_sequence against the reference's own host construction, shard ranges, and the refusal to run
without a GPU (there is no CPU path)."""
import logging

import pytest
import torch
import torch.nn as nn

from paccmann_generator_amd import Generator, Reinforce, SamplingSearch, StackGRUDecoder, dp
from paccmann_generator_amd.decoder import GreedySearch, _CroppedRows
from paccmann_generator_amd.reinforce import pad_rows_on_device

PARAMS = dict(embedding_size=16, rnn_cell_size=128, vocab_size=23, stack_width=10, stack_depth=7, n_layers=2,
              latent_dim=32, pad_index=0)


def test_decoder_parameter_names_and_shapes():
    dec = StackGRUDecoder(PARAMS)
    keys = set(dec.state_dict())
    want = {'embedding.weight', 'stack_controls_layer.weight', 'stack_controls_layer.bias', 'stack_input_layer.weight',
            'stack_input_layer.bias', 'latent_to_hidden.weight', 'latent_to_hidden.bias', 'output_layer.weight',
            'output_layer.bias'}
    for l in range(2):
        want |= {f'gru.weight_ih_l{l}', f'gru.weight_hh_l{l}', f'gru.bias_ih_l{l}', f'gru.bias_hh_l{l}'}
    assert keys == want
    assert dec.gru.weight_ih_l0.shape == (3 * 128, 16 + 10) and dec.gru.weight_ih_l1.shape == (3 * 128, 128)
    assert dec.stack_controls_layer.weight.shape == (3, 128) and dec.stack_input_layer.weight.shape == (10, 128)
    assert float(dec.embedding.weight[0].detach().abs().sum()) == 0.0          # padding_idx row
    dec._update_batch_size(5)
    assert dec.init_stack.shape == (5, 7, 10) and dec.init_hidden.shape == (2, 5, 128)
    assert (dec.latent_dim, dec.n_layers) == (32, 2)
    with pytest.raises(ValueError):
        StackGRUDecoder(dict(PARAMS, bidirectional=True))
    with pytest.raises(ValueError):
        StackGRUDecoder(dict(PARAMS, use_stack=False))


def test_checkpoint_round_trip_with_and_without_encoder(tmp_path):
    torch.manual_seed(0)
    gen = Generator(nn.Linear(4, 4), StackGRUDecoder(PARAMS))
    path = str(tmp_path / 'gen.pt')
    gen.save(path)
    sd = torch.load(path, weights_only=True)
    assert {k.split('.')[0] for k in sd} == {'encoder', 'decoder'}     # TeacherVAE layout
    torch.manual_seed(1)
    full = Generator(nn.Linear(4, 4), StackGRUDecoder(PARAMS))
    full.load(path, map_location='cpu')
    bare = Generator(None, StackGRUDecoder(PARAMS))
    bare.load_model(path)                                              # encoder.* entries skipped
    for g in (full, bare):
        assert all(torch.equal(a, b) for a, b in zip(g.decoder.parameters(), gen.decoder.parameters()))
    with pytest.raises(RuntimeError):                                  # a decoder of another shape must not load
        Generator(None, StackGRUDecoder(dict(PARAMS, rnn_cell_size=256))).load(path)


def test_cropped_rows_iterable():
    toks = torch.tensor([[5, 6, 3, 9], [3, 3, 3, 3], [7, 8, 9, 4]])
    rows = list(iter(_CroppedRows(toks, torch.tensor([2, 0, 4], dtype=torch.int32))))
    assert [r.tolist() for r in rows] == [[5, 6], [], [7, 8, 9, 4]]
    assert all(r.dtype == torch.int64 for r in rows)
    assert SamplingSearch(0.8).temperature == 0.8 and not SamplingSearch().greedy and GreedySearch().greedy


def test_device_padding_equals_reference_host_construction():
    """reinforce.py:177-184 (cat with two stop tokens) + pad_sequence (reinforce_omics.py:304)."""
    g = torch.Generator().manual_seed(3)
    B, N, stop = 9, 12, 3
    toks = torch.randint(4, 20, (B, N), generator=g)
    lens = torch.tensor([0, 12, 5, 1, 12, 7, 3, 0, 11], dtype=torch.int32)
    rows = [torch.cat([toks[i, :n].long(), torch.tensor(2 * [stop])]) for i, n in enumerate(lens.tolist())]
    want = torch.nn.utils.rnn.pad_sequence(rows)
    assert torch.equal(pad_rows_on_device(toks, lens, stop), want)
    wide = pad_rows_on_device(toks, lens, stop, t_max=N + 2)
    assert wide.shape == (N + 2, B) and torch.equal(wide[:want.shape[0]], want)
    keep = torch.tensor([2, 3, 6])
    want_v = torch.nn.utils.rnn.pad_sequence([rows[i] for i in keep.tolist()])
    assert torch.equal(pad_rows_on_device(toks[keep], lens[keep], stop), want_v)


def test_shard_ranges_cover_the_batch():
    for n, world in ((4096, 8), (37, 4), (3, 4), (1, 1)):
        spans = [dp.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_reinforce_refuses_a_cpu_decoder(tmp_path):
    gen = Generator(None, StackGRUDecoder(PARAMS))
    with pytest.raises(RuntimeError, match='no CPU path'):
        Reinforce(gen, None, dict(model_folder=str(tmp_path)), 'm', logging.getLogger('t'))


def test_stop_rows_is_the_lazy_twin_of_the_reference_row_list():
    """reinforce.py:177-184: every sampled row gets 2 x [stop_index] appended.  StopRows builds those
    rows on demand from the decoded matrix; indexing, iteration, slicing and subset selection must
    give what the reference's list comprehension gives."""
    from paccmann_generator_amd.reinforce import StopRows
    g = torch.Generator().manual_seed(0)
    tokens = torch.randint(4, 20, (7, 9), generator=g)
    lens = torch.tensor([0, 3, 9, 1, 5, 9, 2], dtype=torch.int32)
    stop = 3
    ref = [torch.cat([tokens[i, :int(lens[i])], torch.tensor([stop, stop])]) for i in range(7)]
    rows = StopRows(tokens, lens, stop)
    assert len(rows) == 7
    assert all(torch.equal(a, b) for a, b in zip(rows, ref))
    assert torch.equal(rows[2], ref[2]) and torch.equal(rows[-1], ref[-1])
    sub = rows.subset([1, 4, 6])
    assert len(sub) == 3 and sub.index == [1, 4, 6]
    assert all(torch.equal(a, ref[i]) for a, i in zip(sub, [1, 4, 6]))
    assert [r.tolist() for r in rows[1:3]] == [ref[1].tolist(), ref[2].tolist()]
    padded = torch.nn.utils.rnn.pad_sequence(list(sub))          # reinforce_omics.py:304 accepts it
    assert padded.shape == (7, 3) and padded[:, 0].tolist() == ref[1].tolist() + [0, 0]


def test_poisoned_outputs_of_a_clustered_launch_are_reported():
    """A clustered launch that gives up on a barrier poisons its outputs (include/pmg.h): lengths of -1 after
    pmg_decode, NaN loss after the training pair.  The host mirror turns both into PmgError instead of
    cropping with negative lengths or logging NaN."""
    import pytest
    import torch
    from paccmann_generator_amd import _ffi
    from paccmann_generator_amd.decoder import _CroppedRows
    from paccmann_generator_amd.reinforce import StopRows, _loss_value
    tokens = torch.tensor([[5, 6, 3, 3], [7, 3, 3, 3]])
    rows = StopRows(tokens, torch.tensor([2, -1], dtype=torch.int32), 3)
    with pytest.raises(_ffi.PmgError, match='cluster barrier'):
        rows[0]
    with pytest.raises(_ffi.PmgError, match='cluster barrier'):
        _CroppedRows(tokens, torch.tensor([-1, -1], dtype=torch.int32))
    assert len(list(_CroppedRows(tokens, torch.tensor([2, 1], dtype=torch.int32)))) == 2
    assert _loss_value(torch.tensor(0.25)) == 0.25
    with pytest.raises(_ffi.PmgError, match='NaN'):
        _loss_value(torch.tensor(float('nan')))
