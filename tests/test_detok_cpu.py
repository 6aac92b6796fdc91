"""Host side of the detokeniser: the oracle's known answers and the device table layout
(offsets, dropped and missing ids, UTF-8 byte lengths).  No compute call is made."""
from oracle import detok as OD


def test_oracle_known_answers():
    itt = {0: '<PAD>', 2: '<START>', 3: '<STOP>', 4: 'C', 5: '(', 6: ')', 7: '=O', 9: 'µ'}
    assert OD.token_indexes_to_text(itt, [4, 4, 5, 7, 6, 4]) == 'CC(=O)C'
    assert OD.token_indexes_to_text(itt, [2, 4, 8, 99, -1, 3], drop=(2, 3)) == 'C'     # missing ids add nothing
    rows = OD.detokenize_rows(itt, [[4, 9, 4], [4, 4, 4], [7, 7, 7]], [3, 0, 7])
    assert rows == ['CµC', '', '=O=O=O']
    off, blob = OD.pack(rows)
    assert off == [0, 4, 4, 10] and blob == 'CµC=O=O=O'.encode()


def test_token_table_layout():
    from paccmann_generator_amd import ops
    itt = {0: '<PAD>', 1: 'C', 3: 'µ', 4: '[Branch1_1]'}
    t = ops.TokenTable(itt, 'cpu', drop=(0,))
    assert t.V == 5 and t.max_tok_bytes == 11 and not t.ascii
    assert t.tab_off.tolist() == [0, 0, 1, 1, 3, 14]
    assert bytes(t.tab_bytes.tolist()) == 'Cµ[Branch1_1]'.encode()
    t2 = ops.TokenTable(['a', 'bc'], 'cpu')
    assert t2.ascii and t2.tab_off.tolist() == [0, 1, 3]
