"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the detokenisation
step of /root/reference/paccmann_generator/reinforce.py:172-176:

    smiles_language.token_indexes_to_smiles(mol_num.tolist())

`smiles_language` is a pytoda SMILESLanguage (third party, pinned by requirements.txt:2, absent
from /root/reference and from this image), whose published behaviour is the concatenation of
`index_to_token.get(index, '')` over the indexes.  parity unpinned: the reference holds no fixture
for it; the product kernel pmg_detokenize is checked against this restatement byte for byte."""


def token_indexes_to_text(index_to_token, token_indexes, drop=()):
    table = dict(enumerate(index_to_token)) if isinstance(index_to_token, (list, tuple)) else index_to_token
    return ''.join('' if i in drop else table.get(int(i), '') for i in token_indexes)


def detokenize_rows(index_to_token, tokens, lens, drop=()):
    """tokens [B][N] ints, lens [B]: row b -> text of tokens[b][:clamp(lens[b], 0, N)]."""
    out = []
    for row, n in zip(tokens, lens):
        n = max(0, min(int(n), len(row)))
        out.append(token_indexes_to_text(index_to_token, list(row[:n]), drop))
    return out


def pack(strings):
    """The packed layout pmg_detokenize produces: (offsets [B+1], blob bytes)."""
    off = [0]
    blob = bytearray()
    for s in strings:
        blob += s.encode('utf-8')
        off.append(len(blob))
    return off, bytes(blob)
