"""Drop-in for the decoder/generator protocol the reference harness drives.

The reference touches exactly these members (SURVEY.md section 8b;
/root/reference/paccmann_generator/reinforce.py:36,45,160-193,266,279 and
reinforce_omics.py:307-326): ``generator.eval()``, ``generator.decoder``,
``generator.smiles_language``, ``generator.generate(...)``, ``generator.save/load/
load_model/_associate_language``; ``decoder.parameters()``, ``decoder.latent_dim``,
``decoder.n_layers``, ``decoder._update_batch_size``, ``decoder.init_stack``,
``decoder.latent_to_hidden``, ``decoder.output_layer`` and
``decoder(tok, hidden, stack)``.

Parameter names follow paccmann_chemistry's StackGRUDecoder (``embedding``, ``gru``,
``stack_controls_layer``, ``stack_input_layer``, ``latent_to_hidden``,
``output_layer``) so a ``state_dict`` of that class loads unchanged.  All arithmetic
of ``forward`` / ``generate`` runs in the HIP library (ops.py); the torch modules here
only own the parameter storage.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

DEFAULTS = dict(embedding_size=128, rnn_cell_size=512, vocab_size=60, stack_width=50, stack_depth=50,
                n_layers=1, latent_dim=256, pad_index=0, dropout=0.0, batch_size=1)


class StackGRUDecoder(nn.Module):
    def __init__(self, params: dict):
        super().__init__()
        p = dict(DEFAULTS)
        p.update(params)
        if p.get('bidirectional', False):
            raise ValueError('bidirectional decoders are not supported')
        if not p.get('use_stack', True):
            raise ValueError('use_stack=False is not supported')
        self.params = p
        self.embedding_size = p['embedding_size']
        self.rnn_cell_size = p['rnn_cell_size']
        self.vocab_size = p['vocab_size']
        self.stack_width = p['stack_width']
        self.stack_depth = p['stack_depth']
        self.n_layers = p['n_layers']
        self.latent_dim = p['latent_dim']
        self.batch_size = p['batch_size']
        pad = p.get('pad_index', 0)
        pad = pad if (pad is not None and pad >= 0) else None
        self.stack_controls_layer = nn.Linear(self.rnn_cell_size, 3)
        self.stack_input_layer = nn.Linear(self.rnn_cell_size, self.stack_width)
        self.embedding = nn.Embedding(self.vocab_size, self.embedding_size, padding_idx=pad)
        # parameter container only: the recurrence runs in the HIP kernels
        self.gru = nn.GRU(self.embedding_size + self.stack_width, self.rnn_cell_size, self.n_layers,
                          bidirectional=False, dropout=0.0)
        self.latent_to_hidden = nn.Linear(self.latent_dim, self.rnn_cell_size)
        self.output_layer = nn.Linear(self.rnn_cell_size, self.vocab_size)

    # -- protocol ---------------------------------------------------------------
    def _update_batch_size(self, batch_size: int) -> None:
        self.batch_size = int(batch_size)

    @property
    def init_stack(self) -> torch.Tensor:
        w = self.output_layer.weight
        return torch.zeros(self.batch_size, self.stack_depth, self.stack_width, dtype=w.dtype, device=w.device)

    @property
    def init_hidden(self) -> torch.Tensor:
        w = self.output_layer.weight
        return torch.zeros(self.n_layers, self.batch_size, self.rnn_cell_size, dtype=w.dtype, device=w.device)

    def weights(self) -> ops.DecoderWeights:
        return ops.DecoderWeights(self)

    @torch.no_grad()
    def forward(self, input_token, hidden, stack):
        """One step: ``(output[1,B,H], hidden[L,B,H], stack[B,D,W])`` (reinforce_omics.py:323-325).

        Inference only.  The differentiable path of the reference's Python loop is the
        fused operator ``ops.reinforce_loss``."""
        h_out, s_out, _ = ops.step(self.weights(), input_token, hidden, stack, want_logits=False)
        return h_out[-1:].clone(), h_out, s_out


class _CroppedRows:
    """Iterable of variable-length 1-D token tensors (what TeacherVAE.generate yields,
    consumed at reinforce.py:172-186)."""

    def __init__(self, tokens_cpu: torch.Tensor, lens_cpu: torch.Tensor):
        self.tokens = tokens_cpu
        self.lens = lens_cpu
        if lens_cpu.numel() and int(lens_cpu.min()) < 0:
            from . import _ffi
            raise _ffi.PmgError('pmg_decode gave up on a cluster barrier (lengths poisoned with -1): another kernel '
                                'held the CUs; pass PMG_FLAG_NO_CLUSTER on launches that overlap (include/pmg.h)')

    def __iter__(self):
        for row, n in zip(self.tokens, self.lens.tolist()):
            yield row[:n].clone()

    def __len__(self):
        return self.tokens.shape[0]


class Generator(nn.Module):
    """The part of paccmann_chemistry's TeacherVAE the REINFORCE harness uses."""

    def __init__(self, encoder, decoder: StackGRUDecoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.smiles_language = None
        self._philox_seed = 0
        self._philox_offset = 0

    def _associate_language(self, language) -> None:
        self.smiles_language = language

    def seed(self, seed: int) -> None:
        self._philox_seed = int(seed)
        self._philox_offset = 0

    def generate_tokens(self, latent_z, prime_input, end_token, generate_len=100, search=None,
                        seq_index_base: int = 0, early_exit: bool = True):
        """Device-side result: (tokens [B,generate_len], lengths [B])."""
        temperature = getattr(search, 'temperature', 1.0) if search is not None else 1.0
        if search is not None and getattr(search, 'greedy', False):
            temperature = 0.0
        start = int(torch.as_tensor(prime_input).reshape(-1)[-1])
        if torch.as_tensor(prime_input).numel() != 1:
            raise ValueError('priming with more than the start token is not supported')
        stop = int(torch.as_tensor(end_token).reshape(-1)[0])
        dev = self.decoder.output_layer.weight.device
        latent = latent_z.to(dev)
        if latent.shape[-2] == 0:      # empty shard of a data-parallel batch: nothing to decode
            self.skip_draws(generate_len)
            return (torch.zeros(0, int(generate_len), dtype=torch.int64, device=dev),
                    torch.zeros(0, dtype=torch.int32, device=dev))
        tokens, lens = ops.decode(self.decoder.weights(), latent, start, stop, int(generate_len),
                                  temperature=float(temperature), seed=self._philox_seed,
                                  offset=self._philox_offset, seq_index_base=seq_index_base,
                                  early_exit=early_exit)
        self._philox_offset += int(generate_len)
        return tokens, lens

    def skip_draws(self, generate_len: int) -> None:
        """Advance the Philox offset as one generate call would (a data-parallel rank with an empty
        shard stays in step with the others)."""
        self._philox_offset += int(generate_len)

    def generate(self, latent_z, prime_input, end_token, generate_len=100, search=None, seq_index_base: int = 0):
        """Same call and result as at reinforce.py:162-170: an iterable of 1-D token
        tensors, start token stripped, cut before the first end token.  ``seq_index_base`` (not in
        the reference): global index of the first row when the batch is a shard of a larger one."""
        tokens, lens = self.generate_tokens(latent_z, prime_input, end_token, generate_len, search,
                                            seq_index_base=seq_index_base)
        return _CroppedRows(tokens.cpu(), lens.cpu())

    # -- checkpoints: plain state_dict .pt (format unchanged) ---------------------
    def save(self, path, *args, **kwargs):
        torch.save(self.state_dict(), path, *args, **kwargs)

    def load(self, path, *args, **kwargs):
        """Plain ``state_dict`` checkpoint, as TeacherVAE.save writes it (keys ``encoder.*`` and
        ``decoder.*``).  When this generator was built without the SMILES encoder (the REINFORCE loop
        never calls it) the ``encoder.*`` entries of the file are skipped; every ``decoder.*`` entry
        must still match."""
        kwargs.setdefault('weights_only', True)
        state = torch.load(path, *args, **kwargs)
        if self.encoder is None:
            state = {k: v for k, v in state.items() if not k.startswith('encoder.')}
        self.load_state_dict(state)

    load_model = load


class SamplingSearch:
    """paccmann_chemistry.utils.search.SamplingSearch as constructed at reinforce.py:169."""

    def __init__(self, temperature: float = 1.0):
        self.temperature = float(temperature)
        self.greedy = False


class GreedySearch:
    def __init__(self):
        self.temperature = 0.0
        self.greedy = True
