"""Eager-PyTorch restatement of the StackGRU decoder the reference drives.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED.

What is restated, and from where:

* The decoder *protocol* is fixed by the reference's call sites:
  ``decoder(tok[1,B], hidden[L,B,H], stack) -> (out[1,B,H], hidden, stack)``,
  ``decoder.latent_to_hidden``, ``decoder.output_layer``, ``decoder.init_stack``,
  ``decoder._update_batch_size``, ``decoder.latent_dim``, ``decoder.n_layers``
  (/root/reference/paccmann_generator/reinforce_omics.py:307-326,
  /root/reference/paccmann_generator/reinforce.py:160-170).
* The GRU gate arithmetic is torch's ``nn.GRU`` (gate order r,z,n; weight layout
  ``weight_ih_l{k} [3H,in]``, ``weight_hh_l{k} [3H,H]``).
* The stack arithmetic is the published algorithm of paccmann_chemistry's
  ``StackGRU`` (v0.0.4, ``models/stack_rnn.py``: ``_stack_update`` +
  ``stack_augmentation``), itself the stack-augmented RNN of Joulin & Mikolov
  (2015) as used by ReLeaSE: controls = softmax(Linear_{H->3}(h[-1])) read as
  (push, pop, no-op); push value = tanh(Linear_{H->W}(h[-1]));
  ``new = no_op*prev + push*[val; prev[:-1]] + pop*[prev[1:]; 0]``; the GRU input
  is ``concat(embedding(tok), new[:,0,:])``.  That source is NOT in this container,
  so this is a restatement of its published behaviour, not a checked copy.
* ``generate`` follows ``TeacherVAE.generate`` / ``generate_from_latent`` of the
  same package as far as the reference's call site constrains it
  (reinforce.py:162-187): h0 = latent_to_hidden(latent.repeat(L,1,1)), zero stack,
  feed the start token, sample ``generate_len`` tokens, strip the start token and
  cut each row before its first end token.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


DEFAULT_PARAMS = dict(
    embedding_size=128,
    rnn_cell_size=512,
    vocab_size=60,
    stack_width=50,
    stack_depth=50,
    n_layers=1,
    latent_dim=256,
    pad_index=0,
    dropout=0.0,
    batch_size=1,
)


class OracleStackGRUDecoder(nn.Module):
    """StackGRUDecoder restated (parameter names follow paccmann_chemistry)."""

    def __init__(self, params: dict, dtype=torch.float32):
        super().__init__()
        p = dict(DEFAULT_PARAMS)
        p.update(params)
        self.params = p
        self.embedding_size = p['embedding_size']
        self.rnn_cell_size = p['rnn_cell_size']
        self.vocab_size = p['vocab_size']
        self.stack_width = p['stack_width']
        self.stack_depth = p['stack_depth']
        self.n_layers = p['n_layers']
        self.latent_dim = p['latent_dim']
        self.batch_size = p['batch_size']
        self.pad_index = p.get('pad_index', 0)
        self.use_stack = True
        pad = self.pad_index if self.pad_index is not None and self.pad_index >= 0 else None

        self.stack_controls_layer = nn.Linear(self.rnn_cell_size, 3)
        self.stack_input_layer = nn.Linear(self.rnn_cell_size, self.stack_width)
        self.embedding = nn.Embedding(self.vocab_size, self.embedding_size, padding_idx=pad)
        self.gru = nn.GRU(
            self.embedding_size + self.stack_width,
            self.rnn_cell_size,
            self.n_layers,
            bidirectional=False,
            dropout=0.0,
        )
        self.latent_to_hidden = nn.Linear(self.latent_dim, self.rnn_cell_size)
        self.output_layer = nn.Linear(self.rnn_cell_size, self.vocab_size)
        self.to(dtype)

    # -- protocol members the harness touches ---------------------------------
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

    # -- arithmetic -------------------------------------------------------------
    def stack_augmentation(self, input_val, prev_stack, controls):
        """input_val [B,1,W]; prev_stack [B,D,W]; controls [B,3] (push,pop,no-op)."""
        batch = prev_stack.size(0)
        controls = controls.view(-1, 3, 1, 1)
        zeros_at_the_bottom = torch.zeros(batch, 1, self.stack_width, dtype=prev_stack.dtype)
        a_push, a_pop, a_no_op = controls[:, 0], controls[:, 1], controls[:, 2]
        stack_down = torch.cat((prev_stack[:, 1:], zeros_at_the_bottom), dim=1)
        stack_up = torch.cat((input_val, prev_stack[:, :-1]), dim=1)
        return a_no_op * prev_stack + a_push * stack_up + a_pop * stack_down

    def _stack_update(self, embedded_input, hidden, stack):
        stack_controls = F.softmax(self.stack_controls_layer(hidden[-1, :, :]), dim=-1)
        stack_input = torch.tanh(self.stack_input_layer(hidden[-1, :, :].unsqueeze(0)))
        stack = self.stack_augmentation(stack_input.permute(1, 0, 2), stack, stack_controls)
        stack_top = stack[:, 0, :].unsqueeze(0)
        return torch.cat((embedded_input, stack_top), dim=2), stack

    def forward(self, input_token, hidden, stack):
        """input_token [1,B] int64; hidden [L,B,H]; stack [B,D,W]."""
        embedded_input = self.embedding(input_token)
        gru_input, stack = self._stack_update(embedded_input, hidden, stack)
        output, hidden = self.gru(gru_input, hidden)
        return output, hidden, stack


class GreedySearch:
    def step(self, logits):
        return torch.argmax(logits, dim=1, keepdim=True)


class SamplingSearch:
    """softmax(logits / T) then one multinomial draw per row (paccmann_chemistry
    ``utils/search.py`` SamplingSearch, call site reinforce.py:169)."""

    def __init__(self, temperature: float = 1.0):
        self.temperature = temperature

    def step(self, logits):
        probabilities = torch.softmax(logits.div(self.temperature), 1)
        return torch.stack([torch.multinomial(p, 1) for p in probabilities])


class OracleGenerator(nn.Module):
    """The part of TeacherVAE the harness touches: ``.decoder`` and ``.generate``."""

    def __init__(self, decoder: OracleStackGRUDecoder):
        super().__init__()
        self.decoder = decoder

    @torch.no_grad()
    def generate_raw(self, latent_z, prime_input, end_token, generate_len, search):
        """Returns the full [B, generate_len] token matrix (no cropping)."""
        dec = self.decoder
        latent_z = latent_z.repeat(dec.n_layers, 1, 1)
        hidden = dec.latent_to_hidden(latent_z)
        batch = hidden.shape[1]
        dec._update_batch_size(batch)
        stack = dec.init_stack
        input_token = prime_input.view(1, 1).repeat(1, batch)
        toks = []
        for _ in range(generate_len):
            output, hidden, stack = dec(input_token, hidden, stack)
            logits = dec.output_layer(output).squeeze(dim=0)
            top_idx = search.step(logits)
            input_token = top_idx.view(1, -1)
            toks.append(top_idx.view(-1))
        return torch.stack(toks, dim=1)

    def generate(self, latent_z, prime_input, end_token, generate_len=100, search=None):
        search = search if search is not None else SamplingSearch()
        raw = self.generate_raw(latent_z, prime_input, end_token, generate_len, search)
        end = int(end_token.view(-1)[0])
        out = []
        for row in raw:
            r = row.tolist()
            n = r.index(end) if end in r else len(r)
            out.append(torch.tensor(r[:n], dtype=torch.long))
        return iter(out)


def build_decoder(params: dict, seed: int = 0, dtype=torch.float32) -> OracleStackGRUDecoder:
    """Random-init decoder: torch.manual_seed(seed), default torch init."""
    torch.manual_seed(seed)
    dec = OracleStackGRUDecoder(params, dtype=torch.float32)
    dec.eval()
    if dtype != torch.float32:
        dec = dec.to(dtype)
    return dec
