"""Host-side mirror of the reference's REINFORCE classes for the hot path only.

Same constructor arguments, method names, argument meaning and return values as
/root/reference/paccmann_generator/reinforce.py (``Reinforce``),
reinforce_omics.py (``ReinforceOmic``) and reinforce_proteins.py
(``ReinforceProtein``) for: ``get_smiles_from_latent``, ``encode_cell_line`` /
``encode_protein``, ``policy_gradient``, ``reparameterize``, ``save`` / ``load``.
What changed underneath: ``generator.generate`` is one fused HIP decode, and the
teacher-forced Python loop + ``rl_loss.backward()`` (reinforce_omics.py:319-335) is
one fused operator (ops.reinforce_loss).

Out of scope here: the RDKit / toxsmi drug
evaluators, the predictor networks, plotting.  The reward is therefore a pluggable
host callable ``reward_fn(smiles_list, target) -> np.ndarray`` and molecule validity a
pluggable ``is_valid(smiles) -> bool`` (defaults to RDKit's MolFromSmiles when RDKit
is importable, as at reinforce.py:195).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Optional

import numpy as np
import torch

from . import _ffi, dp, ops
from .decoder import SamplingSearch


def _rdkit_validity():
    try:
        from rdkit import Chem  # type: ignore
    except Exception:  # pragma: no cover - rdkit is absent in the build image
        return None
    return lambda s: Chem.MolFromSmiles(s, sanitize=True) is not None


class Reinforce(object):

    def __init__(self, generator, encoder, params, model_name, logger, is_valid: Optional[Callable] = None,
                 reward_fn: Optional[Callable] = None, process_group=None, score_fn: Optional[Callable] = None,
                 is_valid_batch: Optional[Callable] = None, micro_batches: Optional[int] = None):
        """Same positional arguments as reinforce.py:16.  Keyword-only extras:

        is_valid:      host validity check replacing Chem.MolFromSmiles (reinforce.py:195)
        is_valid_batch: same over the whole list ``f(list[str]) -> list[bool]`` (lets the caller fan
                       the RDKit work out over processes); takes precedence over is_valid
        reward_fn:     host reward ``f(smiles, target) -> array`` (reinforce_omics.py:209-222)
        score_fn:      host scoring used by generate_compounds_and_evaluate (defaults to reward_fn)
        process_group: torch.distributed group for data-parallel latent sharding
        micro_batches: k > 1 runs the reward scoring of a step under its teacher-forced pass and the
                       backward passes of earlier slices (_policy_gradient_pipelined); also read from
                       params['micro_batches']; default 1 (the reference's serial order)
        """
        super(Reinforce, self).__init__()
        self.generator = generator
        self.generator.eval()
        self.encoder = encoder
        if self.encoder is not None:
            self.encoder.eval()
        self.logger = logger
        self.device = self.generator.decoder.output_layer.weight.device
        if self.device.type != 'cuda':
            raise RuntimeError('paccmann_generator_amd needs the decoder on a GPU: there is no CPU path')

        # reinforce.py:44-49: Adam over decoder parameters only
        self.optimizer = torch.optim.Adam(
            list(self.generator.decoder.parameters()),
            lr=params.get('learning_rate', 0.0001),
            eps=params.get('eps', 0.0001),
            weight_decay=params.get('weight_decay', 0.00001),
        )
        self.model_name = model_name
        self.model_path = os.path.join(params.get('model_folder', 'biased_models'), model_name)
        self.weights_path = os.path.join(self.model_path, 'weights/{}')
        self.is_valid = is_valid if is_valid is not None else _rdkit_validity()
        self.is_valid_batch = is_valid_batch
        self.reward_fn = reward_fn
        self.score_fn = score_fn
        self.process_group = process_group
        self.micro_batches = max(1, int(micro_batches if micro_batches is not None else params.get('micro_batches', 1)))
        self._streams = []      # one per slice of a pipelined step

        # reinforce.py:59-77
        if params.get('save_untrained', True):
            if not os.path.isdir(self.model_path):
                os.makedirs(os.path.join(self.model_path, 'weights'), exist_ok=True)
                os.makedirs(os.path.join(self.model_path, 'results'), exist_ok=True)
                self.save('generator_epoch_0.pt', 'encoder_epoch_0.pt')
                with open(os.path.join(self.model_path, 'model_params.json'), 'w') as f:
                    json.dump(params, f)
            elif self.logger is not None:
                self.logger.warning('Model exists already. Call model.load() to restore weights.')
        self.update_params(params)

    # reinforce.py:79-97 (evaluator construction is out of scope)
    def update_params(self, params):
        self.gamma = params.get('gamma', 0.99)
        default_len = 100
        predictor = getattr(self, 'predictor', None)
        if predictor is not None and hasattr(predictor, 'params'):
            default_len = predictor.params.get('smiles_padding_length', 102) - 2
        self.generate_len = params.get('generate_len', default_len)
        self.temperature = params.get('temperature', 0.8)
        self.grad_clipping = params.get('clip_grad', None)

    def reparameterize(self, mu, logvar):
        """reinforce.py:99-111: noise-free."""
        return mu

    # ------------------------------------------------------------------ sampling
    def _decoder_latent(self, latent):
        """reinforce.py:160-161."""
        if self.encoder is not None and \
                self.generator.decoder.latent_dim == 2 * getattr(self.encoder, 'latent_size', -1):
            latent = latent.repeat(1, 1, 2)
        return latent

    # ------------------------------------------------------------------ data-parallel sharding
    def _dp_rank_world(self):
        if self.process_group is None:
            return 0, 1
        import torch.distributed as dist
        return dist.get_rank(self.process_group), dist.get_world_size(self.process_group)

    def _local_batch(self, batch_size: int) -> int:
        """Sequences of a GLOBAL batch of ``batch_size`` this rank decodes: the
        contiguous slice dp.shard_range gives it.  Its first global index becomes the Philox
        ``seq_index_base`` of the next decode, so N ranks sample exactly the rows a single process
        would have sampled.  Without a process group: the whole batch."""
        rank, world = self._dp_rank_world()
        lo, hi = dp.shard_range(batch_size, rank, world)
        self._seq_base = lo
        return hi - lo

    def sample_tokens(self, latent, seq_index_base: Optional[int] = None, early_exit: bool = True):
        """Device-side half of get_smiles_from_latent: (tokens [B,generate_len], lengths [B])."""
        if seq_index_base is None:
            seq_index_base = getattr(self, '_seq_base', 0)
        lang = self.generator.smiles_language
        return self.generator.generate_tokens(
            self._decoder_latent(latent),
            prime_input=torch.Tensor([lang.start_index]).long(),
            end_token=torch.Tensor([lang.stop_index]).long(),
            generate_len=self.generate_len,
            search=SamplingSearch(temperature=self.temperature),
            seq_index_base=seq_index_base, early_exit=early_exit)

    def _token_table(self, lang):
        """Device table for pmg_detokenize when the language exposes ``index_to_token`` (pytoda's
        SMILESLanguage does); None otherwise -> per-row ``token_indexes_to_smiles`` as the reference."""
        itt = getattr(lang, 'index_to_token', None)
        if itt is None:
            return None
        cached = getattr(self, '_tt_cache', None)
        if cached is None or cached[0] is not itt or cached[1] != len(itt):
            drop = getattr(lang, 'detokenize_drop', ())
            cached = (itt, len(itt), ops.TokenTable(itt, self.device, drop=drop))
            self._tt_cache = cached
        return cached[2]

    def get_smiles_from_latent(self, latent, remove_invalid=True):
        """reinforce.py:149-211.  Returns (smiles, nums, valid_idxs).

        With a language that exposes ``index_to_token`` the whole batch is detokenised on the device
        (ops.detokenize) and the ``+ 2 x stop`` rows (reinforce.py:177-184) are cut from one padded
        matrix built on the device; otherwise the per-row host loop of the reference runs."""
        lang = self.generator.smiles_language
        table = self._token_table(lang)
        if table is not None:
            tokens, lens = self.sample_tokens(latent)
            strings = ops.detokenize(table, tokens, lens)
            # rows + 2 x stop (reinforce.py:177-184) as a lazy sequence over one host copy of the token
            # matrix: B tensor objects cost more than the decode itself at B=4096, and the learner pads
            # the valid rows on the device anyway (_padded_valid)
            numericals = StopRows(tokens, lens, lang.stop_index)
        else:
            mols_numerical = self.generator.generate(
                self._decoder_latent(latent),
                prime_input=torch.Tensor([lang.start_index]).long(),
                end_token=torch.Tensor([lang.stop_index]).long(),
                generate_len=self.generate_len,
                search=SamplingSearch(temperature=self.temperature),
                seq_index_base=getattr(self, '_seq_base', 0),
            )
            stops = torch.tensor(2 * [lang.stop_index])
            rows = list(iter(mols_numerical))
            strings = [lang.token_indexes_to_smiles(r.tolist()) for r in rows]
            numericals = [torch.cat([r.long(), stops]) for r in rows]
        smiles = [lang.selfies_to_smiles(s) for s in strings]
        ok = self._validity(smiles)
        valid_idxs = [i for i in range(len(ok)) if ok[i]]
        keep = [i for i in range(len(ok)) if not (remove_invalid and not ok[i])]
        if self.logger is not None:
            self.logger.info(f'{self.model_name}: SMILES validity: {100.0 * sum(ok) / max(1, len(ok)):.2f}%.')
        nums = numericals.subset(keep) if isinstance(numericals, StopRows) else [numericals[i] for i in keep]
        return [smiles[i] for i in keep], nums, valid_idxs

    def _validity(self, smiles):
        """reinforce.py:195-196 over a list: one bool per molecule."""
        if self.is_valid_batch is not None:
            return [bool(v) for v in self.is_valid_batch(smiles)]
        if self.is_valid is not None:
            return [bool(self.is_valid(s)) for s in smiles]
        raise RuntimeError('no validity check available: pass is_valid= (RDKit is not importable)')

    # ------------------------------------------------------------------ learning
    def learn(self, latent_z, padded_nums, valid_idx, rewards, lrps: int = 1, trusted_tokens: bool = False):
        """reinforce_omics.py:301-342 given sampled + scored molecules.

        latent_z [1,B,latent_size] (autograd-connected to the encoder), padded_nums
        [T_max,B_valid] (pad_sequence layout), valid_idx indices into B, rewards [B_valid].
        Returns (mean_reward, loss) as 0-dim tensors.  ``trusted_tokens``: the rows were cut from
        pmg_decode's own output (ids in range by construction), skip the host-side id range check.

        With a process group these are this rank's shard and the step is data parallel: the kernel
        forms gradient SUMS (scale 1), ONE all-reduce carries [gradients | loss sum | reward sum |
        valid count] (dp.data_parallel_step), everything is divided by the global valid count afterwards,
        and clip + Adam run replicated.  Every rank takes part in that collective, also a rank with
        no valid molecule (it contributes zeros); the update is skipped only when no rank has one."""
        dec = self.generator.decoder
        dev = self.device
        rewards_t = torch.as_tensor(np.asarray(rewards), dtype=torch.float32) if not torch.is_tensor(rewards) \
            else rewards.to(torch.float32)
        rewards_t = rewards_t.reshape(-1)
        num_mols = padded_nums.shape[1] if padded_nums.dim() == 2 else 0
        ddp = self.process_group is not None
        enc_params = list(self.encoder.parameters()) if self.encoder is not None else []
        dec_params = list(dec.parameters())
        if not ddp and num_mols == 0:
            return torch.tensor(0.0), torch.tensor(0.0)

        def run(scale):
            dec._update_batch_size(num_mols)
            idx = valid_idx if torch.is_tensor(valid_idx) else torch.as_tensor(valid_idx, dtype=torch.long)
            hidden = dec.latent_to_hidden(
                latent_z.to(dev).repeat(dec.n_layers, 1, lrps)[:, idx.to(dev), :])
            loss, _ = ops.reinforce_loss(dec.weights(), padded_nums.to(dev), hidden, rewards_t.to(dev), scale,
                                         trusted_tokens=trusted_tokens)
            loss.backward()
            return loss.detach()

        if ddp:
            r, l, _ = dp.data_parallel_step(dec_params, enc_params, self.optimizer, lambda: run(1.0), num_mols,
                                            rewards_t.sum() if num_mols > 0 else 0.0, self.grad_clipping,
                                            group=self.process_group)
            return r, l
        self.optimizer.zero_grad()
        loss = run(1.0 / num_mols)
        summed_reward = torch.mean(rewards_t.float())
        if self.grad_clipping is not None:
            torch.nn.utils.clip_grad_norm_(dec_params + enc_params, self.grad_clipping)
        self.optimizer.step()
        return summed_reward, loss

    def _lrps(self):
        """reinforce_omics.py:310-312: latent repeats per sample."""
        if self.encoder is not None and \
                self.generator.decoder.latent_dim == 2 * getattr(self.encoder, 'latent_size', -1):
            return 2
        return 1

    def _policy_gradient(self, latent_z, target):
        B_local = latent_z.shape[1]
        if self.micro_batches > 1 and B_local > 0:
            return self._policy_gradient_pipelined(latent_z, target)
        ddp = self.process_group is not None
        try:
            if B_local > 0:
                valid_smiles, valid_nums, valid_idx = self.get_smiles_from_latent(latent_z, remove_invalid=True)
            else:   # empty shard (global batch smaller than the world): keep the Philox offset in step
                self.generator.skip_draws(self.generate_len)
                valid_smiles, valid_nums, valid_idx = [], [], []
        finally:
            self._seq_base = 0   # the shard's base holds for this step's decode only
        if len(valid_nums) == 0:
            if self.logger is not None:
                self.logger.warning('no valid molecule in the batch' +
                                    (' of this rank' if self.process_group is not None else ': update skipped'))
            if not ddp:
                return torch.tensor(0.0), 0.0
        # T_max of the step: the longest valid molecule (+ 2 stops) of the WHOLE batch -- over all ranks
        # when the batch is sharded, because the padded positions are not masked (reinforce_omics.py:304,
        # 322-332) and so every row's loss depends on it
        t_max = self._local_t_max(valid_nums, valid_idx)
        if ddp:
            t_max = dp.global_max(t_max, self.device, self.process_group)
        if len(valid_nums) == 0:
            rewards, padded_nums, trusted = np.zeros(0, dtype=np.float32), torch.zeros(0, 0, dtype=torch.long), True
        else:
            rewards = self.get_reward(valid_smiles, target)
            padded_nums, trusted = self._padded_valid(valid_nums, valid_idx, t_max)
        r, loss = self.learn(latent_z, padded_nums, valid_idx, rewards, self._lrps(), trusted_tokens=trusted)
        return r, _loss_value(loss)

    # ------------------------------------------------------------------ pipelined step
    def _policy_gradient_pipelined(self, latent_z, target):
        """One REINFORCE step (reinforce_omics.py:289-342) with the reward scoring of the host running
        UNDER the GPU work that does not depend on it.

        The reference is serial: decode (D) -> strings + validity (V, reinforce.py:172-211) -> reward
        (R, reinforce_omics.py:300) -> teacher-forced pass (T) -> backward (B: BPTT, W: weight
        gradients).  What can overlap is set by two facts.  (i) D -> V -> R -> B is a true dependency
        chain: validity needs the molecules, T_max (the longest VALID molecule, to which every row is
        padded without a mask, reinforce_omics.py:304,322-332) needs every validity, the backward pass
        needs the rewards.  (ii) The persistent kernels own 16 sequences per workgroup for all time
        steps, so their duration is (steps x step latency) whatever the batch: a slice of the batch is
        not faster than the whole, slices only run side by side on disjoint CUs.  Cutting D or V into
        slices therefore buys nothing (measured: profiles/r03_harness.json, first form).  What is off
        the chain is T -- it needs validity but not the rewards -- and, for all but the last slice, B
        and W.  So, with k = ``micro_batches`` slices of the VALID molecules:

          GPU  : D | idle (V) | T0..Tk-1 side by side | B0 W0 | B1 W1 | .. | Bk-1 Wk-1 |   clip, Adam
          host :   |    V     | R0      R1      ..        Rk-1 |
                                (slice i's backward is queued on stream i the moment R_i returns)

        Every slice has its own stream, workspace and gradient buffers; the streams join before the
        gradients are summed.  Same result as the serial step: same molecules (the decode is the serial
        one), same T_max, loss and gradients are sums over molecules with scale = 1 / valid molecules of
        the whole batch; the encoder and latent_to_hidden are differentiated once, through the
        concatenated hidden-state gradient of the slices."""
        dec, dev = self.generator.decoder, self.device
        ddp = self.process_group is not None
        try:
            valid_smiles, valid_nums, valid_idx = self.get_smiles_from_latent(latent_z, remove_invalid=True)
        finally:
            self._seq_base = 0
        n_valid = len(valid_nums)
        if n_valid == 0:
            if self.logger is not None:
                self.logger.warning('no valid molecule in the batch' + (' of this rank' if ddp else ': update skipped'))
            if not ddp:
                return torch.tensor(0.0), 0.0
        t_max = self._local_t_max(valid_nums, valid_idx)
        if ddp:
            t_max = dp.global_max(t_max, dev, self.process_group)
        dec_params = list(dec.parameters())
        enc_params = list(self.encoder.parameters()) if self.encoder is not None else []
        rewards_all = []

        def run(scale):
            padded, trusted = self._padded_valid(valid_nums, valid_idx, t_max)
            padded = padded.to(dev)
            if not trusted:
                ops.check_token_range(padded, dec.vocab_size)
            # slices: whole 16-sequence row groups, at most one slice per stream
            per = -(-(-(-n_valid // self.micro_batches)) // 16) * 16
            bounds = [(lo, min(n_valid, lo + per)) for lo in range(0, n_valid, per)]
            while len(self._streams) < len(bounds):
                self._streams.append(torch.cuda.Stream(device=dev))
            dec._update_batch_size(n_valid)
            idx = valid_idx if torch.is_tensor(valid_idx) else torch.as_tensor(valid_idx, dtype=torch.long)
            hidden = dec.latent_to_hidden(latent_z.to(dev).repeat(dec.n_layers, 1, self._lrps())[:, idx.to(dev), :])
            weights = dec.weights()
            main = torch.cuda.current_stream()
            ready = main.record_event()
            jobs = []
            for (lo, hi), st in zip(bounds, self._streams):      # T_i: needs validity and T_max, not the rewards
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    jobs.append(ops.TrainSlice(weights, padded[:, lo:hi], hidden.detach()[:, lo:hi]))
            for (lo, hi), st, job in zip(bounds, self._streams, jobs):
                r = np.asarray(self.get_reward(valid_smiles[lo:hi], target), dtype=np.float32).reshape(-1)
                if r.shape[0] != hi - lo:
                    raise RuntimeError('reward_fn must return one reward per molecule')
                rewards_all.append(r)
                r_t = torch.from_numpy(r).to(dev)       # on the (idle) main stream: never waits for T_i
                st.wait_event(main.record_event())
                with torch.cuda.stream(st):
                    job.backward(r_t, scale)
            for st in self._streams[:len(bounds)]:
                main.wait_stream(st)
            grads = jobs[0].grads
            for job in jobs[1:]:
                torch._foreach_add_(grads, job.grads)
            for p, g in zip(weights.flat(), grads):
                p.grad = g if p.grad is None else p.grad.add_(g)
            hidden.backward(torch.cat([job.grad_h0 for job in jobs], dim=1))    # latent_to_hidden, encoder
            loss = jobs[0].loss
            for job in jobs[1:]:
                loss = loss + job.loss
            return loss.reshape(())

        if ddp:
            r, loss, _ = dp.data_parallel_step(dec_params, enc_params, self.optimizer, lambda: run(1.0), n_valid,
                                               lambda: float(sum(float(x.sum(dtype=np.float64)) for x in rewards_all)),
                                               self.grad_clipping, group=self.process_group)
            return r, _loss_value(loss)
        self.optimizer.zero_grad()
        loss = run(1.0 / n_valid)
        r = torch.mean(torch.from_numpy(np.concatenate(rewards_all)).float())
        if self.grad_clipping is not None:
            torch.nn.utils.clip_grad_norm_(dec_params + enc_params, self.grad_clipping)
        self.optimizer.step()
        return r, _loss_value(loss)

    @staticmethod
    def _local_t_max(valid_nums, valid_idx) -> int:
        """Rows of pad_sequence(valid_nums): longest valid molecule + its two stop tokens; 0 without any."""
        if len(valid_nums) == 0:
            return 0
        if isinstance(valid_nums, StopRows):
            idx = torch.as_tensor(valid_nums.index, dtype=torch.long, device=valid_nums.lens.device)
            return int(valid_nums.lens.index_select(0, idx).max().item()) + 2
        return max(int(n.shape[0]) for n in valid_nums)

    def _padded_valid(self, valid_nums, valid_idx, t_max: Optional[int] = None):
        """pad_sequence(valid_nums) (reinforce_omics.py:304), zero-padded on to ``t_max`` rows when
        given.  When the rows are the ones the device detokeniser path just produced, the same
        [T_max, B_valid] matrix is built on the device from the decoded tokens instead of from B_valid
        host tensors.  Returns (padded, trusted): trusted rows were cut from pmg_decode's own output,
        their ids are in range by construction."""
        if isinstance(valid_nums, StopRows) and len(valid_idx) > 0 and valid_nums.index == list(valid_idx):
            idx = torch.as_tensor(valid_idx, dtype=torch.long, device=valid_nums.tokens.device)
            return pad_rows_on_device(valid_nums.tokens.index_select(0, idx), valid_nums.lens.index_select(0, idx),
                                      valid_nums.stop_index, t_max=t_max), True
        padded = torch.nn.utils.rnn.pad_sequence(list(valid_nums))
        if t_max is not None and padded.shape[0] < t_max:
            padded = torch.cat([padded, padded.new_zeros(t_max - padded.shape[0], padded.shape[1])])
        return padded, False

    def get_reward(self, valid_smiles, target):
        if self.reward_fn is None:
            raise RuntimeError('no reward function: pass reward_fn= (drug evaluators are out of scope)')
        return self.reward_fn(valid_smiles, target)

    def _evaluate(self, latent_z, target, return_latent, remove_invalid):
        """Sampling-only entry shared by generate_compounds_and_evaluate of the subclasses
        (reinforce_omics.py:164-180, reinforce_proteins.py:180-198): decode, then score the valid
        molecules with the host-side ``score_fn`` (the predictor forward is out of scope).

        With a process group the evaluation batch is sharded like a training batch: a rank decodes
        and scores its contiguous slice (global-row Philox counters, so the molecules are those of the
        unsharded call), then the (SMILES, prediction) lists are gathered in rank order and every rank
        returns the whole result.  Random latents (target None) are rank 0's, broadcast."""
        rank, world = self._dp_rank_world()
        score_fn = getattr(self, 'score_fn', None) or self.reward_fn
        if score_fn is None:
            raise RuntimeError('no scoring function: pass score_fn= or reward_fn=')
        local = latent_z
        self._seq_base = 0
        if world > 1:
            if target is None:
                dp.broadcast_(latent_z, src=0, group=self.process_group)
            lo, hi = dp.shard_range(latent_z.shape[1], rank, world)
            local = latent_z[:, lo:hi]
            self._seq_base = lo
        try:
            valid_smiles, _, _ = self.get_smiles_from_latent(local, remove_invalid=remove_invalid)
        finally:
            self._seq_base = 0
        preds = np.asarray(score_fn(valid_smiles, target)) if len(valid_smiles) else np.zeros(0)
        if world > 1:
            import torch.distributed as dist
            parts = [None] * world
            dist.all_gather_object(parts, (list(valid_smiles), preds.tolist()), group=self.process_group)
            valid_smiles = [s for p in parts for s in p[0]]
            preds = np.asarray([x for p in parts for x in p[1]])
        if return_latent:
            return valid_smiles, preds, latent_z
        return valid_smiles, preds

    # ------------------------------------------------------------------ checkpoints (reinforce.py:261-285)
    def load(self, generator_filepath=None, encoder_filepath=None, *args, **kwargs):
        if generator_filepath is not None:
            self.generator.load_model(self.weights_path.format(generator_filepath), *args, **kwargs)
        if encoder_filepath is not None and self.encoder is not None:
            self.encoder.load(self.weights_path.format(encoder_filepath), *args, **kwargs)

    def save(self, generator_filepath=None, encoder_filepath=None, *args, **kwargs):
        if generator_filepath is not None:
            self.generator.save(self.weights_path.format(generator_filepath), *args, **kwargs)
        if encoder_filepath is not None and self.encoder is not None and hasattr(self.encoder, 'save'):
            self.encoder.save(self.weights_path.format(encoder_filepath), *args, **kwargs)


class ReinforceOmic(Reinforce):
    """reinforce_omics.py:13.  ``gep_df``: pandas frame with columns ``cell_line`` and
    ``gene_expression`` (a Series of values per row), as at reinforce_omics.py:79-86."""

    def __init__(self, generator, encoder, predictor, gep_df, params, model_name, logger, **kw):
        self.predictor = predictor
        if predictor is not None and hasattr(predictor, 'eval'):
            predictor.eval()
        self.gep_df = gep_df
        super(ReinforceOmic, self).__init__(generator, encoder, params, model_name, logger, **kw)

    def _gep_tensor(self, cell_line):
        row = self.gep_df[self.gep_df['cell_line'] == cell_line].iloc[0]
        return torch.unsqueeze(torch.Tensor(np.asarray(row.gene_expression.values, dtype=np.float32)), 0)

    def encode_cell_line(self, cell_line=None, batch_size=128):
        """reinforce_omics.py:75-95."""
        gep_t = self._gep_tensor(cell_line).to(self.device)
        cell_mu, cell_logvar = self.encoder(gep_t)
        return torch.unsqueeze(
            self.reparameterize(cell_mu.repeat(batch_size, 1), cell_logvar.repeat(batch_size, 1)), 0)

    def policy_gradient(self, cell_line, epoch, batch_size=10):
        """reinforce_omics.py:276-342.  Returns (mean reward, loss float)."""
        latent_z = self.encode_cell_line(cell_line, self._local_batch(batch_size))
        return self._policy_gradient(latent_z, cell_line)

    def generate_compounds_and_evaluate(self, epoch, batch_size, cell_line=None, primed_drug=' ',
                                        return_latent=False, remove_invalid=True):
        """reinforce_omics.py:97-180: random latents when cell_line is None (:122-126), else the
        encoder mean repeated batch_size times (:128-145); decode; score."""
        if type(primed_drug) != str:
            raise TypeError('Provide drug as SMILES string.')
        if cell_line is None:
            latent_z = torch.randn(1, batch_size, self.generator.decoder.latent_dim, device=self.device)
        else:
            with torch.no_grad():
                latent_z = self.encode_cell_line(cell_line, batch_size)
        return self._evaluate(latent_z, cell_line, return_latent, remove_invalid)


class ReinforceProtein(Reinforce):
    """reinforce_proteins.py.  ``protein_df``: frame indexed by protein name whose columns
    ``'0'..'767'`` hold the encoder input vector (reinforce_proteins.py:85-131 with
    encoder_uses_sequence=False); a frame without those column names is taken whole."""

    def __init__(self, generator, encoder, predictor, protein_df, params, model_name, logger, **kw):
        self.predictor = predictor
        if predictor is not None and hasattr(predictor, 'eval'):
            predictor.eval()
        self.protein_df = protein_df
        super(ReinforceProtein, self).__init__(generator, encoder, params, model_name, logger, **kw)

    def encode_protein(self, protein=None, batch_size=128):
        """reinforce_proteins.py:60-83."""
        if protein is None:
            return torch.randn(1, batch_size, self.encoder.latent_size, device=self.device)
        row = self.protein_df.loc[protein]
        locations = [str(x) for x in range(768)]          # reinforce_proteins.py:124-128: embedding columns
        if all(c in row.index for c in (locations[0], locations[-1])):
            row = row[locations]                          # frames that also carry e.g. a 'Sequence' column
        vec = np.asarray(row.values, dtype=np.float32)
        protein_tensor = torch.unsqueeze(torch.Tensor(vec), 0).to(self.device)
        protein_mu, protein_logvar = self.encoder(protein_tensor)
        return torch.unsqueeze(
            self.reparameterize(protein_mu.repeat(batch_size, 1), protein_logvar.repeat(batch_size, 1)), 0)

    def policy_gradient(self, protein, epoch, batch_size=10):
        """reinforce_proteins.py:267-331."""
        latent_z = self.encode_protein(protein, self._local_batch(batch_size))
        return self._policy_gradient(latent_z, protein)

    def generate_compounds_and_evaluate(self, epoch, batch_size, protein=None, primed_drug=' ',
                                        return_latent=False, remove_invalid=True):
        """reinforce_proteins.py:133-198."""
        if type(primed_drug) != str:
            raise TypeError('Provide drug as SMILES string.')
        with torch.no_grad():
            latent_z = self.encode_protein(protein, batch_size)
        return self._evaluate(latent_z, protein, return_latent, remove_invalid)


class ReinforceMultiModalSets(Reinforce):
    """Hot path of reinforce_sets.py: the same sampling + teacher-forced REINFORCE loop
    (reinforce_sets.py:438-502) conditioned on a set of embeddings.  What differs from the omics
    / protein classes is in-tree and reproduced: the latent is the mean of the tensors the set
    encoder returns (:464), the reward takes both targets, and there is NO gradient clipping
    (:499-501).  Building the input set (reinforce_sets.py get_set: predictor embeddings of the
    cell line and the protein) is host work outside the path: pass ``set_fn(cell_line, protein,
    batch_size) -> input_set``."""

    def __init__(self, generator, encoder, params, model_name, logger, set_fn=None, **kw):
        self.set_fn = set_fn
        super(ReinforceMultiModalSets, self).__init__(generator, encoder, params, model_name, logger, **kw)
        self.grad_clipping = None

    def update_params(self, params):
        super(ReinforceMultiModalSets, self).update_params(params)
        self.grad_clipping = None

    def encode_set(self, cell_line, protein, batch_size):
        if self.set_fn is None:
            raise RuntimeError('no set builder: pass set_fn=')
        input_set = self.set_fn(cell_line, protein, batch_size)
        encoder_output = self.encoder(input_set.to(self.device) if torch.is_tensor(input_set) else input_set)
        return torch.mean(torch.stack(list(encoder_output)), dim=0)

    def get_reward(self, valid_smiles, target):
        protein, cell_line = target
        if self.reward_fn is None:
            raise RuntimeError('no reward function: pass reward_fn=')
        return self.reward_fn(valid_smiles, protein, cell_line)

    def policy_gradient(self, cell_line, protein, epoch, batch_size=10):
        """reinforce_sets.py:438-502."""
        latent_z = self.encode_set(cell_line, protein, self._local_batch(batch_size))
        return self._policy_gradient(latent_z, (protein, cell_line))


class StopRows:
    """The list of 1-D token tensors ``row[:len] + [stop, stop]`` the reference builds at
    reinforce.py:177-184, as a lazy sequence over the decoded [B, generate_len] matrix: rows are cut
    (from one host copy, made on first use) only when somebody indexes or iterates.  Keeps the device
    tensors, so that the learner can pad the selected rows without a host round trip."""

    def __init__(self, tokens: torch.Tensor, lens: torch.Tensor, stop_index: int, index=None):
        self.tokens, self.lens, self.stop_index = tokens, lens, int(stop_index)
        self.index = list(range(tokens.shape[0])) if index is None else list(index)
        self._host = None

    def subset(self, keep):
        out = StopRows(self.tokens, self.lens, self.stop_index, [self.index[i] for i in keep])
        out._host = self._host
        return out

    def _rows(self):
        if self._host is None:
            lens = self.lens.cpu().tolist()
            if lens and min(lens) < 0:
                raise _ffi.PmgError('pmg_decode gave up on a cluster barrier (lengths poisoned with -1): another kernel '
                                    'held the CUs; pass PMG_FLAG_NO_CLUSTER on launches that overlap (include/pmg.h)')
            self._host = (self.tokens.cpu(), lens, torch.tensor(2 * [self.stop_index]))
        return self._host

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.subset(range(len(self.index))[i])
        tok, lens, stops = self._rows()
        r = self.index[i]
        return torch.cat([tok[r, :lens[r]], stops])

    def __iter__(self):
        return (self[i] for i in range(len(self.index)))


def _loss_value(loss) -> float:
    """The step's loss on the host.  NaN is reported loudly: a clustered launch that gave up on a barrier poisons
    its outputs with NaN (include/pmg.h, PMG_FLAG_NO_CLUSTER) and the update that followed is not to be trusted;
    a policy that diverged by itself deserves the same message."""
    v = float(loss.detach()) if isinstance(loss, torch.Tensor) else float(loss)
    if v != v:
        raise _ffi.PmgError('pmg: the REINFORCE loss is NaN -- a clustered launch gave up on a cluster barrier '
                            '(another kernel held the CUs: see PMG_FLAG_NO_CLUSTER in include/pmg.h) or the '
                            'policy diverged; the parameters were updated with it')
    return v


def pad_rows_on_device(tokens: torch.Tensor, lens: torch.Tensor, stop_index: int, t_max: Optional[int] = None):
    """Device twin of reinforce.py:177-184 + pad_sequence (reinforce_omics.py:304): every row
    becomes tokens[:len] + [stop, stop], zero-padded, seq-major [T_max, B]."""
    B, N = tokens.shape
    L = lens.to(torch.int64).unsqueeze(0)
    if t_max is None:
        t_max = int(L.max().item()) + 2 if B > 0 else 2
    pos = torch.arange(t_max, device=tokens.device).unsqueeze(1)  # [T,1]
    body = tokens.t()
    # seven small launches where the obvious form (zeros + copy + full_like + zeros_like + two where) took fourteen:
    # at a small batch each one is 5 us of a 2.6 ms step
    body = body[:t_max] if t_max <= N else torch.nn.functional.pad(body, (0, 0, 0, t_max - N))
    return torch.where(pos < L, body, torch.where(pos < L + 2, int(stop_index), 0))
