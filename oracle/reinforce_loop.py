"""Restatement of the reference's teacher-forced REINFORCE loop.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/paccmann_generator/reinforce_omics.py:276-342 (protein twin:
reinforce_proteins.py:267-331; sets twin: reinforce_sets.py:438-502) line by line:

* ``padded_nums = pad_sequence(valid_nums)`` -> [T_max, B_valid], pad value 0 (:304)
* ``hidden = latent_to_hidden(latent_z.repeat(n_layers, 1, lrps)[:, valid_idx, :])`` (:310-317)
* ``stack = decoder.init_stack`` after ``_update_batch_size(num_mols)`` (:307,:319)
* for p in range(T_max-1): one decoder step on ``padded_nums[p]``, ``output_layer``,
  ``log_softmax(dim=1)``, ``gather`` of ``padded_nums[p+1]``, ``rl_loss -= mean(lp * reward)``
  with NO padding mask, NO temperature and NO discount (:322-332)
* ``rl_loss.backward()``, optional ``clip_grad_norm_(decoder+encoder params)``,
  ``optimizer.step()`` (:335-341)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def append_stops(mol_nums, stop_index: int):
    """reinforce.py:177-184: every sampled row gets 2 x [stop_index] appended."""
    return [torch.cat([m.long(), torch.tensor(2 * [stop_index])]) for m in mol_nums]


def pad_tokens(valid_nums):
    """reinforce_omics.py:304."""
    return torch.nn.utils.rnn.pad_sequence(valid_nums)


def initial_hidden(decoder, latent_z, valid_idx, lrps: int):
    """reinforce_omics.py:310-317.  latent_z [1,B,latent_size]."""
    return decoder.latent_to_hidden(latent_z.repeat(decoder.n_layers, 1, lrps)[:, valid_idx, :])


def teacher_forced_logprobs(decoder, padded_nums, hidden):
    """Per-position log-prob of the next token, [T_max-1, B] (reinforce_omics.py:322-329)."""
    decoder._update_batch_size(padded_nums.shape[1])
    stack = decoder.init_stack
    out = []
    for p in range(len(padded_nums) - 1):
        output, hidden, stack = decoder(torch.unsqueeze(padded_nums[p], 0), hidden, stack)
        output = decoder.output_layer(output).squeeze(0)
        log_probs = F.log_softmax(output, dim=1)
        target_char = torch.unsqueeze(padded_nums[p + 1], 1)
        out.append(log_probs.gather(1, target_char).squeeze(1))
    return torch.stack(out, dim=0)


def reinforce_loss(decoder, padded_nums, hidden, reward_tensor):
    """rl_loss exactly as accumulated at reinforce_omics.py:322-332.

    reward_tensor: [B,1] (reinforce_omics.py:301)."""
    decoder._update_batch_size(padded_nums.shape[1])
    stack = decoder.init_stack
    rl_loss = 0
    for p in range(len(padded_nums) - 1):
        output, hidden, stack = decoder(torch.unsqueeze(padded_nums[p], 0), hidden, stack)
        output = decoder.output_layer(output).squeeze(0)
        log_probs = F.log_softmax(output, dim=1)
        target_char = torch.unsqueeze(padded_nums[p + 1], 1)
        rl_loss = rl_loss - torch.mean(log_probs.gather(1, target_char) * reward_tensor)
    return rl_loss


def policy_gradient_step(decoder, optimizer, latent_z, valid_nums, valid_idx, rewards,
                         lrps: int = 1, clip_grad=None, extra_clip_params=()):
    """One learner update given already-sampled, already-scored molecules.

    Mirrors reinforce_omics.py:289,301-342 (everything except sampling and the
    host-side reward, which the caller supplies).  Returns (mean_reward, loss_float)."""
    optimizer.zero_grad()
    reward_tensor = torch.unsqueeze(torch.as_tensor(rewards, dtype=latent_z.dtype), 1)
    padded_nums = pad_tokens(valid_nums)
    hidden = initial_hidden(decoder, latent_z, valid_idx, lrps)
    rl_loss = reinforce_loss(decoder, padded_nums, hidden, reward_tensor)
    summed_reward = torch.mean(torch.as_tensor(rewards, dtype=torch.float32))
    rl_loss.backward()
    if clip_grad is not None:
        torch.nn.utils.clip_grad_norm_(list(decoder.parameters()) + list(extra_clip_params), clip_grad)
    optimizer.step()
    return summed_reward, rl_loss.item()


def make_optimizer(decoder, params: dict):
    """reinforce.py:44-49: Adam over decoder parameters only."""
    return torch.optim.Adam(
        list(decoder.parameters()),
        lr=params.get('learning_rate', 0.0001),
        eps=params.get('eps', 0.0001),
        weight_decay=params.get('weight_decay', 0.00001),
    )
