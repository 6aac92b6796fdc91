"""Training-loop mirror of the reference drivers (hot-path part only).

/root/reference/examples/IC50/train_paccmann_rl.py:185-274 and
examples/affinity/train_conditional_generator.py:232-320: ``epochs x range(1, steps)`` calls of
``learner.policy_gradient(target, epoch, batch_size)`` on a randomly chosen training target, a
checkpoint per epoch (``gen_{epoch}.pt`` / ``enc_{epoch}.pt`` through ``learner.save``), an
evaluation pass on the last training target and on a random held-out target, and the two CSV
files ``results/generated.csv`` and ``results/loss_reward_evolution.csv``.

Out of scope: the baseline comparison plots, tox21 column and
argparse/JSON handling of the reference scripts.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np


def train(learner, train_targets: Sequence, test_targets: Sequence, params: dict, keep: Optional[Callable] = None,
          rng: Optional[np.random.RandomState] = None, log: Callable = print, save_every: int = 1):
    """Runs the loop and returns ``(rewards, rl_losses, generated_rows)``.

    keep(pred) -> bool selects the molecules written to generated.csv (the reference keeps
    ``pred < learner.ic50_threshold``, train_paccmann_rl.py:213-216); default keeps all."""
    import pandas as pd
    rng = rng if rng is not None else np.random
    rewards, rl_losses, rows = [], [], []
    results = os.path.join(learner.model_path, 'results')
    os.makedirs(results, exist_ok=True)
    os.makedirs(os.path.join(learner.model_path, 'weights'), exist_ok=True)
    keep = keep if keep is not None else (lambda p: True)
    target = None
    for epoch in range(1, params['epochs'] + 1):
        for step in range(1, params['steps']):
            target = train_targets[rng.randint(len(train_targets))]
            rew, loss = learner.policy_gradient(target, epoch, params['batch_size'])
            log(f"Epoch {epoch:d}/{params['epochs']:d}, step {step:d}/{params['steps']:d}\t "
                f"loss={loss:.2f}, rew={float(rew):.2f}")
            rewards.append(float(rew))
            rl_losses.append(float(loss))
        if epoch % save_every == 0:
            learner.save(f'gen_{epoch}.pt', f'enc_{epoch}.pt')
        for mode, tgt in (('train', target), ('test', test_targets[rng.randint(len(test_targets))] if len(test_targets)
                                              else None)):
            if tgt is None:
                continue
            smiles, preds = learner.generate_compounds_and_evaluate(epoch, params.get('eval_batch_size', 32), tgt)
            for s, p in zip(smiles, np.atleast_1d(preds)):
                if keep(p):
                    rows.append({'target': tgt, 'SMILES': s, 'pred': float(p), 'mode': mode, 'epoch': epoch})
        pd.DataFrame(rows, columns=['target', 'SMILES', 'pred', 'mode', 'epoch']).to_csv(
            os.path.join(results, 'generated.csv'))
        pd.DataFrame({'loss': rl_losses, 'rewards': rewards}).to_csv(
            os.path.join(results, 'loss_reward_evolution.csv'))
    return rewards, rl_losses, rows
