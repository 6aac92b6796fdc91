"""Training-loop mirror of the reference drivers (hot-path part only).

``train`` follows /root/reference/examples/IC50/train_paccmann_rl.py:185-274 (preset ``'ic50'``)
and examples/affinity/train_conditional_generator.py:232-320 (preset ``'affinity'``): ``epochs x
range(1, steps)`` calls of ``learner.policy_gradient(target, epoch, batch_size)`` on a randomly
chosen training target, checkpoints ``gen_{epoch}.pt`` / ``enc_{epoch}.pt`` through ``learner.save``,
evaluation passes through ``learner.generate_compounds_and_evaluate`` and the CSV files
``results/generated.csv`` and ``results/loss_reward_evolution.csv`` with the reference's column
names.  What differs between the two reference scripts is reproduced per preset:

=========  ===========================  =================================  ==========================
preset     generated.csv columns        kept molecules                     checkpoints / evaluation
=========  ===========================  =================================  ==========================
ic50       cell_line, SMILES, IC50,     pred < learner.ic50_threshold      every epoch; last training
           mode, tox21                  (train_paccmann_rl.py:213-216)     target ('train') + a random
                                                                           held-out one ('test')
affinity   protein, SMILES, Binding     pred > 0.5                         every 10th epoch (:255-256);
           probability, mode, Tox21     (train_conditional_generator.py    one fixed held-out target
                                        :263-264)                          ('eval'); training targets
                                                                           never include it (:236-238)
=========  ===========================  =================================  ==========================

The toxicity column is filled by ``tox_fn(smiles) -> float`` when given (the reference calls the
Tox21 evaluator, out of scope here: SURVEY.md section 2 row 6) and is left out otherwise.  The
'affinity' preset also writes ``results/ratios.csv`` at the end (train_conditional_generator.py:283-289,
317-320): per epoch the percentage of ALL evaluated molecules with pred > 0.5 (``efficacy_ratio``) and,
with ``tox_fn``, of those with tox_fn == 1 (``tox_ratio``), both rounded to one decimal.  Out of scope:
the baseline comparison plots and the argparse/JSON handling.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np

PRESETS = {
    'ic50': dict(target_col='cell_line', pred_col='IC50', tox_col='tox21', save_every=1, modes=('train', 'test')),
    'affinity': dict(target_col='protein', pred_col='Binding probability', tox_col='Tox21', save_every=10,
                     modes=('eval',)),
}


def train(learner, train_targets: Sequence, test_targets: Sequence, params: dict, keep: Optional[Callable] = None,
          rng: Optional[np.random.RandomState] = None, log: Callable = print, save_every: Optional[int] = None,
          preset: str = 'ic50', tox_fn: Optional[Callable] = None):
    """Runs the loop and returns ``(rewards, rl_losses, generated_rows)``.

    keep(pred) -> bool selects the molecules written to generated.csv; the default is the preset's
    rule when it can be evaluated (``learner.ic50_threshold`` for 'ic50', 0.5 for 'affinity'), else
    every molecule.  ``save_every`` overrides the preset's checkpoint period."""
    import pandas as pd
    ps = PRESETS[preset]
    rng = rng if rng is not None else np.random
    save_every = ps['save_every'] if save_every is None else save_every
    rewards, rl_losses, rows = [], [], []
    results = os.path.join(learner.model_path, 'results')
    os.makedirs(results, exist_ok=True)
    os.makedirs(os.path.join(learner.model_path, 'weights'), exist_ok=True)
    if keep is None:
        if preset == 'affinity':
            keep = lambda p: p > 0.5                                     # noqa: E731
        elif hasattr(learner, 'ic50_threshold'):
            keep = lambda p: p < learner.ic50_threshold                  # noqa: E731
        else:
            keep = lambda p: True                                        # noqa: E731
    held_out = test_targets[0] if (preset == 'affinity' and len(test_targets)) else None
    pool = [t for t in train_targets if t != held_out] if held_out is not None else list(train_targets)
    columns = [ps['target_col'], 'SMILES', ps['pred_col'], 'mode'] + ([ps['tox_col']] if tox_fn is not None else [])
    target = None
    biased_ratios, tox_ratios = [], []
    for epoch in range(1, params['epochs'] + 1):
        for step in range(1, params['steps']):
            target = pool[rng.randint(len(pool))]
            rew, loss = learner.policy_gradient(target, epoch, params['batch_size'])
            log(f"Epoch {epoch:d}/{params['epochs']:d}, step {step:d}/{params['steps']:d}\t "
                f"loss={loss:.2f}, rew={float(rew):.2f}")
            rewards.append(float(rew))
            rl_losses.append(float(loss))
        if epoch % save_every == 0:
            learner.save(f'gen_{epoch}.pt', f'enc_{epoch}.pt')
        if preset == 'affinity':
            evals = [('eval', held_out)]
        else:
            evals = [('train', target),
                     ('test', test_targets[rng.randint(len(test_targets))] if len(test_targets) else None)]
        for mode, tgt in evals:
            if tgt is None:
                continue
            smiles, preds = learner.generate_compounds_and_evaluate(epoch, params.get('eval_batch_size', 32), tgt)
            if preset == 'affinity':     # train_conditional_generator.py:283-289 (nan for an empty evaluation, as there)
                pa = np.atleast_1d(np.asarray(preds, dtype=np.float64))
                with np.errstate(invalid='ignore', divide='ignore'):
                    biased_ratios.append(np.round(100 * (np.sum(pa > 0.5) / np.float64(len(pa))), 1))
                    if tox_fn is not None:
                        toxes = np.array([tox_fn(s) for s in smiles])
                        tox_ratios.append(np.round(100 * (np.sum(toxes == 1.) / np.float64(len(toxes))), 1))
            for s, p in zip(smiles, np.atleast_1d(preds)):
                if keep(p):
                    row = {ps['target_col']: tgt, 'SMILES': s, ps['pred_col']: float(p), 'mode': mode}
                    if tox_fn is not None:
                        row[ps['tox_col']] = tox_fn(s)
                    rows.append(row)
        pd.DataFrame(rows, columns=columns).to_csv(os.path.join(results, 'generated.csv'))
        pd.DataFrame({'loss': rl_losses, 'rewards': rewards}).to_csv(
            os.path.join(results, 'loss_reward_evolution.csv'))
    if preset == 'affinity':             # train_conditional_generator.py:317-320
        cols = {'efficacy_ratio': biased_ratios}
        if tox_fn is not None:
            cols['tox_ratio'] = tox_ratios
        pd.DataFrame(cols).to_csv(os.path.join(results, 'ratios.csv'))
    return rewards, rl_losses, rows
