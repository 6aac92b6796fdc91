"""Fan the per-molecule host work of the REINFORCE step out over worker processes.

Between the decode and the backward pass the reference runs, one molecule at a time on one core,
SELFIES -> SMILES, ``Chem.MolFromSmiles`` (reinforce.py:190-196) and the reward evaluators
(reinforce_omics.py:209-222).  At the batch sizes the GPU path is built for (4096 per GPU) that host
stage is 10-100x the GPU step, so it -- not the kernels -- sets the wall clock of ``policy_gradient``.
The functions themselves stay the caller's (RDKit, toxsmi and the predictors are out of scope here,
SURVEY.md section 2 rows 6-10); this module only spreads a per-item function over processes and
keeps the order, so that it can be handed to ``Reinforce(..., is_valid_batch=HostPool(is_valid))`` or
used inside a ``reward_fn``.

Start method: ``fork`` is cheapest but only safe while the parent has not initialised the GPU (a
forked HIP runtime has dead helper threads and possibly held locks), so the default is ``fork`` before
``torch.cuda.is_initialized()`` and ``forkserver`` after; ``spawn`` / ``forkserver`` import the
function's module in every worker (it must be picklable by reference).
"""
from __future__ import annotations

import multiprocessing as mp
import os
from typing import Callable, List, Optional, Sequence

_FN = None


def _init(fn):
    global _FN
    _FN = fn


def _run(chunk):
    out = []
    for item in chunk:
        try:
            out.append((True, _FN(item)))
        except Exception:      # the reference's evaluators swallow per-molecule failures (qed.py:33-37)
            out.append((False, None))
    return out


def _gpu_in_use() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_initialized())
    except Exception:      # pragma: no cover
        return False


class HostPool:
    """``pool(items) -> [fn(item) for item in items]`` evaluated by ``workers`` processes, order kept.

    fn:       picklable per-item function (e.g. SMILES -> bool, SMILES -> float)
    workers:  processes (default: the CPUs this process may run on, at most 16)
    chunk:    items per task; a failing item yields ``on_error`` instead of raising
    start:    multiprocessing start method; None = 'fork' while this process has not initialised the
              GPU, 'forkserver' once it has
    """

    def __init__(self, fn: Callable, workers: Optional[int] = None, chunk: int = 128, on_error=None,
                 start: Optional[str] = None):
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except AttributeError:      # pragma: no cover
                workers = os.cpu_count() or 1
            workers = max(1, min(16, workers))
        self.fn, self.workers, self.chunk, self.on_error = fn, int(workers), int(chunk), on_error
        if start is None:
            start = 'forkserver' if _gpu_in_use() else 'fork'
        self.start = start
        self._pool = mp.get_context(start).Pool(self.workers, initializer=_init, initargs=(fn,)) \
            if self.workers > 1 else None

    def __call__(self, items: Sequence) -> List:
        items = list(items)
        if not items:
            return []
        if self._pool is None:
            _init(self.fn)
            flat = _run(items)
        else:
            chunks = [items[i:i + self.chunk] for i in range(0, len(items), self.chunk)]
            flat = [y for part in self._pool.map(_run, chunks) for y in part]
        return [y if ok else self.on_error for ok, y in flat]

    def close(self) -> None:
        if self._pool is not None:
            self._pool.close()
            self._pool.join()
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def terminate(self) -> None:
        if self._pool is not None:
            self._pool.terminate()
            self._pool = None

    def __del__(self):      # best effort, and never a join() that could block interpreter shutdown
        try:
            self.terminate()
        except Exception:
            pass
