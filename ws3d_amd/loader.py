"""Loader processes: the host-side preparation of scenes (reading a scan, the 16384-point sampler,
label generation, augmentation) for the NEXT mini-batches in forked worker processes -- the role of
``DataLoader(num_workers=...)`` in the reference's drivers (tools/train_rpn.py --workers).  Batch
composition and order are those of the index batches handed in; items whose preparation draws
random numbers use a per-process stream."""
from __future__ import annotations

import collections
import os
from typing import Iterable, Iterator, List

import numpy as np

_DATASET = None      # inherited by the forked processes


def _init(seed: int) -> None:
    seed = (seed + os.getpid()) % (2 ** 31)
    np.random.seed(seed)
    for holder in (_DATASET, getattr(_DATASET, "scenes", None)):
        rng = getattr(holder, "rng", None)
        if isinstance(rng, np.random.RandomState):
            rng.seed(seed)


def _item(i: int):
    return _DATASET[i]


def item_batches(dataset, index_batches: Iterable[List[int]], workers: int = 0, ahead: int = 3, seed: int = 0) -> Iterator[list]:
    """for every list of indices -> the list of ``dataset[i]``; workers > 0: prepared by that many forked
    processes, `ahead` batches in advance"""
    if workers <= 0:
        for ids in index_batches:
            yield [dataset[i] for i in ids]
        return
    import concurrent.futures
    import multiprocessing
    global _DATASET
    _DATASET = dataset
    pool = concurrent.futures.ProcessPoolExecutor(workers, mp_context=multiprocessing.get_context("fork"), initializer=_init,
                                                  initargs=(int(seed),))
    try:
        it, queue = iter(index_batches), collections.deque()
        done = False
        while True:
            while not done and len(queue) < max(1, ahead):
                try:
                    queue.append([pool.submit(_item, int(i)) for i in next(it)])
                except StopIteration:
                    done = True
            if not queue:
                return
            yield [f.result() for f in queue.popleft()]
    finally:
        pool.shutdown(wait=False, cancel_futures=True)
