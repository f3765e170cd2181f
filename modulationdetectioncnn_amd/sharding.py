"""Multi-GPU: frames are independent, so a batch shards across ranks with NO data-path collective.

One process per GPU (torch.distributed; backend "nccl" = RCCL on the GPUs, "gloo" in the CPU
tests).  Rank g owns the contiguous slice [g*N/G, (g+1)*N/G) of the batch; the
weights (<= 11.3 MB) are replicated by each rank's own load.  The only collectives are OFF the
data path: the barriers around a timed region, the MAX of the elapsed times, and an optional
gather of the (N,) labels / (N,C) probabilities when a caller wants them in one place.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, ordered, exhaustive slices; sizes differ by at most one frame."""
    if world < 1 or n < 0:
        raise ValueError("need world >= 1 and n >= 0")
    return [((g * n) // world, ((g + 1) * n) // world) for g in range(world)]


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return shard_bounds(n, world)[rank]


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def world() -> Tuple[int, int]:
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def timed_region(step: Callable[[], None], steps: int, warmup: int, sync: Optional[Callable[[], None]] = None,
                 device=None) -> float:
    """bench.py's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over ranks."""
    d = _dist()
    sync = sync or (lambda: None)
    for _ in range(warmup):
        step()
    sync()
    if d:
        d.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if d:
        d.barrier()
    el = time.perf_counter() - t0
    if d:
        import torch
        t = torch.tensor([el], dtype=torch.float64, device=device if device is not None else "cpu")
        d.all_reduce(t, op=d.ReduceOp.MAX)
        el = float(t.item())
    return el


class ShardedPredictor:
    """predict() over a batch every rank can see (e.g. generated from a common seed or read from a
    shared file): each rank runs `engine` on its own slice only.

    engine(x_shard) -> (probs (m,C) float32, labels (m,) int32) as numpy arrays.  In production
    that is a VTCNN2 bound to this rank's GPU; the CPU tests inject the oracle."""

    def __init__(self, engine: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]], classes: int):
        self.engine = engine
        self.classes = classes
        self.rank, self.world = world()

    @classmethod
    def for_model(cls, model) -> "ShardedPredictor":
        def engine(x):
            return model.predict(x), model.predict_classes(x)
        return cls(engine, model.topology.classes)

    def predict_local(self, X: np.ndarray) -> Tuple[Tuple[int, int], np.ndarray, np.ndarray]:
        lo, hi = shard_range(len(X), self.rank, self.world)
        if hi == lo:
            return (lo, hi), np.zeros((0, self.classes), np.float32), np.zeros((0,), np.int32)
        p, l = self.engine(X[lo:hi])
        return (lo, hi), np.asarray(p, np.float32), np.asarray(l, np.int32)

    def predict(self, X: np.ndarray, gather: bool = True):
        """Every rank returns the full (N,C) probabilities and (N,) labels when gather=True (host-side
        concatenation in rank order, off the hot path); otherwise its own slice and bounds."""
        bounds, p, l = self.predict_local(X)
        if not gather or self.world == 1:
            return (p, l) if gather else (bounds, p, l)
        d = _dist()
        parts: List[object] = [None] * self.world
        d.all_gather_object(parts, (bounds, p, l))
        parts.sort(key=lambda t: t[0][0])
        assert [b for b, _, _ in parts] == shard_bounds(len(X), self.world)
        return np.concatenate([q for _, q, _ in parts]), np.concatenate([q for _, _, q in parts])


def confusion_counts(labels_true: np.ndarray, labels_pred: np.ndarray, classes: int, reduce: bool = True) -> np.ndarray:
    """cnn.py:199-216 `conf[j,k] += 1` as a C x C histogram; with reduce=True the per-rank histograms are
    summed over ranks (C*C integers: the only cross-GPU reduction the evaluation path ever needs)."""
    conf = np.zeros((classes, classes), np.int64)
    np.add.at(conf, (np.asarray(labels_true), np.asarray(labels_pred)), 1)
    d = _dist()
    if reduce and d:
        import torch
        t = torch.from_numpy(conf)
        d.all_reduce(t, op=d.ReduceOp.SUM)
        conf = t.numpy()
    return conf
