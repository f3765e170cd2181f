"""Multi-GPU: frames are independent, so a batch shards across GPUs with NO data-path collective.

Two drivers over the same partition (shard g of G = the contiguous slice [g*N/G, (g+1)*N/G), SURVEY.md 8(e);
weights, <= 11.3 MB, replicated per GPU):

* one process per GPU (`ShardedPredictor`, `timed_region`; torch.distributed, backend "nccl" = RCCL on the GPUs,
  "gloo" in the CPU tests) -- the form bench.py's contract launches.  The only collectives are OFF the data path:
  the barriers around a timed region, the MAX of the elapsed times, and an optional gather of the (N,) labels /
  (N,C) probabilities when a caller wants them in one place;
* ONE process, one model handle and one HIP stream per GPU (`MultiStreamPredictor`) -- BASELINE.json configs[3]'s
  "per-GPU HIP streams": a single host thread enqueues every shard's forward on its device's stream, then
  synchronises each stream once.  No collective at all; mdc_forward never synchronises, so the G forwards overlap.
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


def collective_device(d=None, device=None):
    """The device a tensor must live on to enter a collective of the default process group: RCCL ("nccl") only takes
    tensors in HBM -- this process's current GPU (bench.py / the launcher select it per rank) unless the caller names
    one; gloo (the CPU tests) takes host tensors.  A CPU tensor handed to an nccl collective raises."""
    import torch
    d = d or _dist()
    backend = str(d.get_backend()).lower() if d else "gloo"
    if "nccl" not in backend or "gloo" in backend or "mpi" in backend:      # a host backend is there ("gloo", "cpu:gloo,cuda:nccl")
        return torch.device("cpu")
    if device is not None and torch.device(device).type == "cuda":
        return torch.device(device)
    return torch.device("cuda", torch.cuda.current_device())


def all_reduce_array(a: np.ndarray, op: str = "sum", device=None) -> np.ndarray:
    """Element-wise reduction of a small host array over the ranks (off the data path: elapsed times, C x C counts).
    The array travels through a tensor on `collective_device()`, so the same call is valid under RCCL and gloo."""
    d = _dist()
    a = np.ascontiguousarray(a)
    if not d:
        return a
    import torch
    t = torch.from_numpy(a.copy()).to(collective_device(d, device))
    d.all_reduce(t, op={"sum": d.ReduceOp.SUM, "max": d.ReduceOp.MAX}[op])
    return t.cpu().numpy()


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
        el = float(all_reduce_array(np.array([el], np.float64), "max", device)[0])
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
        """This rank's shard through a VTCNN2 bound to this rank's GPU: ONE upload, ONE forward (forward_device returns
        probabilities and labels of the same pass), one copy back per output."""
        def engine(x):
            import torch
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(f"cuda:{model.device_index}")
            probs, labels, _ = model.forward_device(xt)
            return probs.cpu().numpy(), labels.cpu().numpy()
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
        # (an object collective pickles on the host and, under RCCL, moves the bytes through the CURRENT GPU's memory:
        # torch does that staging itself, nothing here hands a host tensor to the backend)
        d.all_gather_object(parts, (bounds, p, l))
        parts.sort(key=lambda t: t[0][0])
        assert [b for b, _, _ in parts] == shard_bounds(len(X), self.world)
        return np.concatenate([q for _, q, _ in parts]), np.concatenate([q for _, _, q in parts])


class GpuLane:
    """One (model handle, HIP stream) pair of the one-process driver.  `forward` enqueues on the lane's stream and
    returns device tensors without synchronising; `sync` waits for the lane's stream."""

    def __init__(self, model, stream=None):
        import torch
        self.model = model
        self.device = torch.device("cuda", model.device_index)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)

    def upload(self, x: np.ndarray):
        import torch
        with torch.cuda.stream(self.stream):
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.device, non_blocking=True)

    def forward(self, x):
        import torch
        with torch.cuda.stream(self.stream):
            probs, labels, _ = self.model.forward_device(x)
        return probs, labels

    def sync(self) -> None:
        self.stream.synchronize()

    def download(self, t) -> np.ndarray:
        return t.cpu().numpy()


class MultiStreamPredictor:
    """BASELINE.json configs[3] as north_star words it: ONE process, G lanes (one model handle + one HIP stream per
    GPU), frames sharded contiguously over the lanes, every forward enqueued from one host thread, each lane
    synchronised ONCE at the end.  Lanes are anything with upload / forward / sync / download (GpuLane on the GPUs; the
    CPU tests inject a recording fake).  Several lanes may share a device (two streams of one GPU)."""

    def __init__(self, lanes, classes: int):
        if not lanes:
            raise ValueError("need at least one lane")
        self.lanes = list(lanes)
        self.classes = classes

    @classmethod
    def for_models(cls, models, streams_per_device: int = 1) -> "MultiStreamPredictor":
        """models: one finalized-on-first-use VTCNN2 per GPU (same weights, different `device`)."""
        models = list(models)
        lanes = [GpuLane(m) for m in models for _ in range(streams_per_device)]
        return cls(lanes, models[0].topology.classes)

    def plan(self, n: int) -> List[Tuple[int, int, int]]:
        """[(lane, lo, hi)]: the contiguous, ordered, exhaustive partition of n frames over the lanes."""
        return [(g, lo, hi) for g, (lo, hi) in enumerate(shard_bounds(n, len(self.lanes)))]

    def forward_shards(self, shards):
        """shards[g]: lane g's frames, already resident on lane g's device.  Enqueues all, syncs each lane once;
        returns [(probs, labels)] per lane (device tensors)."""
        if len(shards) != len(self.lanes):
            raise ValueError(f"{len(shards)} shards for {len(self.lanes)} lanes")
        outs = [lane.forward(x) for lane, x in zip(self.lanes, shards)]      # no sync inside: the G forwards overlap
        for lane in self.lanes:
            lane.sync()
        return outs

    def predict(self, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Host batch in, host results out (plumbing: PCIe-bound).  Uploads, forwards and syncs per lane as above."""
        X = np.asarray(X)
        plan = self.plan(len(X))
        xs = [self.lanes[g].upload(X[lo:hi]) for g, lo, hi in plan]
        outs = self.forward_shards(xs)
        p = [np.asarray(self.lanes[g].download(o[0]), np.float32).reshape(-1, self.classes) for (g, _, _), o in zip(plan, outs)]
        l = [np.asarray(self.lanes[g].download(o[1]), np.int32).reshape(-1) for (g, _, _), o in zip(plan, outs)]
        return np.concatenate(p), np.concatenate(l)


    def predict_host(self, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Host batch in, host results out with every GPU's streaming driver (mdc_predict_host: pinned ring, copy /
        compute / result streams) running at once: one host thread per model handle -- the C call releases the GIL
        -- each feeding its device its contiguous shard and writing straight into its slice of the result arrays.
        Per GPU the call runs at max(PCIe, kernel); nothing crosses between devices."""
        from concurrent.futures import ThreadPoolExecutor
        X = np.asarray(X)
        models = []
        for lane in self.lanes:
            if not any(lane.model is m for m in models):
                models.append(lane.model)
        n = len(X)
        probs = np.empty((n, self.classes), np.float32)
        labels = np.empty((n,), np.int32)
        jobs = [(m, lo, hi) for m, (lo, hi) in zip(models, shard_bounds(n, len(models))) if hi > lo]

        def run(job):
            m, lo, hi = job
            m.predict_host(X[lo:hi], out=(probs[lo:hi], labels[lo:hi]))
        if jobs:
            with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
                list(ex.map(run, jobs))      # list(): re-raises a worker's exception here
        return probs, labels


def confusion_counts(labels_true: np.ndarray, labels_pred: np.ndarray, classes: int, reduce: bool = True) -> np.ndarray:
    """cnn.py:199-216 `conf[j,k] += 1` as a C x C histogram; with reduce=True the per-rank histograms are
    summed over ranks (C*C integers: the only cross-GPU reduction the evaluation path ever needs)."""
    conf = np.zeros((classes, classes), np.int64)
    np.add.at(conf, (np.asarray(labels_true), np.asarray(labels_pred)), 1)
    if reduce:
        conf = all_reduce_array(conf, "sum")      # through HBM under RCCL, through host memory under gloo
    return conf
