"""N>1 path on CPU: world_size-2 (and 3) gloo process groups.  The compute engine is the CPU
oracle (tests may use it); what is under test is the partitioning, the absence of any data-path
collective, the gather, the timing contract of bench.py and the confusion-count reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_deployed_npz
from modulationdetectioncnn_amd.sharding import shard_bounds, shard_range


def test_shard_bounds_properties():
    for n in (0, 1, 2, 5, 16, 17, 1 << 20, (1 << 24) + 3):
        for w in (1, 2, 3, 4, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
            assert shard_range(n, w - 1, w) == b[-1]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from modulationdetectioncnn_amd import synthetic_frames
        from modulationdetectioncnn_amd.sharding import ShardedPredictor, confusion_counts, timed_region
        from oracle import oracle_np as O
        w = [a for p in load_deployed_npz("3convmodrecnets_CNN2_0.5") for a in p]
        calls = []

        def engine(x):
            calls.append(len(x))
            r = O.forward_deployed(x, *w, dtype=np.float32)
            return r["probs"], r["labels"]

        X = synthetic_frames(n, seed=2016)                 # identical on every rank (common seed)
        sp = ShardedPredictor(engine, 3)
        assert (sp.rank, sp.world) == (rank, world)
        bounds, p_loc, l_loc = sp.predict(X, gather=False)
        assert bounds == shard_range(n, rank, world) and len(p_loc) == bounds[1] - bounds[0]
        assert calls == ([bounds[1] - bounds[0]] if bounds[1] > bounds[0] else [])   # only its own slice was computed
        p, l = sp.predict(X)
        ref = O.forward_deployed(X, *w, dtype=np.float32)
        np.testing.assert_array_equal(l, ref["labels"])
        np.testing.assert_allclose(p, ref["probs"], atol=1e-6)
        # evaluation-side reduction (cnn.py:199-216): per-rank histograms summed over ranks
        y_true = np.arange(n) % 3
        lo, hi = bounds
        conf = confusion_counts(y_true[lo:hi], l[lo:hi], 3)
        full = np.zeros((3, 3), np.int64)
        np.add.at(full, (y_true, ref["labels"]), 1)
        np.testing.assert_array_equal(conf, full)
        # timing contract: MAX over ranks, barriers both sides
        import time
        el = timed_region(lambda: time.sleep(0.01 * (rank + 1)), steps=3, warmup=1)
        assert el >= 0.03 * world * 0.9
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([el]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (2, 1), (3, 64)])
def test_gloo_sharded_predict(tmp_path, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    els = [float(np.load(tmp_path / f"ok{r}.npy")[0]) for r in range(world)]
    assert max(els) - min(els) < 1e-9            # every rank reports the same MAX
