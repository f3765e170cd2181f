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


# ---------------------------------------------------------------------------------------------------------------
# one process, G lanes (per-GPU HIP streams): slicing, enqueue-all-then-sync-once, concatenation -- with fake lanes
class _FakeLane:
    """Records what the driver does to it; computes with the CPU oracle at sync time (like a stream would)."""

    def __init__(self, log, name, weights):
        self.log, self.name, self.w = log, name, weights
        self.pending = []

    def upload(self, x):
        self.log.append(("upload", self.name, len(x)))
        return np.array(x, np.float32)

    def forward(self, x):
        self.log.append(("forward", self.name, len(x)))
        out = {}
        self.pending.append((x, out))
        return ("probs", out), ("labels", out)          # placeholders, filled at sync

    def sync(self):
        from oracle import oracle_np as O
        self.log.append(("sync", self.name))
        for x, out in self.pending:
            out.update(O.forward_deployed(x, *self.w, dtype=np.float32))
        self.pending = []

    def download(self, t):
        return t[1][t[0]]


@pytest.mark.parametrize("lanes,n", [(1, 10), (2, 1001), (8, 100), (3, 2), (4, 0)])
def test_multistream_driver_slices_enqueues_then_syncs_once(lanes, n):
    from modulationdetectioncnn_amd import synthetic_frames
    from modulationdetectioncnn_amd.sharding import MultiStreamPredictor
    from oracle import oracle_np as O
    w = [a for p in load_deployed_npz("3convmodrecnets_CNN2_0.5") for a in p]
    log = []
    fl = [_FakeLane(log, g, w) for g in range(lanes)]
    msp = MultiStreamPredictor(fl, 3)
    assert [(lo, hi) for _, lo, hi in msp.plan(n)] == shard_bounds(n, lanes)
    X = synthetic_frames(n, seed=5)
    p, l = msp.predict(X)
    ref = O.forward_deployed(X, *w, dtype=np.float32)
    np.testing.assert_array_equal(l, ref["labels"])
    np.testing.assert_allclose(p.reshape(-1, 3), ref["probs"].reshape(-1, 3), atol=1e-6)
    kinds = [e[0] for e in log]
    # every forward is enqueued before the first sync; exactly one sync per lane
    assert kinds.index("sync") > max(i for i, k in enumerate(kinds) if k == "forward")
    assert kinds.count("sync") == lanes and kinds.count("forward") == lanes
    assert [e[2] for e in log if e[0] == "forward"] == [hi - lo for lo, hi in shard_bounds(n, lanes)]
    with pytest.raises(ValueError):
        msp.forward_shards([X] * (lanes + 1))


class _FakeHostModel:
    """Stands for one GPU's VTCNN2 in the host-buffer driver: records its calls and the thread they ran on."""

    def __init__(self, log, name, weights, barrier):
        self.log, self.name, self.w, self.barrier = log, name, weights, barrier

    def predict_host(self, X, out=None):
        import threading
        from oracle import oracle_np as O
        self.barrier.wait(timeout=30)        # every model's call must be in flight at once: one thread per model
        r = O.forward_deployed(np.asarray(X), *self.w, dtype=np.float32)
        out[0][...] = r["probs"]
        out[1][...] = r["labels"]
        self.log.append((self.name, len(X), threading.get_ident()))


@pytest.mark.parametrize("gpus,streams,n", [(1, 1, 10), (2, 1, 1001), (4, 2, 103), (3, 1, 2)])
def test_multistream_host_driver_runs_every_device_at_once(gpus, streams, n):
    """predict_host: one host thread per MODEL (not per lane), contiguous shards, results written in place."""
    import threading
    from types import SimpleNamespace
    from modulationdetectioncnn_amd import synthetic_frames
    from modulationdetectioncnn_amd.sharding import MultiStreamPredictor
    from oracle import oracle_np as O
    w = [a for p in load_deployed_npz("3convmodrecnets_CNN2_0.5") for a in p]
    log = []
    active = min(gpus, n)                      # shards of zero frames start no call
    barrier = threading.Barrier(active)
    models = [_FakeHostModel(log, g, w, barrier) for g in range(gpus)]
    msp = MultiStreamPredictor([SimpleNamespace(model=m) for m in models for _ in range(streams)], 3)
    X = synthetic_frames(n, seed=6)
    p, l = msp.predict_host(X)
    ref = O.forward_deployed(X, *w, dtype=np.float32)
    np.testing.assert_array_equal(l, ref["labels"])
    np.testing.assert_allclose(p, ref["probs"], atol=1e-6)      # (the numpy oracle's BLAS sums depend on the shard length)
    assert sorted((name, cnt) for name, cnt, _ in log) == [(g, hi - lo) for g, (lo, hi) in enumerate(shard_bounds(n, gpus)) if hi > lo]
    assert len({tid for _, _, tid in log}) == active


def test_device_is_resolved_once_and_mismatch_is_an_error(monkeypatch):
    """with device=None the engine, the workspace and the input check must all use the device that was
    current when the model was built -- not whatever torch.cuda.current_device() says later."""
    from modulationdetectioncnn_amd import VTCNN2, Topology
    cur = {"d": 0}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: cur["d"])
    m = VTCNN2(Topology.deployed(3, 3))
    assert m.device_index == 0
    cur["d"] = 1                              # the caller switches devices after construction
    assert m.device_index == 0
    assert VTCNN2(Topology.deployed(3, 3)).device_index == 1
    assert VTCNN2(Topology.deployed(3, 3), device="cuda:3").device_index == 3
    assert VTCNN2(Topology.deployed(3, 3), device=2).device_index == 2
    with pytest.raises(ValueError):
        VTCNN2(Topology.deployed(3, 3), device="cpu")

    class _T:                                 # a stand-in for a tensor on another GPU: only the checks run
        is_cuda, dtype, shape = True, torch.float32, (4, 2, 128)
        device = torch.device("cuda", 1)

        def is_contiguous(self):
            return True
    monkeypatch.setattr(torch, "Tensor", _T)
    with pytest.raises(ValueError, match="cuda:1.*cuda:0"):
        m.forward_device(_T())


# ---------------------------------------------------------------------------------------------------------------
# bench.py's N>1 control flow under gloo with a stub engine: the JSON line's n_gpus / global_batch / MAX-over-ranks time
def _bench_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), MDC_BENCH_BACKEND="gloo")
    import contextlib
    import io
    import time
    import bench
    from modulationdetectioncnn_amd import Topology
    from modulationdetectioncnn_amd.sharding import timed_region
    n = 4096

    class _Stub:
        topology = Topology.vtcnn2(11)
        dtype = "bf16"

    calls = []

    def run_workload(name, device, steps, warmup, dist=None, frames=None):
        assert dist is not None and dist.get_world_size() == world and device == rank
        calls.append((name, frames))
        el = timed_region(lambda: time.sleep(0.02 * (rank + 1)), steps, warmup)      # rank 1 is twice as slow
        return _Stub(), None, None, None, (n if frames is None else frames), el

    bench.run_workload = run_workload
    bench.select_device = lambda d: None
    bench.dominant_roofline = lambda *a, **k: ({"bound": "mfma", "frac": 0.0}, {})
    buf = io.StringIO()
    os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)      # a rank started by SOMEONE ELSE's launcher, bare environment
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--no-extras", "--no-cpu-baseline"])
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"   # ... gets the dmabuf-IPC switch RCCL needs all the same
    with open(os.path.join(out_dir, f"out{rank}.txt"), "w") as f:
        f.write(buf.getvalue())
    # every rank ran the headline and then the two other readings, each on ITS shard (strong: 2^20 / world frames)
    assert calls == [("vtcnn2-c11-bf16-n2^20", None), ("vtcnn2-c11-bf16-n2^20", (1 << 20) // world), ("vtcnn2-c11-bf16-n2^21", 1 << 21)], calls


def test_bench_main_under_gloo_world2(tmp_path):
    import json
    port = _free_port()
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "out1.txt").read().strip() == ""           # only rank 0 prints
    lines = [l for l in open(tmp_path / "out0.txt").read().splitlines() if l.strip()]
    assert len(lines) == 1                                            # ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["frames_per_gpu"] == 4096 and j["config"]["global_batch"] == 8192
    assert j["ms_per_step"] >= 40 * 0.9                               # MAX over ranks: the slow rank's 40 ms per step
    assert abs(j["value"] - 8192 * 4 / (j["ms_per_step"] * 4e-3)) / j["value"] < 1e-6     # whole-job frames / MAX time
    assert j["metric"].endswith("batch=4096)") and "cpu_baseline" not in j
    # the other two readings of "batch=2^20 at 1/2/4/8 MI355X" and of configs[3], as extra legs with their batch stated
    strong, shard = j["extra"]
    assert strong["scaling"] == "strong" and strong["global_batch"] == 1 << 20 and strong["frames_per_gpu"] == 1 << 19 and strong["n_gpus"] == 2
    assert shard["scaling"] == "weak" and shard["frames_per_gpu"] == 1 << 21 and shard["global_batch"] == 1 << 22 and shard["workload"].endswith("n2^21")
    for leg in (strong, shard):
        assert abs(leg["value"] - leg["global_batch"] * leg["steps"] / (leg["ms_per_step"] * leg["steps"] * 1e-3)) / leg["value"] < 1e-6
        assert leg["ms_per_step"] >= 40 * 0.9                         # MAX over ranks here too


_STUB_WRAPPER = """
import os, sys, time
sys.path.insert(0, {root!r})
import bench
from modulationdetectioncnn_amd import Topology
from modulationdetectioncnn_amd.sharding import timed_region


class _Stub:
    topology = Topology.vtcnn2(11)
    dtype = "bf16"


def run_workload(name, device, steps, warmup, dist=None, frames=None):
    rank = dist.get_rank() if dist else 0
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"      # the self-launched ranks' environment (the parent had none)
    bench.agree_ok(dist, True, name)
    el = timed_region(lambda: time.sleep(0.01 * (rank + 1)), steps, warmup)
    return _Stub(), None, None, None, (2048 if frames is None else frames), el


bench.run_workload = run_workload
bench.select_device = lambda d: None
bench.dominant_roofline = lambda *a, **k: ({{"bound": "mfma", "frac": 0.5}}, {{}})
sys.exit(bench.main(script=os.path.abspath(__file__)))
"""


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` typed at a shell (no torch.distributed.run around it, WORLD_SIZE
    unset) must still be a 2-rank job: the parent starts the launcher as a child process and relays rank 0's ONE line."""
    import json
    import subprocess
    wrapper = tmp_path / "bench_stub.py"
    wrapper.write_text(_STUB_WRAPPER.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "HSA_ENABLE_IPC_MODE_LEGACY")}
    env["MDC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(wrapper), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["global_batch"] == 4096 and j["config"]["frames_per_gpu"] == 2048
    assert j["ms_per_step"] >= 20 * 0.9                                # MAX over ranks: rank 1 sleeps 20 ms per step
    assert [e["scaling"] for e in j["extra"]] == ["strong", "weak"]
    # compact per-leg numbers, inside config and as the line's last key
    assert list(j)[-1] == "legs" and j["legs"] == j["config"]["legs_frames_per_s_and_roofline_frac"]
    assert j["legs"]["vtcnn2-c11-bf16-n2^20"] == [round(j["value"]), 0.5] and len(j["legs"]) == 3
    assert "starting" in r.stderr and "torch.distributed.run" in r.stderr


_SLEEPER_WRAPPER = """
import os, sys, time
sys.path.insert(0, {root!r})
import bench


def run_workload(name, device, steps, warmup, dist=None, frames=None):
    open(os.path.join({out!r}, "rank%d.pid" % dist.get_rank()), "w").write(str(os.getpid()))
    time.sleep(120)


bench.run_workload = run_workload
bench.select_device = lambda d: None
sys.exit(bench.main(script=os.path.abspath(__file__)))
"""


def test_self_launched_ranks_do_not_outlive_a_terminated_parent(tmp_path):
    """a driver timeout (SIGTERM to `python bench.py --gpus 2`) must take the launcher and BOTH ranks with it --
    orphans would keep their GPUs."""
    import signal
    import subprocess
    import time
    wrapper = tmp_path / "bench_sleeper.py"
    wrapper.write_text(_SLEEPER_WRAPPER.format(root=ROOT, out=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MDC_BENCH_BACKEND"] = "gloo"
    parent = subprocess.Popen([sys.executable, str(wrapper), "--gpus", "2", "--no-extras", "--no-cpu-baseline"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 120
        files = [tmp_path / "rank0.pid", tmp_path / "rank1.pid"]
        while time.time() < deadline and not all(f.exists() and f.read_text().strip() for f in files):
            assert parent.poll() is None, parent.stderr.read()[-2000:]
            time.sleep(0.2)
        pids = [int(f.read_text()) for f in files]
        parent.send_signal(signal.SIGTERM)
        assert parent.wait(timeout=40) == 128 + signal.SIGTERM

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            try:      # (a zombie still answers signal 0: look at its state)
                return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
            except OSError:
                return False
        t_end = time.time() + 20
        while time.time() < t_end and any(alive(p) for p in pids):
            time.sleep(0.2)
        assert not any(alive(p) for p in pids), pids
    finally:
        if parent.poll() is None:
            parent.kill()


def test_init_distributed_binds_rccl_to_the_ranks_gpu(monkeypatch):
    """nccl (= RCCL) gets device_id = this rank's GPU, so the communicator is built on it explicitly; gloo takes none."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    seen = []
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: seen.append((backend, kw)))
    bench.init_distributed(3, "nccl")
    bench.init_distributed(3, "gloo")
    assert seen == [("nccl", {"device_id": torch.device("cuda", 3)}), ("gloo", {})]
    env = bench.rank_environment({})
    assert env == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"} and bench.rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}) == {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}


def test_bench_refuses_a_launcher_whose_world_size_disagrees(monkeypatch, capsys):
    """... and never falls through to a run of another size labelled only by a note on stderr."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.main(["--gpus", "8", "--no-extras", "--no-cpu-baseline"]) == 2
    cap = capsys.readouterr()
    assert cap.out == "" and "refusing" in cap.err
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.main(["--gpus", "1"]) == 2


def _skip_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    try:
        bench.agree_ok(dist, True, "leg-a")                            # all fine: returns
        try:
            bench.agree_ok(dist, rank != 1, "leg-b")                   # rank 1 failed alone: EVERY rank raises
            res = "ran"
        except bench.LegSkipped as e:
            res = "skipped:" + str(e)
        dist.barrier()                                                 # the next leg's collectives still line up
        with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
            f.write(res)
    finally:
        dist.destroy_process_group()


def test_a_leg_that_fails_on_one_rank_is_skipped_by_all(tmp_path):
    """one rank's set-up failure (HBM shared with another job) must not leave its peers in the leg's barrier."""
    mp.spawn(_skip_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = open(tmp_path / "r0.txt").read(), open(tmp_path / "r1.txt").read()
    assert r0.startswith("skipped:") and r1.startswith("skipped:") and "(this one)" in r1 and "(this one)" not in r0


def test_collective_device_follows_the_backend(monkeypatch):
    """under RCCL ("nccl") a CPU tensor must never reach a collective; under gloo it must stay on the host."""
    from modulationdetectioncnn_amd import sharding

    class _D:
        def __init__(self, b):
            self.b = b

        def get_backend(self):
            return self.b
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert sharding.collective_device(_D("gloo")) == torch.device("cpu")
    assert sharding.collective_device(_D("cpu:gloo,cuda:nccl")) == torch.device("cpu")      # a host backend is there
    assert sharding.collective_device(_D("nccl")) == torch.device("cuda", 3)
    assert sharding.collective_device(_D("nccl"), device="cuda:5") == torch.device("cuda", 5)
    assert sharding.collective_device(_D("nccl"), device="cpu") == torch.device("cuda", 3)

    # every reduction of the module goes through all_reduce_array, which places its tensor there
    seen = []

    class _Dist(_D):
        class ReduceOp:
            SUM, MAX = "sum", "max"

        def all_reduce(self, t, op):
            seen.append((t.device.type, op))
    monkeypatch.setattr(sharding, "_dist", lambda: _Dist("gloo"))
    monkeypatch.setattr(sharding, "collective_device", lambda d=None, device=None: torch.device("meta"))
    with pytest.raises(Exception):          # a meta tensor cannot come back to numpy: what matters is where it WENT
        sharding.confusion_counts(np.array([0, 1]), np.array([1, 1]), 3)
    assert seen == [("meta", "sum")]


def test_bench_live_traffic_parsing_and_fallback(tmp_path, monkeypatch):
    """bench.py's roofline.traffic: the counter CSV of a rocprofv3 --pmc child pass is reduced per kernel slot by the MEDIAN
    over dispatches (a stray small launch does not move it), a slot's bytes are (2 x FETCH_SIZE + WRITE_SIZE) KiB (the
    guide's gfx950 correction), and without rocprofv3 nothing is recorded -- the line then falls back to the committed
    profile's figure and says so."""
    sys.path.insert(0, ROOT)
    import bench
    csv_path = tmp_path / "1_counter_collection.csv"
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    conv = '"void mdc::(anonymous namespace)::vt_conv_bf16_sched_kernel<0, false, false>(float const*, long)"'
    d1 = '"void mdc::(anonymous namespace)::vt_dense1_bf16_phased_kernel<0, true, false>(unsigned short const*, long)"'
    for i, v in enumerate([100.0, 525620.0, 525621.0, 525619.0, 525620.0]):          # one stray small launch
        rows.append(f"{i},{i},1,1,1,1,65536,7,{conv},256,0,0,128,128,64,FETCH_SIZE,{v},0,1")
    for i, v in enumerate([11152612.0, 11152610.0, 11152614.0]):
        rows.append(f"{9 + i},{9 + i},1,1,1,1,65536,8,{d1},512,0,0,128,128,64,FETCH_SIZE,{v},0,1")
    rows.append(f"20,20,1,1,1,1,65536,8,{d1},512,0,0,128,128,64,WRITE_SIZE,48.0,0,1")
    csv_path.write_text("\n".join(rows) + "\n")
    med = bench.pmc_medians(str(csv_path), "FETCH_SIZE", "bf16")
    assert med == {"mdc_vt_conv": 525620.0, "mdc_vt_dense1": 11152612.0}
    assert bench.pmc_medians(str(csv_path), "WRITE_SIZE", "bf16") == {"mdc_vt_dense1": 48.0}
    assert bench.pmc_medians(str(csv_path), "FETCH_SIZE", "f32") == {}
    # no rocprofv3 on PATH (this container's case is simulated explicitly): nothing recorded, no exception
    monkeypatch.setattr(bench.shutil if hasattr(bench, "shutil") else __import__("shutil"), "which", lambda name: None)
    bench.LIVE_TRAFFIC.clear()
    bench.live_traffic("vtcnn2-c11-bf16-n2^20")
    assert bench.LIVE_TRAFFIC == {}
    bench.live_traffic("deployed3-f32-n2^20")                        # not a VT-CNN2 workload: nothing to do
    assert bench.LIVE_TRAFFIC == {}
    traffic, src = bench.measured_traffic("mdc_vt_conv/bf16", 1 << 20)
    assert traffic is not None and "not measured in this run" in src and traffic / (1 << 20) == pytest.approx(22147, rel=2e-3)
