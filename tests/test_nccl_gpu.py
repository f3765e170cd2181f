"""The collectives of the evaluation / timing paths under the backend the GPUs really use: RCCL ("nccl"), world size 1 on
the one GPU of the test box (RCCL refuses two ranks on one device, so this is the largest world a one-GPU box can form;
the world-size-2 and -3 control flow is covered under gloo in tests/test_sharding.py).  VERDICT r2 weak #3:
sharding.confusion_counts(reduce=True) all-reduced a CPU tensor, which raises under nccl and could not be seen under gloo."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r'''
import os, sys, socket
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from modulationdetectioncnn_amd import sharding, VTCNN2, Topology, synthetic_frames
assert str(dist.get_backend()).lower() == "nccl" and sharding.collective_device().type == "cuda"
# a CPU tensor is what must never reach the backend: prove that it would have raised
try:
    dist.all_reduce(torch.zeros(4, dtype=torch.int64))
    raised = False
except Exception:
    raised = True
assert raised, "RCCL took a CPU tensor?"
# every tensor the module hands to a collective lives in HBM
seen = []
orig = dist.all_reduce
def spy(t, *a, **k):
    seen.append(t.device.type)
    return orig(t, *a, **k)
dist.all_reduce = spy
y = np.arange(1000) %% 3
conf = sharding.confusion_counts(y, (y + (np.arange(1000) %% 7 == 0)) %% 3, 3)            # reduce=True
assert conf.sum() == 1000 and conf.dtype == np.int64 and np.trace(conf) == 1000 - 143
import time
el = sharding.timed_region(lambda: time.sleep(0.005), steps=3, warmup=1, sync=torch.cuda.synchronize, device=torch.device("cuda", 0))
assert el >= 0.015
assert seen and set(seen) == {"cuda"}, seen
# the sharded predictor's gather (object collective) under RCCL, with a real model on this rank's GPU
m = VTCNN2.synthetic(Topology.deployed(3, 3), seed=3, device=0)
X = synthetic_frames(257, seed=1)
p, l = sharding.ShardedPredictor.for_model(m).predict(X)
assert p.shape == (257, 3) and (l == p.argmax(1)).all()
dist.barrier()
dist.destroy_process_group()
print("NCCL-OK")
''' % ROOT


@pytest.mark.gpu
def test_collectives_of_the_evaluation_path_under_rccl_world1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NCCL-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


CHILD_BENCH = r'''
import os, sys, socket
sys.path.insert(0, %r)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)          # a rank somebody else's launcher started with a bare environment
import bench
bench.rank_environment(os.environ)
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
import torch
bench.select_device(0)
dist = bench.init_distributed(0)                             # "nccl" with device_id = this rank's GPU
assert str(dist.get_backend()).lower() == "nccl"
from modulationdetectioncnn_amd import sharding
import time
el = sharding.timed_region(lambda: time.sleep(0.002), steps=2, warmup=1, sync=torch.cuda.synchronize, device=torch.device("cuda", 0))
assert el >= 0.004
assert bench.agree_ok(dist, True, "leg")
dist.barrier()
dist.destroy_process_group()
print("BENCH-RCCL-OK")
''' % ROOT


@pytest.mark.gpu
def test_bench_rank_initialisation_under_rccl_world1():
    """bench.py's own rank set-up on the real backend: the environment every rank gets, RCCL bound to the rank's
    GPU with device_id, the barriers and the MAX-reduce of the timing contract and the leg agreement, at the largest world a one-GPU
    box can form."""
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    r = subprocess.run([sys.executable, "-c", CHILD_BENCH], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "BENCH-RCCL-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
