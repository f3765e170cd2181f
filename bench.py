#!/usr/bin/env python3
"""Headline benchmark: I/Q frames/s of the batched VT-CNN2 forward on MI355X.

Contract (one JSON line on rank 0):  python bench.py --gpus N --steps K --warmup W
For N>1 the driver launches one process per GPU with torch.distributed.run (and when nobody
does -- `python bench.py --gpus N` typed at a shell -- this file starts that launcher itself as a
child process and relays rank 0's line; a launcher whose WORLD_SIZE disagrees with --gpus is
refused with exit code 2, never silently measured as something else); frames are
independent, so each rank runs the same per-GPU batch on its own shard with NO data-path
collective ("scaling": "weak"); the only torch.distributed calls are the barriers around
the timed region and the MAX of the elapsed time.

A "step" = one pass of the hot path (conv -> ... -> softmax -> argmax) over one batch of
synthetic frames already resident in HBM.  Default workload = the configuration
BASELINE.json's metric is quoted on: canonical VT-CNN2 (11 classes), batch 2^20, bf16 MFMA
with f32 accumulation (configs[2]).  Other BASELINE configs that fit one GPU are run as
short `extra` legs (not the headline value):  --workload selects any of them as headline.

N > 1: the headline line is the weak-scaling reading (2^20 frames PER GPU).  The metric's
"batch=2^20 at 1/2/4/8 MI355X" also reads as a fixed 2^20-frame global batch, and configs[3]
names 2^24 frames over 8 GPUs (2^21 per GPU): both are run by every rank as `extra` legs
("scaling": "strong" / "weak", with global_batch and frames_per_gpu stated), same timing contract.

roofline: the dominant kernel's algorithmic FLOPs (MFMA-bound nets) or bytes (HBM-bound
nets) per launch / its mean launch duration, measured with HIP events recorded inside
mdc_forward on the launch stream in a separate, untimed pass.  roofline.traffic (VT-CNN2
workloads, N = 1, default run): HBM bytes per launch from two rocprofv3 --pmc child passes of
this very workload, run first (live_traffic); otherwise the committed profile's figure.
cpu_baseline: the numpy oracle ("port": a CPU restatement, NOT Keras -- Keras/TensorFlow
are not installed) timed on this box's host cores on a bounded sample; rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0                                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 5000.0}  # dense MFMA peaks (no sparsity)
PEAK_VALU_F32_TFLOPS = 157.3                                 # same guide: vector f32 peak (= the f32 matrix rate) at the 2.4 GHz spec clock
# What an FMA stream can ISSUE on this chip, measured (HISTORY.md 4.1; tools/microbench/gen_valu_banks.py, random mantissas, two
# or more waves per SIMD, wall time): one wave64 v_fma_f32 per 1.06 ns per SIMD -- 2 cycles at the ~1.9 GHz the chip holds under a
# vector load, not at 2.4 -- = 64 lanes x 2 FLOP x 1,024 SIMDs / 1.06 ns; v_pk_fma_f32 takes 2.28 ns (two FMAs per lane in the
# time of two v_fma_f32: packing saves issue slots, not time).  The VALU-bound legs report their fraction of BOTH.
VALU_F32_ISSUE_CEILING_TFLOPS = 64 * 2 * 1024 / 1.06e-9 / 1e12      # 123.6
# Integer roof of the Q6.12 legs: every 18 x 18-bit product pair is two v_mad_i64_i32, which issue at HALF the plain VALU rate
# (profiles/r03_imul_rate.log: 9.5 cycles per wave-instruction, 2.38 per SIMD with four waves) -> 64 lanes / 2.38 cycles x 1,024
# SIMDs x 2.4 GHz multiply-accumulates per second
INT_MAC_ISSUE_PEAK_TMACS = 64 / 2.38 * 1024 * 2.4e9 / 1e12           # 66.1
# deployed legs whose bounding roof is the f32 vector ALU, not HBM (the 10-filter net at every dtype's
# conv, the 3-filter net once its input is 256 B of raw bytes): they report bound = "valu" with the HBM fraction beside it
VALU_BOUND = {("deployed", 10, "f32", "frames"), ("deployed", 10, "f32", "u8"), ("deployed", 3, "f32", "u8"),
              ("deployed", 10, "bf16", "frames"), ("deployed", 10, "f16", "frames")}

WORKLOADS = {
    # name: (topology kind, filters, classes, dtype, per-GPU frames, weights)
    "vtcnn2-c11-bf16-n2^20": ("vtcnn2", 256, 11, "bf16", 1 << 20, "synthetic seed 2016"),
    "vtcnn2-c11-bf16-n2^21": ("vtcnn2", 256, 11, "bf16", 1 << 21, "synthetic seed 2016"),     # configs[3]: 2^24 over 8 GPUs
    "vtcnn2-c11-fp8-n2^20": ("vtcnn2", 256, 11, "fp8", 1 << 20, "synthetic seed 2016"),       # configs[4]: fp8 MFMA conv2
    "vtcnn2-c3-f32-n65536": ("vtcnn2", 256, 3, "f32", 1 << 16, "synthetic seed 2016"),
    "vtcnn2-c11-f32-n65536": ("vtcnn2", 256, 11, "f32", 1 << 16, "synthetic seed 2016"),
    "deployed3-f32-n2^20": ("deployed", 3, 3, "f32", 1 << 20, "3convmodrecnets_CNN2_0.5 (bundled)"),
    "deployed10-f32-n2^20": ("deployed", 10, 3, "f32", 1 << 20, "convmodrecnets_CNN2_0.5 (bundled)"),
    "deployed3-bf16-n2^20": ("deployed", 3, 3, "bf16", 1 << 20, "3convmodrecnets_CNN2_0.5 (bundled)"),       # dense layer on bf16 MFMA
    "deployed10-bf16-n2^20": ("deployed", 10, 3, "bf16", 1 << 20, "convmodrecnets_CNN2_0.5 (bundled)"),       # configs[2] read literally: that file, bf16
    "deployed3-f16-n2^20": ("deployed", 3, 3, "f16", 1 << 20, "3convmodrecnets_CNN2_0.5 (bundled)"),         # conv in packed f16, dense on f16 MFMA
    "deployed10-f16-n2^20": ("deployed", 10, 3, "f16", 1 << 20, "convmodrecnets_CNN2_0.5 (bundled)"),
    "deployed3-f32-n2^21": ("deployed", 3, 3, "f32", 1 << 21, "3convmodrecnets_CNN2_0.5 (bundled)"),        # configs[3], T1 reading
    "deployed3-fp8-n2^20": ("deployed", 3, 3, "fp8", 1 << 20, "5convmodrecnets_CNN2_0.5 (bundled)"),         # configs[4] read literally: that file, fp8 MFMA
    "cnnpy-f32-n2^20": ("cnnpy", 10, 5, "f32", 1 << 20, "synthetic seed 2016"),                           # cnn.py literal model
}
DEFAULT = "vtcnn2-c11-bf16-n2^20"
EXTRAS = ["vtcnn2-c3-f32-n65536", "vtcnn2-c11-fp8-n2^20", "deployed3-f32-n2^20", "deployed10-f32-n2^20",
          "deployed3-bf16-n2^20", "deployed10-bf16-n2^20", "deployed3-f16-n2^20", "deployed10-f16-n2^20", "deployed3-fp8-n2^20",
          "cnnpy-f32-n2^20"]


def make_model(name, device):
    from modulationdetectioncnn_amd import VTCNN2, Topology
    kind, filters, classes, dtype, n, _ = WORKLOADS[name]
    if kind == "vtcnn2":
        m = VTCNN2.synthetic(Topology.vtcnn2(classes), seed=2016, device=device, dtype=dtype)
    elif kind == "cnnpy":
        m = VTCNN2.synthetic(Topology.cnnpy(filters, 10, classes), seed=2016, device=device, dtype=dtype)
    else:
        g = os.path.join(ROOT, "tests", "golden", "weights")
        f = "3convmodrecnets_CNN2_0.5.npz" if filters == 3 else "convmodrecnets_CNN2_0.5.npz"
        if name == "deployed3-fp8-n2^20":
            f = "5convmodrecnets_CNN2_0.5.npz"
        m = VTCNN2.from_npz(os.path.join(g, f), device=device, dtype=dtype)
    return m, n, dtype


def measured_traffic(key, frames_per_launch):
    """(HBM bytes per launch, where the number comes from).  NOT measured in this run: PMC counters need rocprofv3 around
    the process, so the bytes per frame come from the newest COMMITTED rocprofv3 PMC passes (profiles/rNN_traffic.json,
    made by tools/summarize_profiles.py on the GPU box: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections
    applied) and are scaled by this run's frames per launch; (None, None) if no profile covers this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    k = json.load(open(files[-1])).get("kernels", {}).get(key)
    if k is None:
        return None, None
    return k["hbm_bytes_per_frame"] * frames_per_launch, f"profiles/{os.path.basename(files[-1])} (committed rocprofv3 PMC pass, bytes per frame x frames per launch; not measured in this run)"


# rocprofv3 kernel-name substrings of the profiling slots, per dtype (tools/summarize_profiles.py uses the same ones)
PMC_KERNELS = {"mdc_vt_conv": {"bf16": "vt_conv_bf16", "fp8": "vt_conv_fp8_kernel", "f32": "vt_conv_f32_kernel"},
               "mdc_vt_dense1": {"bf16": "vt_dense1_bf16", "fp8": "vt_dense1_bf16", "f32": "vt_dense1_f32_kernel"}}
LIVE_TRAFFIC = {}      # {"<slot>/<dtype>": (HBM bytes per launch, frames per launch)} measured by live_traffic() in THIS run


def pmc_medians(csv_path, counter, dtype):
    """{profiling slot: median Counter_Value over that kernel's dispatches} from one rocprofv3 *_counter_collection.csv
    (columns Kernel_Name, Counter_Name, Counter_Value; one row per dispatch and counter).  The median, not the mean: a
    child run's launches are all alike, and a stray small launch must not move the figure."""
    import csv
    vals = {}
    with open(csv_path, newline="") as fd:
        for row in csv.DictReader(fd):
            if row.get("Counter_Name") != counter:
                continue
            for slot, names in PMC_KERNELS.items():
                if names[dtype] in row.get("Kernel_Name", ""):
                    vals.setdefault(slot, []).append(float(row["Counter_Value"]))
    return {slot: sorted(v)[len(v) // 2] for slot, v in vals.items()}


def live_traffic(name, timeout_s=120):
    """HBM bytes per launch of the workload's VT-CNN2 kernels, measured in THIS run: two child processes -- `rocprofv3 --pmc
    FETCH_SIZE -- python3 bench.py --workload <name> --steps 2 ...` and the same with WRITE_SIZE, separate passes as
    MI355X_MICROARCH.md's HBM section prescribes -- started BEFORE this process touches the GPU, their counter CSVs averaged
    per kernel, corrected as the guide says (KiB -> bytes; gfx950's FETCH_SIZE reports half of a wide coalesced read: x 2).
    Any failure (no rocprofv3, a refused or timed-out pass) leaves LIVE_TRAFFIC empty and the line falls back to the
    committed profile's figure, saying so in traffic_source."""
    import glob
    import shutil
    import subprocess
    import tempfile
    kind, _f, _c, dtype, n, _w = WORKLOADS[name]
    if kind != "vtcnn2" or shutil.which("rocprofv3") is None:
        return
    per_launch = min(n, 1 << 20) if dtype in ("bf16", "fp8") else min(n, 1 << 16)
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mdc_pmc_", dir="/tmp")
        try:
            # the program after `--` is THIS interpreter (sys.executable: never a PATH shim that would exec again inside a process
            # the profiler has already GPU-initialised, and the one interpreter known to have torch)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--workload", name, "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                print(f"[bench] live HBM-traffic pass {counter} failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
                return
            for slot, v in pmc_medians(max(files, key=os.path.getmtime), counter, dtype).items():
                got.setdefault(slot, {})[counter] = v
        except Exception as e:      # noqa: BLE001 -- measurement garnish: never in the way of the timed run
            print(f"[bench] live HBM-traffic pass {counter} failed: {e!r}", file=sys.stderr)
            return
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for slot, c in got.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            LIVE_TRAFFIC[f"{slot}/{dtype}"] = ((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, per_launch)


def dominant_roofline(m, x, probs, labels, steps):
    """Untimed profiling pass: per-kernel HIP-event time -> roofline object of the dominant kernel."""
    import torch
    topo = m.topology
    m.set_profiling(True)
    for _ in range(steps):
        m.forward_device(x, probs=probs, labels=labels, batch_size=getattr(m, "bench_chunk", None))
    torch.cuda.synchronize()
    prof = m.read_profile()
    m.set_profiling(False)
    name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
    n = x.shape[0]
    launches_per_step = max(1, cnt // steps)
    frames_per_launch = n / launches_per_step
    avg_ms = ms / cnt
    # (a slot without launches: the VT-CNN2 head, which the 16-bit batch path runs inside dense1's epilogue)
    kernels = {k: {"ms_per_step": v[0] / steps, "launches_per_step": v[1] // steps} for k, v in prof.items()}
    if topo.kind == "vtcnn2":
        fused_head = prof.get("mdc_vt_head", (0.0, 0))[1] == 0
        flops = {"mdc_vt_conv": topo.conv_flops_per_frame, "mdc_vt_dense1": 2 * 10560 * 256 + (2 * 256 * topo.classes if fused_head else 0),
                 "mdc_vt_head": 2 * 256 * topo.classes}[name] * frames_per_launch
        ach = flops / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[m.dtype]
        live = LIVE_TRAFFIC.get(f"{name}/{m.dtype}")
        if live and live[1] == frames_per_launch:
            traffic, tsrc = live[0], ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this workload "
                                      "(bench.live_traffic), per launch, gfx950 corrections applied")
        else:
            traffic, tsrc = measured_traffic(f"{name}/{m.dtype}", frames_per_launch)
        # the WHOLE step against the same roof (SURVEY.md 8(d)'s 38,252,032 FLOP per frame at C = 11): every kernel's launch time
        step_ms = sum(v[0] for v in prof.values()) / steps
        whole = topo.flops_per_frame * n / (step_ms * 1e-3) / 1e12
        rl = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
              "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc,
              "avg_launch_ms": avg_ms, "frames_per_launch": frames_per_launch,
              "whole_step_frac": whole / peak, "whole_step_achieved": whole, "whole_step_kernel_ms": step_ms}
    else:
        rl = deployed_roofline(topo, m.dtype, name, avg_ms, frames_per_launch, "frames")
    return rl, kernels


def deployed_roofline(topo, dtype, name, avg_ms, frames_per_launch, source):
    """HBM roofline on the algorithmic bytes (1,024 B of f32 frame -- or 256 B of raw uint8 I/Q -- in, 4C out), or, for
    the legs DESIGN.md calls VALU-bound, the f32 vector roofline on the algorithmic FLOPs with the HBM fraction beside it."""
    by = (topo.io_bytes_per_frame if source == "frames" else 256 + 4 * topo.classes) * frames_per_launch
    gbs = by / (avg_ms * 1e-3) / 1e9
    key = (f"{name}/F{topo.filters}" + ("/" + dtype if dtype != "f32" else "") + ("/u8" if source == "u8" else "")) if topo.kind == "deployed" else name
    traffic, tsrc = measured_traffic(key, frames_per_launch)
    if (topo.kind, topo.filters, dtype, source) in VALU_BOUND:
        tf = topo.flops_per_frame * frames_per_launch / (avg_ms * 1e-3) / 1e12
        return {"bound": "valu", "kernel": name, "achieved": tf, "peak": PEAK_VALU_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_VALU_F32_TFLOPS, "peak_is": "vector f32 spec peak (256 CUs x 2.4 GHz x 256 FLOP/clk)",
                "issue_ceiling": VALU_F32_ISSUE_CEILING_TFLOPS, "frac_of_issue_ceiling": tf / VALU_F32_ISSUE_CEILING_TFLOPS,
                "issue_ceiling_is": "measured: one wave64 v_fma_f32 per 1.06 ns per SIMD on random data (v_pk_fma_f32: 2.28 ns), HISTORY.md 4.1",
                "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": avg_ms,
                "frames_per_launch": frames_per_launch, "hbm_gbs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}
    return {"bound": "hbm", "kernel": name, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": avg_ms, "frames_per_launch": frames_per_launch}


def usable_cpus():
    """Host threads this process may really use: CPU affinity capped by the cgroup quota (a GPU box hands each job a
    share of its cores; asking a BLAS/OpenMP pool for every core of the machine oversubscribes that share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(name, budget_s=20.0):
    """The CPU restatement on the host cores, bounded sample of the same workload (same seeds).  Two ports are
    timed -- torch-CPU library ops (oracle_torch) and plain numpy (oracle_np) -- and the faster one is reported;
    neither is Keras (which cannot be installed here): kind = "port"."""
    import torch
    from modulationdetectioncnn_amd import synthetic_frames
    from oracle import oracle_np as O        # checker / baseline only
    from oracle import oracle_torch as OT
    kind, filters, classes, dtype, n, _ = WORKLOADS[name]
    from modulationdetectioncnn_amd import VTCNN2, Topology
    if kind == "vtcnn2":
        w = VTCNN2.synthetic(Topology.vtcnn2(classes), seed=2016).get_weights()
        sample = 2048
    elif kind == "cnnpy":
        w = VTCNN2.synthetic(Topology.cnnpy(filters, 10, classes), seed=2016).get_weights()
        sample = 65536
    else:
        g = os.path.join(ROOT, "tests", "golden", "weights")
        f = "3convmodrecnets_CNN2_0.5.npz" if filters == 3 else "convmodrecnets_CNN2_0.5.npz"
        w = VTCNN2.from_npz(os.path.join(g, f)).get_weights()
        sample = 65536
    x = np.asarray(synthetic_frames(sample, seed=2016))
    ncpu = usable_cpus()
    torch.set_num_threads(ncpu)
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(ncpu)
    except Exception:
        pass

    def timed(fn, budget):
        fn(x[: max(1, sample // 8)])        # warm-up
        t0 = time.perf_counter()
        done = 0
        while True:
            fn(x)
            done += sample
            el = time.perf_counter() - t0
            if el > budget or done >= 32 * sample:
                return done, el

    ports = {"torch-CPU ops": (lambda a: OT.forward(kind, a, w), torch.get_num_threads()),
             "numpy": (lambda a: O.forward(kind, a, w, dtype=np.float32), None)}
    best = None
    for label, (fn, threads) in ports.items():
        done, el = timed(fn, budget_s * 0.4)
        if threads is None:
            threads = ncpu
        cand = {"value": done / el, "unit": "frames/s", "cores": int(threads), "kind": "port",
                "sample": f"{done} frames ({done // sample} x {sample}) of {name}, {label} f32 restatement (not Keras), {el:.1f} s"}
        if best is None or cand["value"] > best["value"]:
            other = best
            best = cand
        else:
            other = cand
    best["other_port"] = {"value": other["value"], "sample": other["sample"]}
    return best


def launch_chunk(kind, dtype, n):
    """Frames per mdc_forward call.  The library's default for the VT-CNN2 family is 65,536 (1.45 GB of workspace per
    stream); the bench has the HBM to spare and asks for ONE launch of each kernel per 2^20 frames in the 16-bit modes
    (23 GB of workspace, +1.5 % over sixteen launches) -- stated in the JSON line's config."""
    if kind == "vtcnn2" and dtype in ("bf16", "fp8"):
        return min(n, 1 << 20)
    return None


class LegSkipped(RuntimeError):
    """Raised on EVERY rank when any rank could not set a leg up (so no rank enters that leg's collectives alone)."""


def agree_ok(dist, ok, what):
    """All ranks learn whether every rank's setup succeeded: one tiny MAX-reduce of a failure flag, OUTSIDE the timed
    region.  A rank that failed alone (e.g. out of HBM on a GPU it shares) must not leave its peers waiting in the
    leg's barrier while it moves on to the next leg's."""
    if not dist:
        return bool(ok)
    from modulationdetectioncnn_amd.sharding import all_reduce_array
    failed = float(all_reduce_array(np.array([0.0 if ok else 1.0], np.float64), "max")[0])
    if failed:
        raise LegSkipped(f"{what}: set-up failed on at least one rank" + ("" if ok else " (this one)"))
    return True


def run_workload(name, device, steps, warmup, dist=None, frames=None):
    """frames: per-GPU batch of THIS rank when it differs from the workload's own (the strong-scaling leg)."""
    import torch
    from modulationdetectioncnn_amd import synthetic_frames
    err = None
    try:
        m, n, dtype = make_model(name, device)
        if frames is not None:
            n = int(frames)
        rank = dist.get_rank() if dist else 0
        x = synthetic_frames(n, seed=2016 + rank, device=f"cuda:{device}")
        probs = torch.empty((n, m.topology.classes), dtype=torch.float32, device=x.device)
        labels = torch.empty((n,), dtype=torch.int32, device=x.device)
        chunk = launch_chunk(WORKLOADS[name][0], dtype, n)
        m.bench_chunk = chunk      # (the untimed profiling pass of dominant_roofline launches the same way)
        m._engine()                         # library loaded, model created, weights packed and uploaded (no launch: the
        if chunk:                           # profiles average per kernel over this process's launches)
            m.reserve_workspace(chunk)      # the big allocation happens HERE, where a failure can still be agreed on
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001 -- reported below, on every rank
        err = e
    if dist:
        try:
            agree_ok(dist, err is None, name)
        except LegSkipped:
            if err is not None:
                raise LegSkipped(f"{name}: {err!r}") from err
            raise
    elif err is not None:
        raise err
    from modulationdetectioncnn_amd.sharding import timed_region
    # barrier + torch.cuda.synchronize() on both sides of exactly `steps` steps, MAX over ranks
    el = timed_region(lambda: m.forward_device(x, probs=probs, labels=labels, batch_size=chunk), steps, warmup,
                      sync=torch.cuda.synchronize, device=x.device)
    return m, x, probs, labels, n, el


def multi_gpu_legs(name, ngpu, rank):
    """The other two readings of BASELINE.json's metric / configs at N > 1 (the headline line is weak scaling at the
    workload's own per-GPU batch): [(label, scaling, workload, frames of this rank, global batch)]."""
    from modulationdetectioncnn_amd.sharding import shard_range
    kind, filters, classes, dtype, n, _ = WORKLOADS[name]
    lo, hi = shard_range(n, rank, ngpu)
    legs = [(f"metric read as a fixed global batch of {n} frames (strong scaling)", "strong", name, hi - lo, n)]
    big = {"vtcnn2-c11-bf16-n2^20": "vtcnn2-c11-bf16-n2^21", "deployed3-f32-n2^20": "deployed3-f32-n2^21"}.get(name)
    if big:
        per = WORKLOADS[big][4]
        legs.append((f"configs[3]: 2^24 frames over 8 GPUs = {per} per GPU (weak scaling at that shard)", "weak", big, per, per * ngpu))
    return legs


def run_iq_u8(filters, device, steps=20, warmup=5, n=1 << 22):
    """Extra leg: raw RTL-SDR bytes (256 B/frame) straight into a deployed net (mdc_forward_iq_u8)
    with the two-pass path (mdc_iq_u8_windows + mdc_forward) beside it.  Roofline from the kernel's own launch time
    (HIP events inside the library, untimed extra pass) on the algorithmic bytes (256 in + 12 out) or, where the leg is
    VALU-bound, on its FLOPs."""
    import torch
    from modulationdetectioncnn_amd import frames_from_iq_u8
    from modulationdetectioncnn_amd.sharding import timed_region
    m, _, _ = make_model("deployed3-f32-n2^20" if filters == 3 else "deployed10-f32-n2^20", device)
    iq = torch.randint(0, 256, (n * 256,), dtype=torch.uint8, device=f"cuda:{device}")
    scale = 0.02 / 127.5
    el = timed_region(lambda: m.predict_iq_u8(iq, scale), steps, warmup, sync=torch.cuda.synchronize, device=iq.device)
    m.set_profiling(True)
    for _ in range(5):
        m.predict_iq_u8(iq, scale)
    torch.cuda.synchronize()
    (kname, (kms, kcnt)), = m.read_profile().items()
    m.set_profiling(False)
    probs = torch.empty((n, 3), dtype=torch.float32, device=iq.device)
    labels = torch.empty((n,), dtype=torch.int32, device=iq.device)
    el2 = timed_region(lambda: m.forward_device(frames_from_iq_u8(iq, scale), probs=probs, labels=labels), steps, warmup,
                       sync=torch.cuda.synchronize, device=iq.device)
    # the same bytes through the f16-mode kernel (conv in packed f16, dense on the f16 MFMA)
    from modulationdetectioncnn_amd import VTCNN2
    g = os.path.join(ROOT, "tests", "golden", "weights")
    mh = VTCNN2.from_npz(os.path.join(g, "3convmodrecnets_CNN2_0.5.npz" if filters == 3 else "convmodrecnets_CNN2_0.5.npz"), device=device, dtype="f16")
    elh = timed_region(lambda: mh.predict_iq_u8(iq, scale), steps, warmup, sync=torch.cuda.synchronize, device=iq.device)
    # sliding windows over one capture (hop 16 pairs = 32 new bytes per window): the live-stream form of README.md:5
    hop = 16
    nw = (n * 128 - 128) // hop + 1
    nw = min(nw, 1 << 22)
    elw = timed_region(lambda: m.predict_iq_u8(iq[: 2 * (128 + hop * (nw - 1))], scale, hop=hop), 5, 2, sync=torch.cuda.synchronize, device=iq.device)
    return {"workload": f"deployed{filters}-iq-u8-n2^22", "value": n * steps / el, "unit": "frames/s", "ms_per_step": el / steps * 1e3,
            "dtype": "f32", "input": "uint8 interleaved I/Q, 256 B/frame, resident in HBM",
            "two_pass_value": n * steps / el2, "f16_mode_value": n * steps / elh,
            "sliding_windows_hop16_value": nw * 5 / elw,
            "roofline": deployed_roofline(m.topology, "f32", kname, kms / kcnt, n, "u8")}


def run_q612(filters, device, steps=10, warmup=3, n=1 << 20):
    """Extra leg: the FPGA's Q6.12 integer forward (mdc_forward_q612) on float frames resident in HBM.
    Algorithmic bytes: 1,024 in + 12 (class sums) + 4 (label) out per frame; the kernel is integer-VALU-bound (18 x 18
    bit products in 64-bit arithmetic), so the HBM fraction is a report, not its roof."""
    import torch
    from modulationdetectioncnn_amd import synthetic_frames
    from modulationdetectioncnn_amd.sharding import timed_region
    m, _, _ = make_model("deployed3-f32-n2^20" if filters == 3 else "deployed10-f32-n2^20", device)
    x = synthetic_frames(n, seed=2016, sigma=0.3, device=f"cuda:{device}")
    el = timed_region(lambda: m.predict_q612(x, as_float=False), steps, warmup, sync=torch.cuda.synchronize, device=x.device)
    gbs = (1024 + 16) * n * steps / el / 1e9
    traffic, tsrc = measured_traffic(f"mdc_deployed_q612/F{filters}", n)
    # algorithmic integer work: one 18 x 18-bit multiply-accumulate per conv tap and per dense weight = 258 F (2 + 3) per frame
    tmacs = 258 * filters * 5 * n * steps / el / 1e12
    return {"workload": f"deployed{filters}-q612-n2^20", "value": n * steps / el, "unit": "frames/s", "ms_per_step": el / steps * 1e3,
            "dtype": "int18/int32 (Q6.12)",
            "roofline": {"bound": "int-valu", "kernel": "mdc_deployed_q612", "achieved": tmacs, "peak": INT_MAC_ISSUE_PEAK_TMACS, "unit": "TMAC/s",
                         "frac": tmacs / INT_MAC_ISSUE_PEAK_TMACS,
                         "peak_is": "issue rate of v_mad_i64_i32 (half the plain VALU rate, profiles/r03_imul_rate.log) x 64 lanes x 1,024 SIMDs x 2.4 GHz; "
                                    "the bit selection {m[35], m[28:12]}, the 18-bit wraps and the wave reductions come on top of the multiplies",
                         "hbm_gbs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": tsrc,
                         "from": "wall time incl. two small output allocations, one launch per step"}}


def run_training(device, epochs=20):
    """Extra leg: the training step of SURVEY.md 8(f) item 4 at the reference's own geometry -- CNN.ipynb cell 5 / 7: 18,900
    training frames, batch_size 1024 (18 full batches + one of 468), Adam -- for the two deployed nets and cnn.py's literal
    one.  Frames resident in HBM, an epoch = 19 x 2 launches enqueued back to back with its shuffle as an index array, one
    synchronisation per epoch; beside it the same epoch replayed as ONE captured hipGraph, and the numpy oracle's
    train_step on this box's host cores (a port, NOT Keras)."""
    import torch
    from modulationdetectioncnn_amd import Topology, synthetic_weights, synthetic_frames
    from modulationdetectioncnn_amd.training import Trainer, to_onehot
    from oracle import oracle_train as OT      # cpu baseline only
    n, batch = 18900, 1024
    rows = []
    for tag, kind, topo in (("deployed3", "deployed", Topology.deployed(3, 3)), ("deployed10", "deployed", Topology.deployed(10, 3)),
                            ("cnnpy", "cnnpy", Topology.cnnpy(10, 10, 5))):
        w = synthetic_weights(topo, seed=2016)
        x = synthetic_frames(n, seed=2016, device=f"cuda:{device}") * (1.0 if kind == "deployed" else 40.0)
        lab = torch.randint(0, topo.classes, (n,), device=x.device)
        tr = Trainer(topo, w, device=device)
        xd, yd = tr._frames(x), tr._targets(lab, n)
        order = torch.randperm(n, device=x.device).to(torch.int32)

        def epoch():
            for s0 in range(0, n, batch):
                tr.train_batch(xd, yd, order, s0, min(batch, n - s0))
        epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            epoch()
            tr.read()                                   # the epoch's one synchronisation (fit reads loss / val_loss here)
        el = (time.perf_counter() - t0) / epochs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                epoch()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        reps = []
        for _ in range(epochs):      # each replay timed on its own, the MEDIAN reported (a one-off 50 ms stall of a single replay -- seen once
            t0 = time.perf_counter()  # in three runs -- would otherwise pass for ten slow epochs)
            g.replay()
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t0)
        elg = sorted(reps)[len(reps) // 2]
        steps_per_epoch = (n + batch - 1) // batch
        xb = x[:batch].cpu().numpy()
        yb = to_onehot(lab[:batch].cpu().numpy(), topo.classes)
        wo = [(k.copy(), b.copy()) for k, b in w]
        opt = OT.KerasAdam([t.shape for t in OT.flatten_weights(wo)])
        OT.train_step(kind, xb, yb, wo, opt, np.float32)
        t0 = time.perf_counter()
        for _ in range(5):
            OT.train_step(kind, xb, yb, wo, opt, np.float32)
        cpu = (time.perf_counter() - t0) / 5
        rows.append({"net": tag, "frames_per_s": n / el, "epoch_ms": el * 1e3, "us_per_step": el / steps_per_epoch * 1e6,
                     "graph_replay_epoch_ms": elg * 1e3, "graph_replay_frames_per_s": n / elg,
                     "cpu_port_ms_per_step": cpu * 1e3, "cpu_port_frames_per_s": batch / cpu})
        tr.close()
    return {"workload": "training step (forward + loss + backward + Adam), 18,900 frames per epoch in batches of 1,024 (CNN.ipynb cell 5/7), f32",
            "unit": "frames/s", "value": rows[0]["frames_per_s"], "rows": rows,
            "note": "launch-latency bound at this batch: 1,024 frames x 2,334 parameters per step; cpu_port = numpy oracle_train.train_step, not Keras"}


def run_host_path(device, steps=4, warmup=1):
    """Extra leg: the PCIe-INCLUSIVE rate -- frames in pageable host memory (the numpy array of cnn.py:198), results back
    in host memory, through the library's streaming driver (mdc_predict_host: pinned ring, copy / compute / result
    streams).  Never the headline `value`, which starts from HBM-resident input.  Every timed call gets a FRESH array
    (the HIP runtime remembers host ranges it has pinned once; a caller's next batch is new memory).  Beside it: the same
    call on memory the caller pinned, and the unpipelined form (torch .to(device) + forward + .cpu())."""
    import time
    import numpy as np
    import torch
    from modulationdetectioncnn_amd import synthetic_frames

    def timed(fn, fresh, k, w):
        ts = []
        for i in range(w + k):
            a = fresh()
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn(a)
            torch.cuda.synchronize()
            if i >= w:
                ts.append(time.perf_counter() - t)
            del a
        return sum(ts) / len(ts)

    rows = []
    for name, n in (("vtcnn2-c11-bf16-n2^20", 1 << 20), ("deployed3-f32-n2^20", 1 << 20)):
        m, _, _ = make_model(name, device)
        x = synthetic_frames(n, seed=2016)                      # numpy, pageable
        t_host = timed(lambda a: m.predict_host(a), x.copy, steps, warmup)

        def plain(a):
            p, l, _ = m.forward_device(torch.from_numpy(a).to(f"cuda:{device}"))
            return p.cpu(), l.cpu()
        t_plain = timed(plain, x.copy, steps, warmup)
        xp = torch.from_numpy(x).pin_memory().numpy()
        t_pinned = timed(lambda a: m.predict_host(a), lambda: xp, steps, warmup)
        row = {"workload": name, "frames": n, "streaming_driver_frames_per_s": n / t_host, "ms": t_host * 1e3,
               "input_gb_per_s": n * 1024 / t_host / 1e9, "caller_pinned_frames_per_s": n / t_pinned,
               "unpipelined_frames_per_s": n / t_plain}
        if name.startswith("deployed3"):
            iq = np.random.default_rng(1).integers(0, 256, size=(1 << 22) * 256, dtype=np.uint8)
            t_iq = timed(lambda a: m.predict_iq_u8(a, 0.02 / 127.5), iq.copy, steps, warmup)
            row["raw_iq_u8_n2^22_frames_per_s"] = (1 << 22) / t_iq
            del iq
        rows.append(row)
        del m, x, xp
        torch.cuda.empty_cache()
    return {"workload": "host-resident input and output (PCIe-inclusive; mdc_predict_host), fresh pageable numpy arrays unless noted",
            "unit": "frames/s", "rows": rows}


def metric_name(kind, filters, n):
    """BASELINE.json's metric string for the workloads it is quoted on (VT-CNN2 at batch 2^20); every other workload
    or batch size says what it is."""
    if kind == "vtcnn2":
        return "I/Q frames/sec (2x128, VT-CNN2, batch=2^20)" if n == 1 << 20 else f"I/Q frames/sec (2x128, VT-CNN2, batch={n})"
    what = "cnn.py literal model" if kind == "cnnpy" else f"deployed {filters}-filter net"
    return f"I/Q frames/sec (2x128, {what}, batch={n})"


def select_device(device):
    import torch
    torch.cuda.set_device(device)


def rank_environment(env):
    """What EVERY rank needs in its environment before it touches the GPU, whoever launched it (this file's own launcher
    or the driver's torch.distributed.run): dmabuf IPC -- the host driver supports no other, and without it RCCL across
    processes fails with `hipIpcGetMemHandle: invalid argument`."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def init_distributed(device, backend=None):
    """One process per GPU: "nccl" IS RCCL on ROCm.  The rank's GPU is bound explicitly (device_id) so that RCCL builds its
    communicator on THAT device at once instead of guessing from the first collective's tensor; gloo (the CPU rehearsal)
    takes no device."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or os.environ.get("MDC_BENCH_BACKEND", "nccl")
    kw = {"device_id": torch.device("cuda", device)} if backend == "nccl" else {}
    dist.init_process_group(backend, **kw)
    return dist


def self_launch(ngpu, argv, script):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the one-process-per-GPU job ourselves --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <argv>`
    -- as a CHILD process (never exec: nothing here has touched the GPU, and nothing will in this parent), pass rank 0's
    JSON line through on stdout and return the child's exit code.  Works the same under the driver's own launcher,
    which sets WORLD_SIZE and never reaches this function."""
    import signal
    import subprocess
    env = rank_environment(dict(os.environ))
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher's own c10d store on a free port IT picks (no bind-then-close window for another process to
    # take the port in); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={ngpu}", script] + list(argv)
    print(f"[bench] --gpus {ngpu} without a launcher: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

    def stop_child(grace=10.0):
        """The launcher and its ranks form their own session: signal the whole group, so no rank outlives us holding a GPU."""
        if child.poll() is not None:
            return
        for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(child.pid, sig)
            except (ProcessLookupError, PermissionError):
                return
            try:
                child.wait(timeout=wait)
                return
            except subprocess.TimeoutExpired:
                continue

    def on_signal(signum, _frame):      # a driver's timeout (SIGTERM) or ^C reaches the ranks too
        stop_child()
        sys.exit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    lines = 0
    try:
        for line in child.stdout:       # (the ranks' stderr goes straight through; their stdout carries rank 0's one line
            if line.lstrip().startswith("{"):                   # and whatever a backend chats there, e.g. gloo's "[Gloo] Rank 0
                lines += 1                                      # is connected ...": that goes to OUR stderr, so stdout is the line)
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        rc = child.wait()
    finally:
        stop_child()
        for sg, h in old.items():
            signal.signal(sg, h)
    if rc == 0 and lines != 1:
        print(f"[bench] the {ngpu}-rank job printed {lines} JSON lines (expected 1)", file=sys.stderr)
        return 3
    return rc


def compact_legs(out):
    """{workload: [frames/s, roofline fraction]} of the headline and every extra leg that has a value -- a few hundred
    bytes at the END of the line, so the per-config numbers survive a log that keeps only the tail."""
    legs = {out["config"]["workload"]: [round(out["value"]), round(out["roofline"]["frac"], 4) if out.get("roofline") else None]}
    for e in out.get("extra", []):
        if isinstance(e, dict) and "value" in e and "workload" in e:
            key = e["workload"] + (f" [{e['scaling']} x{e['n_gpus']}]" if "scaling" in e else "")
            legs[key] = [round(e["value"]), round(e["roofline"]["frac"], 4) if isinstance(e.get("roofline"), dict) else None]
    return legs


def main(argv=None, script=None):
    """script: the file the self-launched ranks run (default: this file; a wrapper that stubs the engine names itself)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child passes (use the committed profile's figure)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    rank_environment(os.environ)      # every rank, launched by whomever, BEFORE torch is imported or the GPU touched

    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and not launched:
        # however this file was invoked, --gpus N means N ranks: become the launcher's parent (BEFORE torch is imported)
        return self_launch(args.gpus, sys.argv[1:] if argv is None else argv, script or os.path.abspath(__file__))
    if args.gpus != world:
        # never measure a different job than the one asked for and label it with a stderr note
        print(f"[bench] --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: refusing to run", file=sys.stderr)
        return 2

    if world == 1 and not args.no_live_traffic and not args.no_extras:
        # roofline.traffic of the headline's kernels, measured in this run -- child processes, BEFORE this one touches the GPU
        # (a process that has initialised the GPU must not start another program on this pool)
        live_traffic(args.workload)

    import torch
    dist = None
    if world > 1:
        # MDC_BENCH_BACKEND=gloo / MDC_BENCH_ONE_DEVICE=1 exist only to rehearse the N>1 control flow on a one-GPU box (all
        # ranks share cuda:0); the driver's scaling run uses neither.
        device = 0 if os.environ.get("MDC_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
        select_device(device)
        dist = init_distributed(device)
    else:
        device = 0
        select_device(device)
    rank = dist.get_rank() if dist else 0
    ngpu = world if dist else 1

    name = args.workload
    kind, filters, classes, dtype, _, weights = WORKLOADS[name]
    m, x, probs, labels, n, el = run_workload(name, device, args.steps, args.warmup, dist)
    total_frames = n * ngpu * args.steps
    out = {
        "metric": metric_name(kind, filters, n),
        "value": total_frames / el, "unit": "frames/s",
        "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"f32": "f32", "bf16": "bf16", "fp8": "fp8", "f16": "f16"}[dtype],
        "data": "synthetic N(0,5e-3) f32 frames resident in HBM; " + weights,
        "config": {"workload": name, "topology": kind, "classes": classes, "frames_per_gpu": n,
                   "global_batch": n * ngpu, "parallelism": f"batch-shard x{ngpu} (no collective)",
                   "outputs": "softmax probabilities f32 + argmax int32",
                   "frames_per_forward_call": launch_chunk(kind, dtype, n) or "library default"},
    }
    if rank == 0:
        rl, kernels = dominant_roofline(m, x, probs, labels, max(2, min(args.steps, 5)))
        out["roofline"] = rl
        out["kernels"] = kernels
    del m, x, probs, labels
    torch.cuda.empty_cache()
    if dist and ngpu > 1:      # every rank runs these (barriers inside timed_region); rank 0 reports
        legs = []
        for label, scaling, wl, frames, global_batch in multi_gpu_legs(name, ngpu, rank):
            lsteps, lwarm = max(1, min(args.steps, 5)), min(args.warmup, 2)
            try:
                lm, lx, lp, ll, ln, lel = run_workload(wl, device, lsteps, lwarm, dist, frames=frames)
                legs.append({"workload": wl, "reading": label, "scaling": scaling, "n_gpus": ngpu, "global_batch": global_batch,
                             "frames_per_gpu": global_batch // ngpu, "value": global_batch * lsteps / lel, "unit": "frames/s",
                             "ms_per_step": lel / lsteps * 1e3, "steps": lsteps, "warmup": lwarm, "dtype": WORKLOADS[wl][3]})
                del lm, lx, lp, ll
                torch.cuda.empty_cache()
            except LegSkipped as e:     # agreed on by every rank BEFORE the leg's collectives (run_workload): all skip together
                legs.append({"workload": wl, "reading": label, "error": str(e)})
        if rank == 0:
            out["extra"] = legs
    if rank == 0 and ngpu == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name)
        if not args.no_extras:
            extras = []
            for en in EXTRAS:
                if en == name:
                    continue
                try:
                    em, ex, ep, elab, en_n, eel = run_workload(en, device, 5, 2)
                    # an extra leg is the MEDIAN of three 5-step regions: a box now and then stalls one launch for tens of
                    # milliseconds (seen on the training leg and, once, here: 25.0 ms per step around 18.8 ms of kernels), and a
                    # 5-step total has nothing to absorb that with.  The headline above keeps the contract's single K-step region.
                    from modulationdetectioncnn_amd.sharding import timed_region
                    more = [timed_region(lambda: em.forward_device(ex, probs=ep, labels=elab, batch_size=em.bench_chunk), 5, 0,
                                         sync=torch.cuda.synchronize, device=ex.device) for _ in range(2)]
                    eel = sorted([eel] + more)[1]
                    erl, _ = dominant_roofline(em, ex, ep, elab, 3)
                    extras.append({"workload": en, "value": en_n * 5 / eel, "unit": "frames/s", "ms_per_step": eel / 5 * 1e3,
                                   "timing": "median of three 5-step regions", "dtype": WORKLOADS[en][3], "roofline": erl})
                    del em, ex, ep, elab
                    torch.cuda.empty_cache()
                except Exception as e:     # an extra leg never hides the headline
                    extras.append({"workload": en, "error": repr(e)})
            for filters in (3, 10):
                try:
                    extras.append(run_iq_u8(filters, device))
                    torch.cuda.empty_cache()
                except Exception as e:
                    extras.append({"workload": f"deployed{filters}-iq-u8-n2^22", "error": repr(e)})
                try:
                    extras.append(run_q612(filters, device))
                    torch.cuda.empty_cache()
                except Exception as e:
                    extras.append({"workload": f"deployed{filters}-q612-n2^20", "error": repr(e)})
            try:      # one window at a time, as the reference's deployment runs (tools/latency.py)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import latency
                extras.append({"workload": "latency: one forward call, frames resident in HBM, enqueue -> result on the device, median of 100",
                               "unit": "us", "rows": latency.rows(device, sizes=(1, 16, 64, 4096), reps=100)})
            except Exception as e:
                extras.append({"workload": "latency", "error": repr(e)})
            try:      # the training step at the reference's batch geometry
                extras.append(run_training(device))
                torch.cuda.empty_cache()
            except Exception as e:
                extras.append({"workload": "training step", "error": repr(e)})
            try:      # numpy in, numpy out: what cnn.py:198 hands over (never the headline value)
                extras.append(run_host_path(device))
            except Exception as e:
                extras.append({"workload": "host-resident input", "error": repr(e)})
            out["extra"] = out.get("extra", []) + extras
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the other BASELINE configs' numbers twice in compact form: inside `config` (a key every record keeps) and as the
        # line's LAST key (a record that keeps only the tail of stdout still shows them)
        legs = compact_legs(out)
        out["config"]["legs_frames_per_s_and_roofline_frac"] = legs
        out["legs"] = legs
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
