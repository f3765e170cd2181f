#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

  kernel stats : gpurun_out/prof_<tag>/runc/*_kernel_stats.csv      -> profiles/<round>_<tag>_kernel_stats.csv
  HBM traffic  : gpurun_out/pmc_fetch_*/, pmc_write_*/ (separate --pmc passes, as
                 MI355X_MICROARCH.md "HBM" prescribes) -> profiles/<round>_traffic.json
Corrections applied (same guide): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so the read side is
doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counters(d, kern):
    fs = glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv"))
    if not fs:
        return {}
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # gpurun merges new files next to old ones: newest run
        if kern in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


os.makedirs(P, exist_ok=True)
for d in sorted(glob.glob(os.path.join(G, "prof_*"))):
    tag = os.path.basename(d)[5:]
    if tag[:1] == "r" and tag[1:3].isdigit() and not tag.startswith(rnd):
        continue      # another round's scratch directory still lying in gpurun_out/
    for f in sorted(glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:      # newest run only
        shutil.copy(f, os.path.join(P, (tag if tag.startswith(rnd) else f"{rnd}_{tag}") + "_kernel_stats.csv"))

spec = [  # (json key, pmc dir suffix, kernel substring, frames per launch in that run)
    ("mdc_vt_conv/bf16", "vt", "vt_conv_bf16", 1 << 20),      # bench.py's headline: one launch per 2^20-frame step
    ("mdc_vt_dense1/bf16", "vt", "vt_dense1_bf16", 1 << 20),
    # (no mdc_vt_head row since round 3: the 16-bit batch path runs dense2 + softmax inside dense1's epilogue)
    ("mdc_vt_conv/fp8", "vtfp8", "vt_conv_fp8_kernel", 1 << 20),
    ("mdc_vt_dense1/fp8", "vtfp8", "vt_dense1_bf16", 1 << 20),
    ("mdc_vt_conv/f32", "vtf32", "vt_conv_f32_kernel", 1 << 16),
    ("mdc_vt_dense1/f32", "vtf32", "vt_dense1_f32_kernel", 1 << 16),
    ("mdc_deployed_fwd/F3", "dep", "deployed_fwd_kernel<3, 0, 0, false, false, 0>", 1 << 20),      # <F, TAP, ABL, TAIL, U8, RING>
    ("mdc_deployed_fwd/F10", "dep", "deployed_fwd_kernel<10, 0, 0, false, false, 0>", 1 << 20),
    # deployed_bf16_kernel<F, MODE, U8>: MODE 0 bf16, 1 f16, 2 fp8
    ("mdc_deployed_fwd/F3/bf16", "dep", "deployed_bf16_kernel<3, 0, false>", 1 << 20),
    ("mdc_deployed_fwd/F10/bf16", "dep", "deployed_bf16_kernel<10, 0, false>", 1 << 20),
    ("mdc_deployed_fwd/F3/f16", "dep", "deployed_bf16_kernel<3, 1, false>", 1 << 20),
    ("mdc_deployed_fwd/F10/f16", "dep", "deployed_bf16_kernel<10, 1, false>", 1 << 20),
    ("mdc_deployed_fwd/F3/fp8", "dep", "deployed_bf16_kernel<3, 2, false>", 1 << 20),
    ("mdc_deployed_fwd/F10/fp8", "dep", "deployed_bf16_kernel<10, 2, false>", 1 << 20),
    ("mdc_deployed_q612/F3", "dep", "deployed_q612_kernel<3>", 1 << 20),
    ("mdc_deployed_q612/F10", "dep", "deployed_q612_kernel<10>", 1 << 20),
    # raw uint8 I/Q input (256 B/frame): the U8 = true forms of the same kernels
    ("mdc_deployed_fwd/F3/u8", "dep", "deployed_fwd_kernel<3, 0, 0, false, true, 0>", 1 << 20),
    ("mdc_deployed_fwd/F10/u8", "dep", "deployed_fwd_kernel<10, 0, 0, false, true, 0>", 1 << 20),
    ("mdc_deployed_fwd/F3/bf16/u8", "dep", "deployed_bf16_kernel<3, 0, true>", 1 << 20),
    ("mdc_deployed_fwd/F10/bf16/u8", "dep", "deployed_bf16_kernel<10, 0, true>", 1 << 20),
    ("mdc_deployed_fwd/F3/f16/u8", "dep", "deployed_bf16_kernel<3, 1, true>", 1 << 20),
    ("mdc_deployed_fwd/F10/f16/u8", "dep", "deployed_bf16_kernel<10, 1, true>", 1 << 20),
]
out = {"note": "bytes per launch; FETCH_SIZE doubled (gfx950 wide-read correction: calibrated for 16 B/lane streaming reads; the /u8 "
               "kernels read 8 B per lane or 16 B per LDS-DMA lane, so their read side is the guide's 'uncalibrated width' case -- the "
               "raw counter is kept beside it), KiB -> bytes", "kernels": {}}
for key, sfx, kern, frames in spec:
    fe = counters("pmc_fetch_" + sfx, kern).get("FETCH_SIZE")
    wr = counters("pmc_write_" + sfx, kern).get("WRITE_SIZE")
    if fe is None or wr is None:
        continue
    rd_b, wr_b = 2.0 * fe * 1024.0, wr * 1024.0
    out["kernels"][key] = {"frames_per_launch": frames, "FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB_raw": wr,
                           "read_bytes": rd_b, "write_bytes": wr_b, "hbm_bytes": rd_b + wr_b,
                           "hbm_bytes_per_frame": (rd_b + wr_b) / frames}
json.dump(out, open(os.path.join(P, f"{rnd}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# MFMA-pipe utilisation of the bf16 VT-CNN2 kernels (own --pmc pass): busy = SQ_VALU_MFMA_BUSY_CYCLES /
# (4 SIMDs x 256 CUs x cycles), cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8.
mf = {}
SETS = (("mdc_vt_conv/bf16", "vt_conv_bf16", "vt"), ("mdc_vt_dense1/bf16", "vt_dense1_bf16", "vt"),
        ("mdc_vt_conv/fp8", "vt_conv_fp8_kernel", "vtfp8"), ("mdc_vt_dense1/fp8", "vt_dense1_bf16", "vtfp8"))
for key, kern, sfx in SETS:
    c = counters("pmc_mfma_" + sfx, kern)
    if not c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    mf[key] = {"GRBM_GUI_ACTIVE_sum": c["GRBM_GUI_ACTIVE"], "cycles_per_launch": cyc, "SQ_INSTS_MFMA": c["SQ_INSTS_MFMA"],
               "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
               "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc)}
for key, kern, sfx in SETS:
    c = counters("pmc_lds_" + sfx, kern)
    if c and key in mf:
        mf[key].update({"SQ_LDS_BANK_CONFLICT": c.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": c.get("SQ_LDS_IDX_ACTIVE"),
                        "lds_conflict_frac": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                        "SQ_WAIT_ANY": c.get("SQ_WAIT_ANY"), "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"),
                        "wave_parked_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None})
if mf:
    json.dump(mf, open(os.path.join(P, f"{rnd}_mfma.json"), "w"), indent=1)
    print(json.dumps(mf, indent=1))
