#!/usr/bin/env python3
"""gpurun_out/pmc_<round>_dep*_{A,B}/ (tools/collect_dep_counters.sh) -> profiles/<round>_dep_counters.json:
per deployed-net kernel, the mean per-launch SQ / LDS counters and the derived fractions
  valu_busy   = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-equivalent (quad-cycles: x4 / (SQ_WAVE_CYCLES / waves))  -- see below
  wave_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (waves sitting in s_waitcnt / barriers)
  issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     (waves with an instruction that cannot issue)
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}
for d in sorted(glob.glob(os.path.join(G, f"pmc_{rnd}_dep*_[AB]"))):
    form = "f32 MFMA-dense variant" if "depmfma" in d else "default"
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:      # newest run only
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "deployed" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            name = k.replace("void mdc::(anonymous namespace)::", "").split("(")[0]
            if form != "default" and "deployed_f32m_kernel" not in name:
                continue
            e = out.setdefault(name, {"vgpr": None})
            for c, v in cs.items():
                e[c] = sum(v) / len(v)
for name, e in out.items():
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        e["wave_parked_frac"] = e.get("SQ_WAIT_ANY", 0) / wc
        e["issue_stall_frac"] = e.get("SQ_WAIT_INST_ANY", 0) / wc
        e["inst_active_frac"] = e.get("SQ_ACTIVE_INST_ANY", 0) / wc
        e["valu_active_frac_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_VALU", 0) / wc
    if e.get("SQ_BUSY_CYCLES") and e.get("SQ_ACTIVE_INST_VALU") is not None:
        # SQ_BUSY_CYCLES: cycles summed over the chip's SQ instances (x-ref GRBM_GUI_ACTIVE); VALU-active quad-cycles x 4
        e["valu_insts_per_launch"] = e.get("SQ_INSTS_VALU")
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"]
    e.pop("vgpr", None)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_dep_counters.json"), "w"), indent=1, sort_keys=True)
for name, e in sorted(out.items()):
    print(f"{name[:70]:70s} parked {e.get('wave_parked_frac', float('nan')):.2f} stall {e.get('issue_stall_frac', float('nan')):.2f} "
          f"active {e.get('inst_active_frac', float('nan')):.2f} valu {e.get('valu_active_frac_of_wave_cycles', float('nan')):.2f} "
          f"VALU insts {e.get('SQ_INSTS_VALU', 0):.3g} MFMA {e.get('SQ_INSTS_MFMA', 0):.3g} GUI {e.get('GRBM_GUI_ACTIVE', 0):.3g}")
