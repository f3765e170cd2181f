#!/bin/bash
# ON THE GPU BOX: shader clock held by the asm-sequenced conv kernel and by its timing probes.
#   clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, from one rocprofv3 run each.
# usage: bash tools/clock_probe.sh 0 6 7      (probe numbers of tools/ablate_sched.py; needs -DMDC_ABLATIONS)
set -e -o pipefail
R=$PWD
python3 -c "import sys; sys.path.insert(0,'$R'); from modulationdetectioncnn_amd import build as b; b.build(force=True, extra_flags=['-DMDC_ABLATIONS'])" > /dev/null
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  rm -rf /tmp/clk_$a
  MDC_ABLATE_S=$a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/clk_$a -- python3 $R/tools/prof_conv.py > /tmp/clk_$a.log 2>&1
  python3 - "$a" <<'PY'
import csv, glob, sys
a = sys.argv[1]
f = glob.glob(f"/tmp/clk_{a}/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "vt_conv_bf16" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
cyc = [float(r["Counter_Value"]) / 8 for r in rows]
kt = glob.glob(f"/tmp/clk_{a}/*/*_kernel_trace.csv")[0]
dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "vt_conv_bf16" in r["Kernel_Name"]]
c = sum(cyc) / len(cyc); d = sum(dur) / len(dur)
print(f"probe {a}: {len(cyc)} launches, {c:.0f} cycles, {d/1e3:.1f} us -> {c/d:.3f} GHz", flush=True)
PY
done
