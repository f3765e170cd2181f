#!/usr/bin/env python3
"""Per-kernel resource metadata of a BUILT library (not of a recompile): every gfx950 code object inside libmdc.so's
.hip_fatbin section (one clang offload bundle per source file) is unbundled and its AMDGPU metadata note read --
{kernel symbol: {"scratch": .private_segment_fixed_size, "vgpr": .vgpr_count, "agpr": .agpr_count, "sgpr": .sgpr_count,
"lds": .group_segment_fixed_size, "vgpr_spill": .vgpr_spill_count, "sgpr_spill": .sgpr_spill_count}}.
    python tools/kernel_meta.py [modulationdetectioncnn_amd/libmdc.so]      # prints the table
tests/test_isa_hazards.py holds every product kernel to scratch == 0."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = {"scratch": ".private_segment_fixed_size", "vgpr": ".vgpr_count", "agpr": ".agpr_count", "sgpr": ".sgpr_count",
          "lds": ".group_segment_fixed_size", "vgpr_spill": ".vgpr_spill_count", "sgpr_spill": ".sgpr_spill_count"}


def kernel_metadata(so_path):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(d, "copy.so")],
                       check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, f"bundle{i}.bin")
            with open(part, "wb") as fh:
                fh.write(blob[s: starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"co{i}.o")
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            if not os.path.getsize(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            # the metadata is YAML: one "- .agpr_count: N ... .name: sym ... " block per kernel
            for block in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
                block = ".agpr_count:" + block
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                rec = {}
                for key, field in FIELDS.items():
                    m = re.search(re.escape(field) + r":\s+(\d+)", block)
                    rec[key] = int(m.group(1)) if m else None
                out[name.group(1)] = rec
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "modulationdetectioncnn_amd", "libmdc.so")
    meta = kernel_metadata(path)
    for k in sorted(meta):
        r = meta[k]
        full = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", full)
        short = m.group(1) if m else full
        print(f"{short[-70:]:70s} scratch {r['scratch']:5d}  vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} "
              f"spills v{r['vgpr_spill']} s{r['sgpr_spill']}")
    print(len(meta), "kernels")
