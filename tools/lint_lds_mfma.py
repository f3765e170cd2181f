#!/usr/bin/env python3
"""Scan every kernel of the BUILT libmdc.so for MFMAs that wait out an LDS read issued just before them:
    ds_read* ; s_waitcnt lgkmcnt(0) ; v_mfma*
In a long chain that triple means the compiler re-used the operand registers and exposes one LDS round trip per group -- what
round 5 found in cnn.py's layer 1 (31 per 16-row tile; csrc/dense_chain.hip) and fixed by requesting the next operands before the
current MFMAs issue.  A few such triples in short tail layers are normal.
    python tools/lint_lds_mfma.py [path/to/libmdc.so]        -> kernels sorted by the count, with their MFMA totals"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def scan(so_path):
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(d, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(MAGIC, blob)]
        for n, i in enumerate(starts):
            part = os.path.join(d, f"b{n}.bin")
            open(part, "wb").write(blob[i:(starts[n + 1] if n + 1 < len(starts) else len(blob))])
            co = os.path.join(d, f"co{n}.o")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co):
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True).stdout
            cur, body = None, {}
            for ln in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
                if m:
                    cur = m.group(1)
                    body[cur] = []
                    continue
                mm = re.match(r"\s+(\S+)\s*(.*?)\s*//", ln)
                if cur and mm:
                    body[cur].append((mm.group(1), mm.group(2)))
            for name, ins in body.items():
                mf = sum(1 for o, _ in ins if o.startswith("v_mfma"))
                if mf:
                    serial = sum(1 for a, b, c in zip(ins, ins[1:], ins[2:]) if a[0].startswith("ds_read") and b[0] == "s_waitcnt"
                                 and "lgkmcnt(0)" in b[1] and c[0].startswith("v_mfma"))
                    out.append((serial, mf, name))
    return sorted(out, reverse=True)


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "modulationdetectioncnn_amd", "libmdc.so")
    for serial, mf, name in scan(so):
        if serial:
            print(f"{serial:4d} of {mf:5d} MFMAs  {name[:140]}")
    print("kernels with MFMAs:", len(scan(so)))
