#!/usr/bin/env python3
"""ISA lint for a hazard hipcc does not model in the asm-sequenced conv kernel (vtcnn2_bf16_sched.hip).

(a) PROVEN on this kernel: hipcc implements `acc = bias` (AGPR to AGPR) with v_accvgpr_mov and may sink such a move
right in front of an asm MFMA that reads the AGPR as its C operand; it pads that VALU-write -> MFMA-read hazard for
its own MFMAs only, and the asm MFMA read garbage in exactly that register.

(b) PRECAUTION: a ds_write demonstrably reads its data registers some time AFTER it issues (an MFMA overwriting them
meanwhile corrupts the stored value), and an MFMA that writes VGPRs lands its result long after IT issues.  When
the register allocator hands the (dead) data or address registers of a global_store / ds_write to a following
VGPR-writing MFMA, a late operand read would send the store out with the MFMA's bits -- in the address, too.  The
pattern appeared once in this kernel's ISA (feature store followed by the first conv1 MFMA); whether a global_store
reads late was not established, so the kernel keeps the two apart and this lint enforces it.

The lint disassembles the kernel and reports every store-like instruction whose VGPR sources overlap the
destination of a VGPR-writing MFMA issued within WINDOW instructions after it, unless an s_waitcnt that covers the
memory instruction (lgkmcnt(0) for DS) sits between them.  Exit status 1 if any is found.
usage: tools/lint_async_hazards.py [file.hip] [kernel-name-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 8


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernel_isa(src, kernel):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc"), "-c", src, "-o", os.path.join(d, "k.o"), "-save-temps"]
        subprocess.run(cmd, cwd=d, check=True, capture_output=True)
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(d, asm)).read().split("\n")
    out, inside = [], False
    for ln in lines:
        if re.match(r"^_Z\w*%s\w*:" % kernel, ln):
            inside = True
            continue
        if inside:
            code = ln.split(";")[0].strip()
            if code.startswith("s_endpgm"):
                break
            if code and not code.startswith(".") and not code.endswith(":"):
                out.append(code)
    return out


def lint(isa):
    found = []
    # (b) a VALU write (hipcc pads v_accvgpr_mov/write -> MFMA only for its own MFMAs, not for asm ones) of a register
    #     that an MFMA reads fewer than two wait states later (an instruction = 1, s_nop N = N + 1)
    for i, ins in enumerate(isa):
        if ins.startswith(("v_accvgpr_mov", "v_accvgpr_write")):
            dst = ins.split()[1].rstrip(",")
            waits = 0
            for j in range(i + 1, min(i + 3, len(isa))):
                if isa[j].startswith("v_mfma") and re.search(r"\ba\[(\d+):(\d+)\]", isa[j]):
                    n = int(dst[1:])
                    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", isa[j].split(",", 1)[1]):
                        if int(m.group(1)) <= n <= int(m.group(2)):
                            found.append((i, ins, isa[j]))
                m = re.match(r"s_nop (\d+)", isa[j])
                waits += int(m.group(1)) + 1 if m else 1
                if waits >= 2:
                    break
    # (a) store-like instruction vs a following VGPR-writing MFMA
    for i, ins in enumerate(isa):
        if ins.startswith(("global_store", "ds_write", "buffer_store", "scratch_store")):
            src = vregs(ins)
        else:
            continue
        for j in range(i + 1, min(i + 1 + WINDOW, len(isa))):
            nxt = isa[j]
            if ins.startswith("ds_") and nxt.startswith("s_waitcnt") and "lgkmcnt(0)" in nxt:
                break
            if nxt.startswith("v_mfma") and nxt.split()[1].startswith("v"):
                if vregs(nxt.split(",")[0]) & src:
                    found.append((i, ins, nxt))
    return found


def main():
    src = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc", "vtcnn2_bf16_sched.hip")
    kernel = sys.argv[2] if len(sys.argv) > 2 else "vt_conv_bf16_sched_kernelILi0ELb0ELb0"
    isa = kernel_isa(src, kernel)
    found = lint(isa)
    n_mfma_v = sum(1 for x in isa if x.startswith("v_mfma") and x.split()[1].startswith("v"))
    print(f"{len(isa)} instructions, {n_mfma_v} VGPR-writing MFMAs, {len(found)} hazards")
    for i, a, b in found:
        print(f"  [{i}] {a}\n        -> {b}")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
