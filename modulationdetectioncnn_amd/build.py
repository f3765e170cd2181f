"""Build libmdc.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so
travels to the GPU box with the repo snapshot.  Objects are cached per source by
mtime AND by the exact flag set (a stamp file records the flags of the last build: a build
with other flags, e.g. a -DMDC_ABLATIONS probe build, never leaves its objects for a plain
build to pick up), so rebuilding after touching one kernel file takes seconds.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libmdc.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"),
            "-I", CSRC, "-Wall", "-Wno-unused-function", "-pthread"]
# (no -fno-exceptions: the C ABI catches std::bad_alloc & co. at the boundary and returns MDC_ENOMEM, mdc_api.hip)
STAMP = os.path.join(OBJ, "flags.stamp")


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmdc.so cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a: str, deps) -> bool:
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "mdc.h")]
    flag_key = hashlib.sha256(" ".join([*CXXFLAGS, *extra_flags]).encode()).hexdigest()
    if not (os.path.exists(STAMP) and open(STAMP).read().strip() == flag_key):
        force = True        # objects (if any) were built with other flags
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *CXXFLAGS, *extra_flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    if jobs:
        if os.path.exists(STAMP):
            os.remove(STAMP)        # a build interrupted half way must not look complete
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    with open(STAMP, "w") as f:
        f.write(flag_key + "\n")
    if jobs or not _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
