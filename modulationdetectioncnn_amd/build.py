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
VERSION_SCRIPT = os.path.join(CSRC, "libmdc.map")
# Two libraries come out of the same sources:
#   "product"     libmdc.so      one kernel per role; nothing under mdc_forward* reads the environment
#   "alternates"  libmdc_alt.so  -DMDC_ALTERNATES: additionally the measured-slower alternate kernels that the GPU suite
#                                uses as bit-identity / race screens (the hipcc-scheduled bf16 conv, the one-barrier
#                                dense1, the deployed nets' f32-MFMA dense layer, the head as its own launch), selected
#                                per model by environment variables that mdc_create reads once
VARIANTS = {"product": (OBJ, LIB, ()),
            "alternates": (os.path.join(HERE, "csrc", "build_alt"), os.path.join(HERE, "libmdc_alt.so"), ("-DMDC_ALTERNATES",))}

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"),
            "-I", CSRC, "-Wall", "-Wno-unused-function", "-pthread",
            # only the MDC_API entry points of include/mdc.h are dynamic symbols (tests/test_cabi.py: nm -D == the header)
            "-fvisibility=hidden", "-fvisibility-inlines-hidden"]
# (no -fno-exceptions: the C ABI catches std::bad_alloc & co. at the boundary and returns MDC_ENOMEM, mdc_api.hip)


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


def flag_key(extra_flags=()) -> str:
    return hashlib.sha256(" ".join([*CXXFLAGS, *extra_flags]).encode()).hexdigest()


def build(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = "product") -> str:
    obj_dir, lib_path, vflags = VARIANTS[variant]
    extra_flags = (*vflags, *extra_flags)
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, "flags.stamp")
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "mdc.h")]
    key = flag_key(extra_flags)
    if not (os.path.exists(stamp) and open(stamp).read().strip() == key):
        force = True        # objects (if any) were built with other flags
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            jobs.append((src, obj))
    for f in os.listdir(obj_dir):      # an object whose source is gone (a removed kernel file) does not linger in the tree
        if f.endswith(".o") and os.path.join(obj_dir, f) not in objs:
            os.remove(os.path.join(obj_dir, f))

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
        if os.path.exists(stamp):
            os.remove(stamp)        # a build interrupted half way must not look complete
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    with open(stamp, "w") as f:
        f.write(key + "\n")
    if jobs or not _newer(lib_path, objs + [VERSION_SCRIPT]):
        cmd = [hipcc, "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", f"-Wl,--version-script={VERSION_SCRIPT}", "-o", lib_path, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib_path


def build_all(force: bool = False, verbose: bool = False):
    """The product library and the alternates test build."""
    return [build(force=force, verbose=verbose, variant=v) for v in VARIANTS]


if __name__ == "__main__":
    for path in build_all(force="--force" in sys.argv, verbose="-v" in sys.argv):
        print(path)
