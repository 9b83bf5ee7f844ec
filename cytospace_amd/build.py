"""Builds libcytohip.so (HIP, gfx950 only) in-tree with hipcc.  No CPU fallback exists.

Every source is compiled to its own object (in parallel; an object is reused while it is newer than its source and
every header) and the objects are linked into the shared library."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcytohip.so")
SOURCES = ["core.hip", "lap_jv.hip", "lap_wide.hip", "cost.hip", "batch.hip", "comm.hip"]
# -ffp-contract=off: the JV kernels must evaluate exactly the subtract/compare sequence of the
# oracle (no FMA contraction, no re-association).  MFMA use in the cost kernels is explicit.
# (CYTO_EXTRA_FLAGS: developer builds, e.g. -DWIDE_STOP_CAP=128 for tools/exp -- use with --force and rebuild afterwards)
FLAGS = (["-DCYTO_WIDE_PROF"] if os.environ.get("CYTO_WIDE_PROF") else []) + os.environ.get("CYTO_EXTRA_FLAGS", "").split() + [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "cytohip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "cytohip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    tag = "_prof" if os.environ.get("CYTO_WIDE_PROF") else ""
    newest_header = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", tag + ".o"))
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), newest_header):
            return o
        cmd = [hipcc] + FLAGS + ["-c", "-o", o, s]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
