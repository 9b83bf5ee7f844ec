"""Build cost-GEMM variants (-D switches) into tools/libcytohip_gv_<name>.so and time each at c3 size (one subprocess per variant)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import build as B
VARIANTS = {"fold2": [], "fold1": ["-DGEMM_FOLD=1"]}      # name -> extra hipcc flags (GEMM_FOLD: k-tiles per fold)
names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
if "--run-one" in sys.argv:
    from cytospace_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", f"libcytohip_gv_{names[0]}.so")
    sys.argv = [sys.argv[0], "20000", "5000", "50000", "1"]      # unique rows only (what the context path writes)
    import tools.gemm_only  # noqa
    sys.exit(0)
for nm in names:
    lib = os.path.join(ROOT, "tools", f"libcytohip_gv_{nm}.so")
    if not os.path.exists(lib) or "--rebuild" in sys.argv:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + VARIANTS[nm] + ["-o", lib] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"])
if "--build-only" in sys.argv:
    sys.exit(0)
for nm in names:
    out = subprocess.run([sys.executable, os.path.abspath(__file__), nm, "--run-one"], capture_output=True, text=True, cwd=ROOT)
    print(nm, "|", (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
