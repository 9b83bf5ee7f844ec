#!/bin/bash
# A variant build of the library for same-box A/B runs: tools/build_variant.sh <name> [extra hipcc flags for lap_wide.hip ...]
#   -> cytospace_amd/build/libcytohip_<name>.so (git-ignored, travels to the GPU box); run with CYTOHIP_LIB=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_<name>.so
# e.g.  tools/build_variant.sh coop0 -DCYTO_COOP_MIN_N=0      (tools/run/r06r.sh, r06z.sh)
#       tools/build_variant.sh split -DCYTO_AUG_FIN_SPLIT     (tools/run/r06u.sh)
# Only lap_wide.hip is recompiled (the other objects come from the last `python -m cytospace_amd.build`).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
python -m cytospace_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c -o $R/cytospace_amd/build/lap_wide_$name.o $R/cytospace_amd/csrc/lap_wide.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cytospace_amd/build/libcytohip_$name.so $R/cytospace_amd/build/{core,lap_jv,cost,batch,comm}.o $R/cytospace_amd/build/lap_wide_$name.o -L/opt/rocm/lib -lrccl -lpthread
ls -la $R/cytospace_amd/build/libcytohip_$name.so
