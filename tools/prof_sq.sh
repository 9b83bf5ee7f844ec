#!/bin/bash
# Instruction-mix counters of the two persistent kernels of one 20 000^2 solve (separate PMC passes; no trace domains besides --kernel-trace):
# what a search round / a bid is made of -- VALU, SALU, LDS, VMEM instructions, busy and wait cycles.  usage: prof_sq.sh [tag]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r03h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o $name -- python $GRAFT_REPO_ROOT/tools/quick_lap_bench.py 20000 > $OUT/$name.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections, json
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/sq_" + (os.environ.get("TAG") or "r03h")
res = collections.defaultdict(dict)
for f in glob.glob(out + "/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        if "wide_a" not in k: continue
        k = "wide_arr" if "wide_arr<" in k or "wide_arrI" in k else ("wide_aug" if "wide_aug" in k else k[:40])
        agg[(k, r.get("Counter_Name"))][0] += 1
        agg[(k, r.get("Counter_Name"))][1] += float(r.get("Counter_Value", 0))
    for (k, c), (cnt, v) in agg.items():
        res[k][c] = v / max(cnt, 1)
print(json.dumps(res, indent=1, sort_keys=True))
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
PY
