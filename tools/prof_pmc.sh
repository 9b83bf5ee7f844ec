#!/bin/bash
# PMC pass (separate from any trace run, as MI355X_MICROARCH.md prescribes): HBM bytes of the kernels of one solve
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python $GRAFT_REPO_ROOT/tools/quick_lap_bench.py 20000 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python $GRAFT_REPO_ROOT/tools/quick_lap_bench.py 20000 > $OUT/write.log 2>&1
ls $OUT
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc"
for tag in ("fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "?")[:60]
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0))
        for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print(tag, f"{k:60s} dispatches={c} sum={v:.1f} (KB units per rocprofv3) per-dispatch={v/c:.1f}")
PY
