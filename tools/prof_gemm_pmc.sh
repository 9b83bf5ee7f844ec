#!/bin/bash
# MFMA utilisation of the cost GEMM: one PMC pass (SQ + GRBM counters), separate from any trace run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gemm_pmc; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o gemm -- python $R/tools/gemm_only.py > $OUT/gemm.log 2>&1
tail -2 $OUT/gemm.log
python - <<'PY'
import csv, os, collections
f = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gemm_pmc/gemm_counter_collection.csv"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "pearson_gemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: f"{x:.3e}" for c, x in m.items()})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print("MfmaUtil = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs * 1024 SIMDs) = %.1f %%  (GRBM_GUI_ACTIVE is summed over the 8 XCDs)" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
