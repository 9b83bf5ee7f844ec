mkdir -p gpurun_out/r04ap
for st in 0 1; do
  export CYTO_CACHE_STREAM=$st
  echo "== CYTO_CACHE_STREAM=$st"
  timeout 60 python tools/wide_large.py u20000 u50000 t20000 --reps 2 2>&1 | grep "rep=1" | sed 's/total diff [^ ]* //' | sed 's/ | free.*dense=/ dense=/' | sed 's/colsol==golden [A-Za-z]* spot-level [A-Za-z]* //' | cut -c1-118
done > gpurun_out/r04ap/ab.log 2>&1
cat gpurun_out/r04ap/ab.log
unset CYTO_CACHE_STREAM
timeout 60 python -m pytest tests/test_lap_gpu.py -x -q -k "row_cache_builders_agree" > gpurun_out/r04ap/t.log 2>&1; grep -E "passed|failed" gpurun_out/r04ap/t.log
