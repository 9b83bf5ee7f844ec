mkdir -p gpurun_out/r04z
for w in 8 20 8 20; do
  export CYTO_CACHE_WAVES=$w CYTO_CACHE_UNROLL=8
  echo "== CYTO_CACHE_WAVES=$w"
  timeout 200 python tools/wide_large.py u50000 t20000 u20000 c4s10000 --reps 3 2>&1 | grep -v "^    wide_arr" | grep "rep=[12]" | sed 's/colsol==golden [A-Za-z]* spot-level [A-Za-z]* total diff [^ ]* //' | cut -c1-200
done > gpurun_out/r04z/ab.log 2>&1
cat gpurun_out/r04z/ab.log
