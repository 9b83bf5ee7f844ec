mkdir -p gpurun_out/r04aa
for u in 4 8 4 8; do
  export CYTO_BID_UNROLL=$u
  echo "== CYTO_BID_UNROLL=$u"
  timeout 200 python tools/wide_large.py t20000 c4s10000 u20000 --reps 3 2>&1 | grep -v "^    wide_arr" | grep "rep=[12]" | sed 's/colsol==golden [A-Za-z]* spot-level [A-Za-z]* total diff [^ ]* //' | cut -c1-150
  timeout 100 python tools/batch_chunks_bench.py 2>&1 | tail -3 | cut -c1-200
done > gpurun_out/r04aa/ab.log 2>&1
cat gpurun_out/r04aa/ab.log
