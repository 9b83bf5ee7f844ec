mkdir -p gpurun_out/r04af
for g in 4096 2048 1024 512 256 4096 1024; do
  export CYTO_BID_GRID=$g
  echo "== CYTO_BID_GRID=$g"
  timeout 200 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 2>&1 | grep -v "^    wide_arr" | grep "rep=[2]" | sed 's/colsol==golden [A-Za-z]* spot-level [A-Za-z]* total diff [^ ]* //' | cut -c1-110
done > gpurun_out/r04af/ab.log 2>&1
cat gpurun_out/r04af/ab.log
