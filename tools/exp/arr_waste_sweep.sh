mkdir -p gpurun_out/r04w
for w in 0 50 25 12 6; do
  if [ $w = 0 ]; then unset CYTO_ARR_WASTE; else export CYTO_ARR_WASTE=$w; fi
  echo "== CYTO_ARR_WASTE=$w"
  timeout 200 python tools/wide_large.py t20000 c4s10000 u20000 --reps 2 2>&1 | grep -v "^    wide_arr" | grep "rep=1"
done > gpurun_out/r04w/arr_waste.log 2>&1
cat gpurun_out/r04w/arr_waste.log | cut -c1-330
