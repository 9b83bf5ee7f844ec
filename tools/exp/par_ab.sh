mkdir -p gpurun_out/r04al
for p in 8 16 24 32 16; do
  echo "== --par $p"
  timeout 200 python tools/wide_large.py u20000 u50000 t20000 --par $p --reps 3 2>&1 | grep "rep=2" | sed 's/colsol==golden [A-Za-z]* spot-level [A-Za-z]* total diff [^ ]* //' | sed 's/ | free.*par_batches/ par_batches/' | cut -c1-160
done > gpurun_out/r04al/ab.log 2>&1
cat gpurun_out/r04al/ab.log
