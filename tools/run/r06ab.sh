# round 6: full-row bids from an 8-bit copy of the cost rows, ONE pass with the cached second-best value as the bound (CYTO_WIDE_QUANT=1; the copy is built at a problem's first cache rebuild, the round kernel that reads it takes over from there)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O; rm -f $O/ab.log
for qv in 0 1 0 1; do
  echo "== CYTO_WIDE_QUANT=$qv" >> $O/ab.log
  CYTO_WIDE_QUANT=$qv timeout 400 python tools/wide_large.py c4s10000 t20000 k5t20000 u20000 u50000 --reps 3 2>&1 | grep -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/relax=.*dense=/dense=/' -e 's/trivial=.*//' >> $O/ab.log
  CYTO_WIDE_QUANT=$qv timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep -E "rep=1|identical|rror" | cut -c1-230 >> $O/ab.log
done
cat $O/ab.log
CYTO_WIDE_QUANT=1 timeout 900 python tools/stress_lap.py 2300 40 1100 4000 > $O/s1.log 2>&1; tail -2 $O/s1.log
