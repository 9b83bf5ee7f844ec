# round 6: the full-row sweep's loads in flight per thread once more (CYTO_BID_UNROLL 4 = default / 8 / 3), after the observation that the
# launches with a few hundred full-row bids are bound by one workgroup's memory-level parallelism
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O; rm -f $O/ab.log
for u in 4 8 3 4 8; do
  echo "== CYTO_BID_UNROLL=$u" >> $O/ab.log
  CYTO_BID_UNROLL=$u timeout 400 python tools/wide_large.py c4s10000 t20000 u20000 u50000 --reps 3 2>&1 | grep -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/ | free=.*//' >> $O/ab.log
  CYTO_BID_UNROLL=$u timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep -E "rep=1|rror" | cut -c1-150 >> $O/ab.log
done
cat $O/ab.log
