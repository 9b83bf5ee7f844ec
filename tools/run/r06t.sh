# round 6: rebuilds that leave healthy rows alone (CYTO_CACHE_KEEP = columns of a row's cache that must still lie below its floor for the
# rebuild to skip the row; 0 = every row, the behaviour so far): single problems and the 256-chunk batch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O; rm -f $O/ab.log
for keep in 0 2 4 8 16 32 0; do
  echo "== CYTO_CACHE_KEEP=$keep" >> $O/ab.log
  CYTO_CACHE_KEEP=$keep timeout 400 python tools/wide_large.py c4s10000 t20000 k5t20000 t30000 u20000 u50000 --reps 3 2>&1 | grep -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/relax=.*dense=/dense=/' -e 's/trivial=.*//' >> $O/ab.log
  CYTO_CACHE_KEEP=$keep timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep -E "rep=1|identical|rror" | cut -c1-230 >> $O/ab.log
done
cat $O/ab.log
