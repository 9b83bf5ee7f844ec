# round 6: the one-edge searches in GROUPS (64 positions of the free list read at once, up to six runs' cache rows and prices in flight
# together) against the three-steps-deep pipeline before (cytospace_amd/build/libcytohip_pre.so via CYTOHIP_LIB): c3, the c3-shaped golden,
# the instances without one-edge searches (the search kernel's allocation), parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O; rm -f $O/ab.log
PRE=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_pre.so
for lib in HEAD pre HEAD pre; do
  echo "== lib=$lib" >> $O/ab.log
  L=$PRE; [ $lib = HEAD ] && L=
  CYTOHIP_LIB=$L timeout 400 python tools/wide_large.py c3s50000 c4s10000 t20000 u20000 u50000 --reps 3 2>&1 | grep -A1 -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/ | free=.*par_batches/ par_batches/' -e 's/wide_arr: .*| wide_aug/wide_aug/' | grep -v "^--" >> $O/ab.log
  CYTOHIP_LIB=$L timeout 400 python tools/c3_lap_breakdown.py 2>&1 | tail -2 | cut -c1-700 >> $O/ab.log
done
cat $O/ab.log
timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_large_gpu.py -m gpu -x -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
timeout 600 python tools/stress_lap.py 4000 60 200 3000 > $O/s1.log 2>&1; tail -1 $O/s1.log
