# correctness (check_wide) + timing (wide_large) + per-launch trace of the round kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=" $O/check_wide.log | cut -c1-200
timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 > $O/wide_large.log 2>&1; echo "rc=$?" >> $O/wide_large.log
grep "rep=2\|rc=" $O/wide_large.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 20000 50000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/t$n -o t$n -- python $R/tools/quick_lap_bench.py $n > $R/$O/t$n.log 2>&1
  f=$(find $R/$O/t$n -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round
done
