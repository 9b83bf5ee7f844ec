cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 > $O/wide_large.log 2>&1; echo "rc=$?" >> $O/wide_large.log
grep "rep=2\|rc=" $O/wide_large.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in t20000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$tag -o $tag -- python $R/tools/wide_large.py $tag --reps 2 > $R/$O/$tag.log 2>&1
  f=$(find $R/$O/$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round
done
