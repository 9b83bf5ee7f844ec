# kernel trace of one 20 000^2 / 50 000^2 solve: per-launch durations of the round kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
for n in 20000 50000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$n -o t$n -- python $R/tools/quick_lap_bench.py $n > $O/t$n.log 2>&1
  f=$(find $O/t$n -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round
  s=$(find $O/t$n -name "*kernel_stats.csv" | head -1); head -12 $s | cut -c1-150
  tail -2 $O/t$n.log | cut -c1-300
done
