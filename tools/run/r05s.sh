cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64 -o b64 -- python $R/tools/batch_chunks_bench.py 64 10000 > $O/b64.log 2>&1
tail -3 $O/b64.log | cut -c1-160
s=$(find $O/b64 -name "*kernel_stats.csv" | head -1); head -14 $s | cut -c1-170
python $R/tools/trace_rounds.py $(find $O/b64 -name "*kernel_trace.csv" | head -1) wide_sc_round
