# round 6, first GPU call: the whole -m gpu suite on the advisor fixes, the baseline bench line, and the evidence the batched legs never had:
# kernel stats + per-launch trace + PMC traffic of 256 (and 64) concurrent c4-chunk LAPs and of 50 c5 chunks; the knob A/B of round 5's batch settings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -3 $O/gputest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a
for K in 256 32; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4_K$K -o c4 -- python $R/tools/batch_chunks_bench.py $K 10000 > $O/prof_c4_K$K.log 2>&1
  f=$(find $O/prof_c4_K$K -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round > $O/rounds_c4_K$K.txt 2>&1
  python $R/tools/trace_overlap.py $f > $O/overlap_c4_K$K.txt 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python $R/tools/c5_chunks.py 10000 500 50 > $O/prof_c5.log 2>&1
f=$(find $O/prof_c5 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_overlap.py $f > $O/overlap_c5.txt 2>&1
mkdir -p $O/pmc_c4
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c4 -o fetch -- python $R/tools/batch_chunks_bench.py 256 10000 > $O/pmc_c4/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c4 -o write -- python $R/tools/batch_chunks_bench.py 256 10000 > $O/pmc_c4/write.log 2>&1
python $R/tools/pmc_to_json.py $O/pmc_c4 10000 $O/pmc_traffic_c4_K256.json "256 concurrent c4 sub-spot chunk LAPs of 10 000 cells (tools/batch_chunks_bench.py 256 10000: warm-up solve + 2 batched calls)" > $O/pmc_c4/json.log 2>&1
cd $R
python tools/wide_large.py t20000 c4s10000 c3s50000 --reps 2 > $O/wide_large.log 2>&1
python tools/c3_lap_breakdown.py > $O/c3_breakdown.log 2>&1
bash tools/exp/batch_knobs_ab.sh r06a/knobs > $O/knobs.log 2>&1
find $O -name "*.csv" -size +20M -exec rm {} \;
find $O -name "*.db" -delete
du -sh $O; tail -4 $O/prof_c4_K256.log | cut -c1-260; cat $O/overlap_c4_K256.txt | head -30
