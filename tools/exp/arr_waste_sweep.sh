# full-row bids of a problem after which the row reduction asks the driver for fresh row caches (CYTO_ARR_WASTE), batches and single problems
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-waste}; mkdir -p $O
for w in default 500 5000 20000 1000000; do
  echo "== CYTO_ARR_WASTE=$w"
  if [ $w = default ]; then unset CYTO_ARR_WASTE; else export CYTO_ARR_WASTE=$w; fi
  timeout 300 python tools/batch_chunks_bench.py 16 5000 2>&1 | grep "rep=1" | cut -c1-150
  timeout 300 python tools/batch_chunks_bench.py 64 10000 2>&1 | grep "rep=1" | cut -c1-150
  timeout 300 python tools/wide_large.py t20000 c4s10000 --reps 2 2>&1 | grep "rep=1" | cut -c1-25,128-175
done 2>&1 | tee $O/sweep.log
