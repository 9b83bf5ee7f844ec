# round 6, third GPU call: where c3's wall goes beyond its kernels; the rebuild threshold of the row reduction re-measured (few-cell-type single
# problem, chunk batches); every certified-unique cross instance again with the float64 polish; the tests added since the second call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_lap_gpu.py -m gpu -q -k "polish or certificate or certified_unique or one_edge" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
timeout 600 python -m pytest tests/test_large_gpu.py -m gpu -q -k "cytolike" >> $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log | cut -c1-200
timeout 600 python tools/c3_walls.py > $O/c3_walls.log 2>&1; cat $O/c3_walls.log | cut -c1-250
for w in 100 400 1600 1000000; do
  echo "== CYTO_ARR_WASTE=$w"
  CYTO_ARR_WASTE=$w timeout 300 python tools/wide_large.py t20000 c4s10000 --reps 3 2>&1 | grep "rep=2" | cut -c1-330
  CYTO_ARR_WASTE=$w timeout 300 python tools/batch_chunks_bench.py 64 10000 2>&1 | grep "rep=1" | cut -c1-200
  CYTO_ARR_WASTE=$w timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep "rep=1" | cut -c1-200
done > $O/arr_waste.log 2>&1; cat $O/arr_waste.log
timeout 1500 python tools/cross_unique.py --opts polish=1 > $O/cross_unique_polish.log 2>&1; tail -2 $O/cross_unique_polish.log; grep "!=" $O/cross_unique_polish.log | head
