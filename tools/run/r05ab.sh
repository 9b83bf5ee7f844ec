cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ab; mkdir -p $O
timeout 300 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=\|fault" $O/check_wide.log | cut -c1-300 | head
timeout 600 python tools/wide_large.py t20000 c4s10000 u20000 u50000 --reps 3 2>&1 | grep "rep=2" | cut -c1-75,128-175,250-330
timeout 300 python tools/batch_chunks_bench.py 64 10000 2>&1 | grep "rep=1" | cut -c1-150
timeout 300 python tools/batch_chunks_bench.py 16 5000 2>&1 | grep "rep=1" | cut -c1-150
timeout 900 python -m pytest tests/test_lap_gpu.py -x -q -m gpu -k "wide" > $O/wide_tests.log 2>&1; echo "rc=$?" >> $O/wide_tests.log
tail -3 $O/wide_tests.log
