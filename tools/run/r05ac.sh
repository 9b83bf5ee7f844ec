cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ac; mkdir -p $O
timeout 300 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=\|fault" $O/check_wide.log | cut -c1-300 | head
grep "ties \|dups " $O/check_wide.log | cut -c1-40,250-330
timeout 300 python tools/c3_lap_breakdown.py 2>&1 | tail -2 | cut -c1-700
timeout 600 python tools/wide_large.py c3s50000 u20000 --reps 2 2>&1 | grep "rep=1" | cut -c1-75,128-175
timeout 1500 python -m pytest tests/test_lap_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
