cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
timeout 300 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=\|fault" $O/check_wide.log | cut -c1-300 | head
grep "c3-\|dups" $O/check_wide.log | cut -c1-60,200-330
timeout 600 python tools/wide_large.py c3s50000 c4s10000 u20000 --reps 3 > $O/wide_large.log 2>&1; echo "rc=$?" >> $O/wide_large.log
grep "rep=2\|rc=" $O/wide_large.log | cut -c1-25,38-70,128-175
grep -A1 "c3s50000 mode=2 rep=2" $O/wide_large.log | tail -1 | cut -c1-250
timeout 1200 python -m pytest tests/test_lap_gpu.py -x -q -m gpu > $O/lap_tests.log 2>&1; echo "rc=$?" >> $O/lap_tests.log
tail -3 $O/lap_tests.log
