cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout 300 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=\|fault" $O/check_wide.log | cut -c1-300 | head
timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 > $O/wide_large.log 2>&1; echo "rc=$?" >> $O/wide_large.log
grep "rep=2\|rc=" $O/wide_large.log | cut -c1-230
for g in 0 16 64; do echo "== CYTO_SC_STAY=$g"
CYTO_SC_STAY=$g timeout 600 python tools/wide_large.py u20000 u50000 --reps 3 2>&1 | grep "rep=2\|rc=" | cut -c1-25,128-175; done
for m in 128 256 1024; do echo "== CYTO_SC_STAY_MAX=$m"
CYTO_SC_STAY_MAX=$m timeout 600 python tools/wide_large.py u20000 u50000 --reps 3 2>&1 | grep "rep=2\|rc=" | cut -c1-25,128-175; done
timeout 900 python -m pytest tests/test_lap_gpu.py -x -q -m gpu -k "wide" > $O/wide_tests.log 2>&1; echo "rc=$?" >> $O/wide_tests.log
tail -3 $O/wide_tests.log
