# the one-launch-per-round machine: parity of the wide tests, then true-size timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_lap_gpu.py -x -q -m gpu -k "wide" > $O/wide_tests.log 2>&1; echo "rc=$?" >> $O/wide_tests.log
tail -15 $O/wide_tests.log
timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 c3s50000 --reps 3 > $O/wide_large.log 2>&1; echo "rc=$?" >> $O/wide_large.log
tail -40 $O/wide_large.log
