# round 6: the full-row relaxations of a search round by the whole workgroup (coop_dense) -- parity suite, c3 timing, the deep-search stress instance
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_large_gpu.py -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "passed|failed" $O/gputest.log | tail -1
timeout 600 python tools/c3_walls.py > $O/c3_walls.log 2>&1; cat $O/c3_walls.log | cut -c1-250
timeout 900 python tools/stress_lap.py 1527 1 9000 14000 > $O/deep.log 2>&1; tail -2 $O/deep.log | cut -c1-250
timeout 900 python tools/stress_lap.py 2000 60 200 3000 --rebuild -1 > $O/s1.log 2>&1; tail -1 $O/s1.log
timeout 900 python tools/stress_lap.py 2100 36 200 3000 --par 5 > $O/s2.log 2>&1; tail -1 $O/s2.log
timeout 600 python tools/wide_large.py c3s50000 c4s10000 t20000 --reps 2 2>&1 | grep "rep=1" | cut -c1-200
