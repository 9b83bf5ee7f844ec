# randomised parity stress after the whole-chip one-edge searches and the pipelined loop over runs (a third of the instances have repeated rows or integer ties)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05af; mkdir -p $O
timeout 400 python tools/stress_lap.py 4000 360 100 3000 > $O/s_small.log 2>&1
timeout 500 python tools/stress_lap.py 4400 42 > $O/s_default.log 2>&1
timeout 300 python tools/stress_lap.py 4500 60 1000 4000 --rebuild 2 > $O/s_rebuild.log 2>&1
for f in $O/s_*.log; do echo "$f: $(tail -1 $f)"; done
grep -h MISMATCH $O/s_*.log | head
