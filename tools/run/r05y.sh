# randomised parity stress of the round's last state (one-launch rounds, speculative bids, price arrays, wipes, whole-chip one-edge searches)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
( timeout 900 python tools/stress_lap.py 3000 72 > $O/s_default.log 2>&1
  timeout 600 python tools/stress_lap.py 3100 120 200 3000 > $O/s_small.log 2>&1
  timeout 900 python tools/stress_lap.py 3300 24 9000 14000 > $O/s_large.log 2>&1
  timeout 600 python tools/stress_lap.py 3400 48 --wipe 2 --rounds 30 > $O/s_wipe_budget.log 2>&1
  timeout 600 python tools/stress_lap.py 3500 48 --par 4 --wipe 1 > $O/s_par_wipe1.log 2>&1
  timeout 600 python tools/stress_lap.py 3600 36 --rebuild 3 > $O/s_rebuild.log 2>&1 )
for f in $O/s_*.log; do echo "$f: $(tail -1 $f)"; grep -c MISMATCH $f; done
grep -h MISMATCH $O/s_*.log | head
