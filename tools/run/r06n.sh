# round 6: randomised stress at the round's last code state (pooled streams / pinned blocks / one work block per problem; the claims driver without host sync)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python tools/stress_lap.py 1000 120 100 3000 > $O/a.log 2>&1; tail -1 $O/a.log
timeout 900 python tools/stress_lap.py 1200 60 200 3000 --par 4 --wipe 1 > $O/b.log 2>&1; tail -1 $O/b.log
timeout 900 python tools/stress_lap.py 1300 60 200 3000 --wipe 2 --rounds 30 > $O/c.log 2>&1; tail -1 $O/c.log
timeout 900 python tools/stress_lap.py 1400 60 200 3000 --rebuild 3 > $O/d.log 2>&1; tail -1 $O/d.log
timeout 1200 python tools/stress_lap.py 1500 30 9000 14000 > $O/e.log 2>&1; tail -1 $O/e.log
timeout 900 python tools/stress_lap.py 1600 36 300 3000 --chain > $O/f.log 2>&1; tail -1 $O/f.log
timeout 900 python tools/stress_lap.py 1700 24 300 2500 --f64 > $O/g.log 2>&1; tail -1 $O/g.log
grep -h MISMATCH $O/*.log | head
