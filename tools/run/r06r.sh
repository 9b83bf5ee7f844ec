# round 6: the state after r06q (full rows of n >= 16384 by the whole workgroup, shorter ones in the round): parity suite, the randomised
# stress on the product build AND on a build that takes every full row through the cooperative path (-DCYTO_COOP_MIN_N=0 via CYTOHIP_LIB),
# c3's walls, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_large_gpu.py -m gpu -x -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
C0=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_coop0.so
timeout 900 python tools/stress_lap.py 2000 60 200 3000 --rebuild -1 > $O/s1.log 2>&1; tail -1 $O/s1.log
CYTOHIP_LIB=$C0 timeout 900 python tools/stress_lap.py 2000 60 200 3000 --rebuild -1 > $O/s1_coop0.log 2>&1; tail -1 $O/s1_coop0.log
CYTOHIP_LIB=$C0 timeout 900 python tools/stress_lap.py 2100 36 200 3000 --par 5 > $O/s2_coop0.log 2>&1; tail -1 $O/s2_coop0.log
CYTOHIP_LIB=$C0 timeout 900 python tools/stress_lap.py 2200 40 3000 6000 > $O/s3_coop0.log 2>&1; tail -1 $O/s3_coop0.log
CYTOHIP_LIB=$C0 timeout 600 python -m pytest tests/test_lap_gpu.py -m gpu -x -q > $O/gputest_coop0.log 2>&1; grep -E "passed|failed" $O/gputest_coop0.log | tail -1
timeout 600 python tools/c3_walls.py > $O/c3_walls.log 2>&1; cat $O/c3_walls.log | cut -c1-250
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06r/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v.get("wall_s") or v.get("wall_ms")) for k,v in d.items() if isinstance(v,dict) and k.startswith("c") and k not in ("config","cpu_baseline")})
PY
