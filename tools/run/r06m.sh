cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout 1200 python -m pytest tests/test_lap_gpu.py -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; grep -E "passed|failed" $O/gputest.log | tail -1
timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep "rep=" | cut -c1-120
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06m/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
    v = d.get(k)
    print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "first_call_wall_s", "wall_ms_incl_h2d", "seconds", "first_pass_seconds_rank0") if a in v}, v.get("roofline", {}).get("frac"))
PY
