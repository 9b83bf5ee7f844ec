# round 6, fourth GPU call: c3 after the block cache keeps its 4-GB operand and the pipeline starts on a quarter block; the fused / context tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 600 python tools/c3_walls.py > $O/c3_walls.log 2>&1; cat $O/c3_walls.log | cut -c1-250
timeout 1200 python -m pytest tests/test_cost_gpu.py tests/test_large_gpu.py -m gpu -q -k "not t30000" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log | cut -c1-200
timeout 600 python -m pytest tests/test_lap_gpu.py -m gpu -q -k "polish or near_tie" >> $O/gputest.log 2>&1; tail -2 $O/gputest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06d/bench.json"))
print("headline", d["value"], d["ms_per_step"])
for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
    v = d.get(k)
    print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "wall_ms_incl_h2d", "seconds") if a in v}, v.get("roofline", {}).get("frac"), v.get("float32_counts", ""), v.get("counts_resident_in_hbm", {}).get("wall_ms", ""))
PY
