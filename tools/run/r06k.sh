cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06k/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
    v = d.get(k)
    print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "first_call_wall_s", "wall_ms_incl_h2d", "seconds", "first_pass_seconds_rank0") if a in v}, v.get("roofline", {}).get("frac"))
PY
