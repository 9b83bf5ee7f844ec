cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout 300 python tools/c3_lap_breakdown.py > $O/c3_lap.log 2>&1; tail -12 $O/c3_lap.log | cut -c1-300
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05v/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["roofline"]["frac"])
for k in ("c2","c2_cytolike","c3","c4_chunks","c5_chunks","c4_strong","c4_sharded"):
    v=d.get(k,{}); print(k, {kk:v[kk] for kk in v if kk in ("ms_per_solve","wall_s","seconds","wall_ms_incl_h2d","kernel_ms","counts_resident_in_hbm")})
PY
