# full -m gpu suite + the default bench line at the state without the stay kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
true
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05p/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["roofline"]["frac"])
for k in ("c2","c2_cytolike","c3","c4_chunks","c5_chunks","c4_strong","c4_sharded"):
    v=d.get(k,{}); print(k, {kk:v[kk] for kk in v if kk in ("ms_per_solve","wall_s","seconds","wall_ms_incl_h2d","kernel_ms")})
PY
