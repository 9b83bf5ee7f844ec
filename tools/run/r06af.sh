# round 6, the LAST check (two parity tests more than r06w; product code unchanged): the whole -m gpu suite, smoke(), the default bench line (-> profiles/r06af_bench.json), bench.py under an external launcher and
# self-spawned with two logical ranks, the cross check with and without the polish
TAG=r06af
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -E "passed|failed" $O/gputest.log | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-extras --no-cpu-baseline > $O/bench_torchrun2.json 2> $O/bench_torchrun2.err; echo "torchrun bench rc=$?"
grep -o '"n_gpus": [0-9]*' $O/bench_torchrun2.json
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-cpu-baseline > $O/bench_spawn2.json 2> $O/bench_spawn2.err; echo "spawn bench rc=$?"
grep -o '"n_gpus": [0-9]*' $O/bench_spawn2.json; grep -o '"logical_ranks": "[^"]*"' $O/bench_spawn2.json | head -2
timeout 600 python tools/cross_unique.py > $O/cross_unique.log 2>&1; tail -1 $O/cross_unique.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06af/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
    v = d.get(k)
    print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "wall_ms_incl_h2d", "seconds", "counts_dtype") if a in v}, v.get("roofline", {}).get("frac"))
PY
