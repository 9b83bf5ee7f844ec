# round 6: the whole -m gpu suite, smoke(), the default bench line, the launcher forms of bench.py, randomised stress (the one-edge claims without host
# synchronisation, pooled pinned blocks, pooled streams) -- at the state with the stream pool and the larger block cache
TAG=r06g
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-extras --no-cpu-baseline --n 20000 > $O/bench_torchrun2.json 2> $O/bench_torchrun2.err; echo "torchrun bench rc=$?"
grep -o '"n_gpus": [0-9]*' $O/bench_torchrun2.json
timeout 900 python tools/stress_lap.py 0 90 100 3000 > $O/stress_a.log 2>&1; tail -1 $O/stress_a.log
timeout 900 python tools/stress_lap.py 300 36 5200 8200 > $O/stress_b.log 2>&1; tail -1 $O/stress_b.log
timeout 600 python tools/stress_lap.py 500 48 200 3000 --rebuild 2 > $O/stress_c.log 2>&1; tail -1 $O/stress_c.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06g/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
    v = d.get(k)
    print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "wall_ms_incl_h2d", "seconds") if a in v}, v.get("roofline", {}).get("frac"), v.get("float32_counts", {}).get("wall_ms_incl_h2d", ""), v.get("counts_resident_in_hbm", {}).get("wall_ms", ""))
PY
