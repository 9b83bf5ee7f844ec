# round 6, second GPU call: the whole -m gpu suite on the certificate / integer count dtypes / wave-per-row full-row bids (t30000 and k5t20000's
# goldens are still being made: their two cases fail on purpose), the A/B of the wave-per-row threshold, every certified-unique cross
# instance, the default bench line, bench.py with two logical ranks (the c4 legs as threads)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -8 $O/gputest.log | cut -c1-200
bash tools/exp/wave_rows_ab.sh r06b/wave_rows > $O/wave_rows.log 2>&1; cat $O/wave_rows/ab.log | cut -c1-200
timeout 1200 python tools/cross_unique.py > $O/cross_unique.log 2>&1; tail -2 $O/cross_unique.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-cpu-baseline --n 20000 > $O/bench_spawn2.json 2> $O/bench_spawn2.err; echo "spawn bench rc=$?"
python - <<'PY'
import json
for f in ("bench.json", "bench_spawn2.json"):
    try:
        d = json.load(open("gpurun_out/r06b/" + f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"])
    for k in ("c2", "c2_batch", "c2_cytolike", "c3", "c4_chunks", "c5_chunks", "c4_strong", "c4_sharded"):
        v = d.get(k)
        if isinstance(v, dict):
            print("  ", k, {a: v[a] for a in ("ms_per_solve", "wall_s", "wall_ms_incl_h2d", "seconds", "full_row_bids", "logical_ranks", "error") if a in v},
                  v.get("roofline", {}).get("frac"), v.get("float32_counts", ""), v.get("kernel_ms", ""))
PY
