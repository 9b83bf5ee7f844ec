cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a/gputest.log
timeout 600 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; echo "bench rc=$?" >> gpurun_out/r05a/bench.err
timeout 300 python bench.py --gpus 2 --oversubscribe --no-extras --no-cpu-baseline > gpurun_out/r05a/bench_g2.json 2> gpurun_out/r05a/bench_g2.err; echo "bench2 rc=$?" >> gpurun_out/r05a/bench_g2.err
tail -5 gpurun_out/r05a/gputest.log; cat gpurun_out/r05a/bench.json | cut -c1-600
