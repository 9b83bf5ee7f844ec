mkdir -p gpurun_out/r04x
timeout 200 python tools/cache_build_bench.py 20000 10000 > gpurun_out/r04x/cache.log 2>&1
timeout 120 python tools/cache_build_bench.py 20000 --typed >> gpurun_out/r04x/cache.log 2>&1
timeout 200 python tools/wide_large.py t20000 c4s10000 u20000 u50000 --reps 2 2>&1 | grep "rep=1" > gpurun_out/r04x/wide.log
timeout 600 python -m pytest tests/test_cost_gpu.py tests/test_lap_gpu.py -x -q > gpurun_out/r04x/tests.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r04x/bench.json 2> gpurun_out/r04x/bench.err
cat gpurun_out/r04x/cache.log gpurun_out/r04x/wide.log | cut -c1-300; tail -2 gpurun_out/r04x/tests.log
