mkdir -p gpurun_out/r04y
timeout 900 python -m pytest tests/test_cost_gpu.py tests/test_large_gpu.py -x -q > gpurun_out/r04y/tests.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r04y/bench.json 2> gpurun_out/r04y/bench.err
tail -3 gpurun_out/r04y/tests.log
