mkdir -p gpurun_out/r04ad
timeout 900 python -m pytest tests/test_large_gpu.py tests/test_cost_gpu.py -x -q -k "c4 or c5 or chunk or gv11 or ctx or context or c1_plugin" > gpurun_out/r04ad/tests.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r04ad/bench.json 2> gpurun_out/r04ad/bench.err
tail -3 gpurun_out/r04ad/tests.log
