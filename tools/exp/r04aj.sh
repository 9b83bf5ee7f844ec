mkdir -p gpurun_out/r04aj
timeout 600 python -m pytest tests/test_lap_gpu.py tests/test_cost_gpu.py -x -q -k "batch or chunk or ctx or context or gv11 or threads" > gpurun_out/r04aj/tests.log 2>&1; tail -2 gpurun_out/r04aj/tests.log
timeout 500 python -m pytest tests/test_large_gpu.py -x -q -k "c4 or c5" > gpurun_out/r04aj/tests2.log 2>&1; tail -2 gpurun_out/r04aj/tests2.log
for K in 8 20 64; do timeout 100 python tools/batch_chunks_bench.py $K 2>&1 | tail -2 | head -1 | cut -c1-120; done
timeout 400 python bench.py > gpurun_out/r04aj/bench.json 2> gpurun_out/r04aj/bench.err
