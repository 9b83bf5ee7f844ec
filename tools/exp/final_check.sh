mkdir -p gpurun_out/r04ag
timeout 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04ag/gputest.log 2>&1
tail -3 gpurun_out/r04ag/gputest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04ag/smoke.log 2>&1; tail -1 gpurun_out/r04ag/smoke.log
timeout 400 python bench.py > gpurun_out/r04ag/bench.json 2> gpurun_out/r04ag/bench.err
tail -c 200 gpurun_out/r04ag/bench.json
