# the whole -m gpu suite, smoke(), the default bench line, and the launcher form of bench.py (two ranks on one device: harness self-test)
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -3 $O/gputest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 300 $O/bench.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-extras --no-cpu-baseline > $O/bench_torchrun2.json 2> $O/bench_torchrun2.err; echo "torchrun bench rc=$?"
cut -c1-200 $O/bench_torchrun2.json; grep -o '"n_gpus": [0-9]*' $O/bench_torchrun2.json
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --oversubscribe --no-extras --no-cpu-baseline > $O/bench_spawn2.json 2> $O/bench_spawn2.err; echo "spawn bench rc=$?"
grep -o '"n_gpus": [0-9]*' $O/bench_spawn2.json
