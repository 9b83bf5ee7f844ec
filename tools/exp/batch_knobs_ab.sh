# batches of chunk LAPs through the one-launch round: launches queued ahead, workgroups per problem, wave-per-bid threshold
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-batch_ab}; mkdir -p $O
run() { echo "== $1"; env $1 timeout 300 python tools/batch_chunks_bench.py 16 5000 2>&1 | grep "rep=1" | cut -c1-150; env $1 timeout 300 python tools/batch_chunks_bench.py 64 10000 2>&1 | grep "rep=1" | cut -c1-150; }
run "X=0" | tee $O/base.log
for a in 8 64; do run "CYTO_SC_AHEAD=$a" | tee -a $O/ahead.log; done
for t in 8192 16384; do run "CYTO_BID_TOTAL=$t" | tee -a $O/total.log; done
for m in 256 512; do run "CYTO_SC_SMALL=$m" | tee -a $O/small.log; done
