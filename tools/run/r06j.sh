cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
nproc; 
for rep in 1 2; do for v in 512 16384 0; do echo "== CYTO_SPIN_POLLS=$v"; CYTO_SPIN_POLLS=$v timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep "rep=" | cut -c1-120; CYTO_SPIN_POLLS=$v timeout 300 python tools/batch_chunks_bench.py 32 20000 2>&1 | grep "rep=1" | cut -c1-120; done; done > $O/spin.log 2>&1; cat $O/spin.log
