# round 6: the new parity test of the cooperative full-row relaxations at n = 16 384 (tests/golden/large_c4s16384_wide.npz)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
timeout 900 python -m pytest tests/test_large_gpu.py -m gpu -x -q -k "full_row_relaxations_by_the_whole_workgroup or chunk_of_16384" > $O/t.log 2>&1; tail -15 $O/t.log
timeout 300 python tools/wide_large.py c4s10000 --reps 1 --rebuild -1 2>&1 | grep -E "rep=0" | sed -e "s/ | free=.*dense=/ dense=/" -e "s/trivial=.*//" | cut -c1-200
