cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 600 python tools/wide_large.py u20000 u50000 --reps 2 2>&1 | grep "sc prof\|rep=1" | cut -c1-420 | tee $O/prof.log
