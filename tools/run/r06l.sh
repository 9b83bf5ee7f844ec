cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python tools/wide_large.py u20000 u24000 u30000 u33000 u50000 --reps 2 2>&1 | grep -E "rep=1|wide_aug" | cut -c1-420 > $O/sizes.log; cat $O/sizes.log
