# the several-searches kernel's workgroups on one XCD (stride 8) or spread over all of them (stride 1), and how many at once
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-stride_ab}; mkdir -p $O
for st in 8 1; do for par in 16 24 32; do
  echo "== CYTO_PAR_STRIDE=$st --par $par"
  CYTO_PAR_STRIDE=$st timeout 600 python tools/wide_large.py u20000 u50000 t20000 --par $par --reps 3 2>&1 | grep "rep=2" | cut -c1-25,128-175,300-420 | tee -a $O/stride_${st}_$par.log
done; done
