# workgroups per launch of the row reduction's round kernel (one problem): CYTO_BID_GRID
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-grid_ab}; mkdir -p $O
for g in 256 512 1024 2048; do
  echo "== CYTO_BID_GRID=$g"
  CYTO_BID_GRID=$g timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 2>&1 | grep "rep=2" | cut -c1-200 | tee -a $O/grid_$g.log
done
