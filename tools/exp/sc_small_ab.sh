# up to how many bids a launch of the round kernel gives every bid its own wave (CYTO_SC_SMALL), and how far the driver queues ahead (CYTO_SC_AHEAD)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-small_ab}; mkdir -p $O
for v in 512 1024 2048 4096; do
  echo "== CYTO_SC_SMALL=$v"
  CYTO_SC_SMALL=$v timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 2>&1 | grep "rep=2" | cut -c1-25,128-175 | tee -a $O/small_$v.log
done
for v in 8 16 128; do
  echo "== CYTO_SC_AHEAD=$v"
  CYTO_SC_AHEAD=$v timeout 600 python tools/wide_large.py u20000 u50000 --reps 3 2>&1 | grep "rep=2" | cut -c1-25,128-175 | tee -a $O/ahead_$v.log
done
