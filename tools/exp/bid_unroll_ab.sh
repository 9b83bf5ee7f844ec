# quads in flight per lane in the full-row sweeps of wide_sc_round (registers -> waves per SIMD): 2 / 3 / 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-unroll_ab}; mkdir -p $O
for u in 2 3 4; do
  echo "== CYTO_BID_UNROLL=$u"
  CYTO_BID_UNROLL=$u timeout 600 python tools/wide_large.py u20000 u50000 t20000 c4s10000 --reps 3 2>&1 | grep "rep=2" | cut -c1-200 | tee -a $O/unroll_$u.log
done
