# round 6: where the cooperative full-row relaxations cost the searches that have none: the phase timers of wide_aug for several builds
# of lap_wide.hip (CYTOHIP_LIB): v2 = outside the round loop, v3 = v2 out of line, v6 = v2 for n >= 16384 and the in-round sweep below, nocoop = before them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O; rm -f $O/ab.log
for pass in 1 2; do
for lib in v6 v3 nocoop; do
  echo "== lib=$lib" >> $O/ab.log
  CYTOHIP_LIB=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_$lib.so timeout 600 python tools/wide_large.py c4s10000 t20000 u20000 --reps 3 2>&1 | grep -A1 -E "rep=2" | sed -e 's/colsol==.*cache=/cache=/' -e 's/ | free=.*//' -e 's/wide_arr.*wide_aug/wide_aug/' >> $O/ab.log
  CYTOHIP_LIB=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_$lib.so timeout 600 python tools/c3_walls.py 2>&1 | grep -E "resident" | cut -c1-220 >> $O/ab.log
done
done
cat $O/ab.log
