# round 6: the scheduler flags of the device compiler on lap_wide.hip (one build of the library per flag, CYTOHIP_LIB):
#   f1 -amdgpu-sched-strategy=max-ilp   f2 -amdgpu-schedule-metric-bias=0   f3 -amdgpu-use-amdgpu-trackers=1
#   f4 -amdgpu-sched-strategy=max-memory-clause   f5 -amdgpu-schedule-relaxed-occupancy   f6 -amdgpu-sched-strategy=iterative-ilp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O; rm -f $O/ab.log
for lib in HEAD f1 f2 f3 f4 f5 f6 HEAD; do
  echo "== lib=$lib" >> $O/ab.log
  L=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_$lib.so; [ $lib = HEAD ] && L=
  CYTOHIP_LIB=$L timeout 300 python tools/wide_large.py c4s10000 t20000 u20000 u50000 --reps 3 2>&1 | grep -A1 -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/ | free=.*//' -e 's/wide_arr: //' | grep -v "^--" >> $O/ab.log
done
cat $O/ab.log
