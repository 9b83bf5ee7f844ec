# round 6: same-box A/B of the cooperative full-row relaxations: HEAD / the library built from
# the lap_wide.hip before them -- CYTOHIP_LIB picks the build -- on instances that have no full-row relaxations to gain from, and on c3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O; rm -f $O/ab.log
for rep in 1 2; do
  for lib in "" cytospace_amd/build/libcytohip_nocoop.so; do
    echo "== lib=${lib:-HEAD} pass $rep" >> $O/ab.log
    CYTOHIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python tools/wide_large.py c4s10000 t20000 u20000 --reps 3 2>&1 | grep -E "rep=[12]|rror" | cut -c1-220 >> $O/ab.log
    CYTOHIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python tools/c3_walls.py 2>&1 | grep -E "LAP alone|resident" | cut -c1-220 >> $O/ab.log
  done
done
cat $O/ab.log | cut -c1-30,120-230
timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_large_gpu.py -m gpu -x -q > $O/gputest.log 2>&1; grep -E "passed|failed" $O/gputest.log | tail -1
timeout 900 python tools/stress_lap.py 2000 60 200 3000 --rebuild -1 > $O/s1.log 2>&1; tail -1 $O/s1.log
timeout 900 python tools/stress_lap.py 2100 36 200 3000 --par 5 > $O/s2.log 2>&1; tail -1 $O/s2.log
