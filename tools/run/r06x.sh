# round 6: how many searches of one problem at once (cyto_lap_opts.wide_par; 16 by default, at most the CUs of one XCD = 32)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O; rm -f $O/ab.log
for par in 16 8 12 24 32 16; do
  echo "== wide_par=$par" >> $O/ab.log
  timeout 400 python tools/wide_large.py c4s10000 t20000 u20000 u50000 --reps 3 --par $par 2>&1 | grep -A1 -E "rep=2|rror" | sed -e 's/colsol==golden \([A-Za-z]*\) duals==wide-golden \([A-Za-z]*\).*cache=/ok=\1,\2 cache=/' -e 's/ | free=.*par_batches/ par_batches/' -e 's/wide_arr: .*| wide_aug/wide_aug/' | grep -v "^--" >> $O/ab.log
done
cat $O/ab.log
