# round 6: the finish of a batch of searches by parts (a -DCYTO_AUG_FIN_SPLIT build via CYTOHIP_LIB: "certificate passes" then also holds the
# claims + conflict check, "one-edge searches" the reset + logs + last barrier, "update+flip+reset" only the update and the flip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
CYTOHIP_LIB=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_split.so timeout 400 python tools/wide_large.py c4s10000 t20000 u20000 u50000 --reps 2 2>&1 | grep -A1 -E "rep=1|rror" | sed -e 's/colsol==golden.*cache=/cache=/' -e 's/relax=.*settled=/settled=/' -e 's/wide_arr: .*| wide_aug/wide_aug/' > $O/split.log
cat $O/split.log
