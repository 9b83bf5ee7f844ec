cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -v "^  File\|^Extension" $O/check_wide.log | cut -c1-330 | tail -50
