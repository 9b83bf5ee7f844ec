cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 600 python tools/check_wide.py --big > $O/check_wide.log 2>&1; echo "rc=$?" >> $O/check_wide.log
grep -c " OK " $O/check_wide.log; grep "BAD\|ALL OK\|FAIL\|rc=" $O/check_wide.log | cut -c1-200
bash tools/run/r05f.sh
