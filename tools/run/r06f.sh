cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
CYTO_TRACE_CHUNKS=1 timeout 600 python tools/c3_walls.py > $O/c3_walls.log 2>&1; grep -v "^\[chunks\]" $O/c3_walls.log | cut -c1-250; grep "^\[chunks\]" $O/c3_walls.log | tail -8
