cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
CYTO_TRACE_CHUNKS=1 timeout 900 python tools/c4_strong_trace.py > $O/trace.log 2>&1; grep -v "^\[chunks\]" $O/trace.log | cut -c1-300; grep "^\[chunks\]" $O/trace.log | head -120
