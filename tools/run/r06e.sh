cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
for k in 16384 1024; do echo "== CYTO_CACHE_KEEP_MB=$k"; CYTO_CACHE_KEEP_MB=$k timeout 600 python tools/c3_walls.py 2>&1 | cut -c1-250; done > $O/c3_walls.log 2>&1
cat $O/c3_walls.log
