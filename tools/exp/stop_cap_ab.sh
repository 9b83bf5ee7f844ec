# phases of the scaled row reduction end at wide_stop(n) = min(n / 128, CAP) active rows: the library must be built with
# CYTO_EXTRA_FLAGS=-DWIDE_STOP_CAP=<cap> (the oracle's twin: -DJV_WIDE_STOP_CAP); timing only, several sizes (seed = n)
cd $GRAFT_REPO_ROOT
timeout 600 python tools/quick_lap_bench.py 16000 30000 40000 45000 60000 2>&1 | sed -E 's/.*(n=[0-9]+) wall=([0-9.]+)ms.*arr_ms=([0-9.]+) aug_ms=([0-9.]+).*/\1 wall \2 arr \3 aug \4/'
