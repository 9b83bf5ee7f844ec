cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/round_floor tools/dbg/round_floor.hip 2>&1 | tail -3
timeout 120 /tmp/round_floor | tee $O/round_floor.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/scope_lat tools/dbg/scope_lat.hip 2>&1 | tail -3
timeout 120 /tmp/scope_lat | tee $O/scope_lat.txt
