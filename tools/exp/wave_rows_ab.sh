# from how many rows per workgroup on a launch's uncertified rows are read by the wave that holds them (sc_top2_wave) instead of the
# whole workgroup one after the other (sc_top2_block): CYTO_SC_WAVE_ROWS = 0 (never: rounds 4-5) / 6 / 12 (default) / 24 / 48
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-wave_rows_ab}; mkdir -p $O
run() {
  echo "== $1"
  env $1 timeout 300 python tools/wide_large.py t20000 c4s10000 --reps 3 2>&1 | grep "rep=2" | cut -c1-190
  env $1 timeout 300 python tools/batch_chunks_bench.py 64 10000 2>&1 | grep "rep=1" | cut -c1-150
  env $1 timeout 300 python tools/batch_chunks_bench.py 256 10000 2>&1 | grep "rep=1" | cut -c1-150
  env $1 timeout 300 python tools/c5_chunks.py 10000 500 50 2>&1 | grep "K= 50" | cut -c1-150
}
for v in 0 6 12 24 48; do run "CYTO_SC_WAVE_ROWS=$v" | tee -a $O/ab.log; done
