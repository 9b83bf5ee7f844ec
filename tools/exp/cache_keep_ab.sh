# (historical: the CYTO_CACHE_KEEP knob -- a rebuild leaving still-valid caches alone -- was measured with this script and removed again: DESIGN "Tried")
mkdir -p gpurun_out/r04ae
for kp in 0 32 48 0 32; do
  export CYTO_CACHE_KEEP=$kp
  echo "== CYTO_CACHE_KEEP=$kp"
  timeout 200 python tools/wide_large.py t20000 c4s10000 u20000 u50000 --reps 3 2>&1 | grep -v "^    wide_arr" | grep "rep=[12]" | sed 's/total diff [^ ]* //' | cut -c1-260
  timeout 100 python tools/batch_chunks_bench.py 2>&1 | tail -3 | cut -c1-200
done > gpurun_out/r04ae/ab.log 2>&1
cat gpurun_out/r04ae/ab.log
