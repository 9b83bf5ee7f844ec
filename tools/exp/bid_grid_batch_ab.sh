mkdir -p gpurun_out/r04ah
for g in 4096 8192 16384 32768 4096 16384; do
  export CYTO_BID_TOTAL=$g
  echo "== CYTO_BID_TOTAL=$g (workgroups of a bid launch over all problems; at least 64 per problem)"
  for K in 20 64 8; do timeout 100 python tools/batch_chunks_bench.py $K 2>&1 | tail -2 | head -1 | cut -c1-150; done
done > gpurun_out/r04ah/ab2.log 2>&1
cat gpurun_out/r04ah/ab2.log
