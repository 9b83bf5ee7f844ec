mkdir -p gpurun_out/r04ai
run() { K=$1; G=$2; if [ $G = 0 ]; then unset CYTO_SUBBATCHES; else export CYTO_SUBBATCHES=$G; fi; echo -n "K=$K subbatches=$G: "; timeout 100 python tools/batch_chunks_bench.py $K 2>&1 | tail -2 | head -1 | cut -c18-120; }
{
for G in 0 2 4 8; do run 8 $G; done
for G in 0 4 5 10 20; do run 20 $G; done
for G in 0 8 16 32; do run 64 $G; done
for G in 0 16 32; do run 256 $G; done
} > gpurun_out/r04ai/ab.log 2>&1
cat gpurun_out/r04ai/ab.log
