#!/bin/bash
# Round-end measurement set: bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
TAG=${1:-r01c}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof.log 2>&1
mkdir -p $OUT/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o fetch -- python $R/tools/quick_lap_bench.py 20000 > $OUT/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o write -- python $R/tools/quick_lap_bench.py 20000 > $OUT/pmc/write.log 2>&1
find $OUT -name "*.csv" | head -20
