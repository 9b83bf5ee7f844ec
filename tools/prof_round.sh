#!/bin/bash
# Round-end measurement set: bench line, rocprofv3 kernel stats of the same command (headline only, and with the extra
# legs), PMC traffic passes of the 20 000 LAP, PMC pass of the c3-sized cost GEMM.  Usage (gpurun): bash tools/prof_round.sh r02a
TAG=${1:-r03b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_extras -o ${TAG}x -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_extras.log 2>&1
mkdir -p $OUT/pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o fetch -- python $R/tools/quick_lap_bench.py 20000 > $OUT/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o write -- python $R/tools/quick_lap_bench.py 20000 > $OUT/pmc/write.log 2>&1
mkdir -p $OUT/gemm_pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/gemm_pmc -o gemm -- python $R/tools/gemm_only.py > $OUT/gemm_pmc/gemm.log 2>&1
find $OUT -name "*.csv" | head -30
# wave-stall reasons and LDS bank conflicts of the same GEMM (second PMC pass: the SQ block has 8 counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/gemm_pmc -o gemm_stall -- python $R/tools/gemm_only.py > $OUT/gemm_pmc/gemm_stall.log 2>&1
