#!/bin/bash
# Round-end measurement set: the default bench line; rocprofv3 kernel stats + trace of the same command (headline only: n = 50 000, and
# n = 20 000; with the extra legs: stats only); PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each in its own run, with --kernel-trace
# only) of one LAP at n = 20 000, n = 50 000 and of the few-cell-type 20 000 instance; PMC pass of the c3-sized cost GEMM.
# Usage (gpurun): bash tools/prof_round.sh r06 ; then tools/prof_collect.sh r06 copies the summaries into profiles/
# (round 6 adds the BATCHED legs: kernel stats + per-launch trace summaries + PMC traffic of 256 concurrent c4-chunk LAPs and of the 50 c5 chunks)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
for n in 50000 20000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n$n -o n$n -- python $R/bench.py --n $n --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof_n$n.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_extras -o extras -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_extras.log 2>&1
for n in 20000 50000; do
  mkdir -p $OUT/pmc_n$n
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_n$n -o fetch -- python $R/tools/quick_lap_bench.py $n > $OUT/pmc_n$n/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_n$n -o write -- python $R/tools/quick_lap_bench.py $n > $OUT/pmc_n$n/write.log 2>&1
done
mkdir -p $OUT/pmc_t20000
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_t20000 -o fetch -- python $R/tools/wide_large.py t20000 --reps 2 > $OUT/pmc_t20000/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_t20000 -o write -- python $R/tools/wide_large.py t20000 --reps 2 > $OUT/pmc_t20000/write.log 2>&1
mkdir -p $OUT/gemm_pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/gemm_pmc -o gemm -- python $R/tools/gemm_only.py > $OUT/gemm_pmc/gemm.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/gemm_pmc -o gemm_stall -- python $R/tools/gemm_only.py > $OUT/gemm_pmc/gemm_stall.log 2>&1
# ---- the batched legs (missing 3): 256 concurrent c4-chunk LAPs and the 50 single-cell chunks ----
for K in 256 32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4_K$K -o c4 -- python $R/tools/batch_chunks_bench.py $K 10000 > $OUT/prof_c4_K$K.log 2>&1
  f=$(find $OUT/prof_c4_K$K -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round > $OUT/rounds_c4_K$K.txt 2>&1
  python $R/tools/trace_overlap.py $f > $OUT/overlap_c4_K$K.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python $R/tools/c5_chunks.py 10000 500 50 > $OUT/prof_c5.log 2>&1
python $R/tools/trace_overlap.py $(find $OUT/prof_c5 -name "*kernel_trace.csv" | head -1) > $OUT/overlap_c5.txt 2>&1
mkdir -p $OUT/pmc_c4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c4 -o fetch -- python $R/tools/batch_chunks_bench.py 256 10000 > $OUT/pmc_c4/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c4 -o write -- python $R/tools/batch_chunks_bench.py 256 10000 > $OUT/pmc_c4/write.log 2>&1
python $R/tools/pmc_to_json.py $OUT/pmc_c4 10000 $OUT/pmc_traffic_c4_K256.json "256 concurrent c4 sub-spot chunk LAPs of 10 000 cells (tools/batch_chunks_bench.py 256 10000: a warm-up solve + 2 batched calls = 513 solves)" > $OUT/pmc_c4/json.log 2>&1
# the trace CSVs are large: keep the round kernel's per-launch summary and drop what gpurun would not carry back (64 MiB)
for n in 50000 20000; do
  f=$(find $OUT/prof_n$n -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_rounds.py $f wide_sc_round > $OUT/rounds_n$n.txt 2>&1
done
find $OUT/prof_c4_K256 $OUT/prof_c4_K32 $OUT/prof_c5 $OUT/pmc_c4 -name "*kernel_trace.csv" -delete
find $OUT -name "*.csv" -size +20M -exec rm {} \;
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" | head -40; du -sh $OUT
