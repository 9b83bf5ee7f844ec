#!/bin/bash
# copies what tools/prof_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked), named <tag>_*
TAG=${1:-r06}
S=gpurun_out/$TAG; D=profiles
cp $S/bench.json $D/${TAG}_bench.json
for n in 50000 20000; do
  cp $(find $S/prof_n$n -name "*kernel_stats.csv" | head -1) $D/${TAG}_kernel_stats_bench_n$n.csv
  f=$(find $S/prof_n$n -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $D/${TAG}_kernel_trace_bench_n$n.csv
  cp $S/rounds_n$n.txt $D/${TAG}_round_launches_n$n.txt
done
cp $(find $S/prof_extras -name "*kernel_stats.csv" | head -1) $D/${TAG}_kernel_stats_bench_extras.csv
for t in n20000 n50000 t20000; do
  cp $S/pmc_$t/fetch_counter_collection.csv $D/${TAG}_pmc_fetch_counter_collection_$t.csv 2>/dev/null || cp $(find $S/pmc_$t -name "fetch*counter_collection.csv" | head -1) $D/${TAG}_pmc_fetch_counter_collection_$t.csv
  cp $S/pmc_$t/write_counter_collection.csv $D/${TAG}_pmc_write_counter_collection_$t.csv 2>/dev/null || cp $(find $S/pmc_$t -name "write*counter_collection.csv" | head -1) $D/${TAG}_pmc_write_counter_collection_$t.csv
done
mkdir -p /tmp/pmcj
for t in n20000 n50000 t20000; do
  rm -rf /tmp/pmcj/$t; mkdir -p /tmp/pmcj/$t
  cp $D/${TAG}_pmc_fetch_counter_collection_$t.csv /tmp/pmcj/$t/fetch_counter_collection.csv
  cp $D/${TAG}_pmc_write_counter_collection_$t.csv /tmp/pmcj/$t/write_counter_collection.csv
  n=${t#n}; n=${n#t}
  if [ $t = t20000 ]; then python tools/pmc_to_json.py /tmp/pmcj/$t $n $D/${TAG}_pmc_traffic_$t.json "few-cell-type 20 000^2 (tools/wide_large.py t20000: instances.typed_unique_cost(n, n, 20))"; else python tools/pmc_to_json.py /tmp/pmcj/$t $n $D/${TAG}_pmc_traffic_$t.json; fi
done
# the batched legs
for K in 256 32; do
  cp $S/prof_c4_K$K/c4_kernel_stats.csv $D/${TAG}_kernel_stats_c4_chunks_K$K.csv
  cp $S/overlap_c4_K$K.txt $D/${TAG}_trace_overlap_c4_chunks_K$K.txt
  cp $S/rounds_c4_K$K.txt $D/${TAG}_round_launches_c4_chunks_K$K.txt
done
cp $S/prof_c5/c5_kernel_stats.csv $D/${TAG}_kernel_stats_c5_chunks_K50.csv
cp $S/overlap_c5.txt $D/${TAG}_trace_overlap_c5_chunks_K50.txt
cp $S/pmc_traffic_c4_K256.json $D/${TAG}_pmc_traffic_c4_chunks_K256.json
python - <<PY
import csv, collections, re
for tag, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("$S/pmc_c4/%s_counter_collection.csv" % tag)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"])).replace("cyto::", "")
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    with open("$D/${TAG}_pmc_%s_by_kernel_c4_chunks_K256.csv" % tag, "w") as f:
        f.write("Kernel,Dispatches,%s_KB_sum,%s_KB_per_dispatch\n" % (key, key))
        for k, (c, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%.1f,%.1f\n' % (k, c, v, v / c))
PY
cp $(find $S/gemm_pmc -name "gemm_counter_collection.csv" | head -1) $D/${TAG}_gemm_pmc_counter_collection.csv
cp $(find $S/gemm_pmc -name "gemm_stall_counter_collection.csv" | head -1) $D/${TAG}_gemm_pmc_stall_counter_collection.csv
ls -la $D | grep ${TAG}_
