"""Turn the two rocprofv3 --pmc passes of tools/prof_pmc.sh into profiles/<tag>_pmc_traffic_n<N>.json.
Usage: pmc_to_json.py <dir with fetch_/write_counter_collection.csv> <n> <out.json> [instance description]"""
import sys, csv, json, re, collections
d, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("cyto::", "").replace(", ", ",")
    return name
agg = {}
for tag, key in (("fetch", "FETCH_SIZE_KB"), ("write", "WRITE_SIZE_KB")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{d}/{tag}_counter_collection.csv")):
        k = short(r["Kernel_Name"]); acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (c, v) in acc.items():
        agg.setdefault(k, {})[key] = round(v / c, 1)
        agg[k]["dispatches"] = c
for k, v in agg.items():
    v["hbm_read_bytes"] = int(2 * v.get("FETCH_SIZE_KB", 0.0) * 1024)
    v["hbm_write_bytes_uncalibrated"] = int(v.get("WRITE_SIZE_KB", 0.0) * 1024)
inst = sys.argv[4] if len(sys.argv) > 4 else f"uniform float32, seed n (tools/quick_lap_bench.py {n})"
json.dump({"n": n, "instance": inst,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per-dispatch averages; on gfx950 FETCH_SIZE reports "
                   "1/2 of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section; calibrated on "
                   "colred_partial / build_row_caches, which read the matrix exactly once): hbm_read_bytes = 2 * FETCH_SIZE * 1024",
           "kernels": agg}, open(out, "w"), indent=1)
print("wrote", out, len(agg), "kernels")
