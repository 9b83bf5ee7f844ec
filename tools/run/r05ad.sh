cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ad; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3s -o c3s -- python $R/tools/wide_large.py c3s50000 --reps 3 > $O/c3s.log 2>&1
grep "rep=2" $O/c3s.log | cut -c1-25,100-175
python - <<'PY'
import csv, glob, os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05ad/c3s/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last solve: find last colred_partial
idx=[i for i,r in enumerate(rows) if 'colred_partial' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[max(0,idx-6):idx+40]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {r['Kernel_Name'][:70]}")
PY
