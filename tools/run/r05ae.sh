cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ae; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python $R/tools/c3_lap_breakdown.py > $O/c3.log 2>&1
tail -2 $O/c3.log | cut -c1-200
python - <<'PY'
import csv, glob, os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05ae/c3/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'colred_partial' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
prev_end=None
for r in rows[max(0,idx-12):idx+34]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us +{(e-s)/1e3:8.1f}  {r['Kernel_Name'][:64]}")
# the end of the solve
j=[i for i,r in enumerate(rows) if 'wide_aug' in r['Kernel_Name']][-1]
r=rows[j]; print('wide_aug', (int(r['Start_Timestamp'])-t0)/1e3, '+', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
