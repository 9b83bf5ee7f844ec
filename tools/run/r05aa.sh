cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05aa; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/tools/wide_large.py t20000 --reps 2 > $O/t.log 2>&1
python - <<'PY'
import csv, numpy as np, glob, os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05aa/t/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'wide_sc_round' in r['Kernel_Name']]
runs=[[sel[0]]]
for a,b in zip(sel,sel[1:]):
    if b[0]-a[1]>2_000_000: runs.append([])
    runs[-1].append(b)
d=np.array([(e-s)/1e3 for s,e in runs[-1]])
print(len(d),'launches sum',d.sum()/1e3,'ms span',(runs[-1][-1][1]-runs[-1][0][0])/1e6)
for lo,hi in ((0,9),(9,15),(15,30),(30,60),(60,120),(120,10000)):
    m=(d>=lo)&(d<hi); print(f'  {lo:4d}-{hi:5d} us: {m.sum():5d} launches {d[m].sum()/1e3:6.2f} ms')
b=[r for r in rows if 'build_row_caches' in r['Kernel_Name']]
print('cache builds', len(b), sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in b)/1e6,'ms (both reps)')
PY
