mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d_prof -- python3 $GRAFT_REPO_ROOT/tools/gemm_sk_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r3d_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3d_prof -name "*kernel_stats*" | head
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3d_prof/**/*kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
import collections
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name'][:90]; grid=r.get('Grid_Size_X',r.get('Grid_Size',''))
    agg[(name,grid)].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); print(f"{k[0]:90s} grid {k[1]:>8s} n={len(v):3d} median {v2[len(v2)//2]/1e3:8.1f} us")
PY
