# developer tool: the judged command under rocprofv3 (kernel trace + stats), exactly `python3 bench.py` with its defaults.
# The default run launches the dominant attention kernel at TWO shapes (the headline S = 17 776 and the shipped-config clip at S = 6 976): the per-kernel
# average that is comparable with bench.py's HIP-event figure is taken per grid size from the trace (attn_by_grid.json) before the trace is deleted.
TAG=${1:-final}
R=$PWD
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py > $R/gpurun_out/$TAG/bench_under_rocprof.log 2>&1
cd $R
python3 - $TAG <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
trace = glob.glob(f"gpurun_out/{tag}/prof/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(list)
for f in trace:
    for r in csv.DictReader(open(f)):
        if "attn16_kernel" in r["Kernel_Name"] or "attn_fwd_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:80], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = [{"kernel": k, "grid_threads_x": g, "launches": len(v), "avg_ms": sum(v) / len(v) / 1e6, "min_ms": min(v) / 1e6, "max_ms": max(v) / 1e6}
       for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))]
json.dump(out, open(f"gpurun_out/{tag}/attn_by_grid.json", "w"), indent=1)
for o in out[:6]:
    print(o)
PY
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
grep -m1 '^{"metric"' gpurun_out/$TAG/bench_under_rocprof.log > gpurun_out/$TAG/bench.json
cut -c1-400 gpurun_out/$TAG/bench.json
