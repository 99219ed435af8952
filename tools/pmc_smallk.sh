# developer tool: HBM read / write bytes of the big GEMM launches (rocprofv3 PMC, separate passes; FETCH x2 on gfx950)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/smallk_traffic
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/smallk_traffic/f -- python3 $R/tools/smallk_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/smallk_traffic/w -- python3 $R/tools/smallk_probe.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
def collect(d, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/smallk_traffic/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "gemm_bf16_kernel" in r["Kernel_Name"]:
                vals[(r["Kernel_Name"][:75], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return vals
f, w = collect("f", "FETCH_SIZE"), collect("w", "WRITE_SIZE")
for k in f:
    fv = sum(f[k]) / len(f[k]) * 2 * 1024 / 1e6
    wv = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024 / 1e6
    print(f"{k[0]} grid={k[1]}: HBM read {fv:.0f} MB (x2-corrected), write {wv:.0f} MB per launch ({len(f[k])} launches)")
PY
