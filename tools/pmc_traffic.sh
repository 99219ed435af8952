# developer tool: HBM traffic of the dominant kernel (joint attention at the BASELINE shape) by rocprofv3 PMC, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), unit = KiB, and on gfx950
# FETCH_SIZE reports half of a wide coalesced stream -> doubled.  Writes profiles/<tag>_attn_traffic.json.
TAG=${1:-r3}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/traffic
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic/f -- python3 $R/tools/microbench.py attn > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic/w -- python3 $R/tools/microbench.py attn > /dev/null 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
def collect(d, counter, kern):
    vals = []
    for f in glob.glob(f"gpurun_out/traffic/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals
kern = "attn16_kernel<3, 4, 3,"      # the shipped long-sequence instantiation of attn16.hip (QB = 3, 4 waves, 3-stage ring), with or without the key-split tail
f, w = collect("f", "FETCH_SIZE", kern), collect("w", "WRITE_SIZE", kern)
fc, wc = collect("f", "FETCH_SIZE", "attn_combine_kernel"), collect("w", "WRITE_SIZE", "attn_combine_kernel")   # the tail's merge kernel, one per launch
avg = lambda v: sum(v) / len(v) if v else 0.0
out = {"kernel": "attn16_kernel<3,4,3,true> + attn_combine_kernel", "shape": "B=2 H=48 S=17776 D=64", "launches": len(f),
       "FETCH_SIZE_KiB_per_launch": avg(f) + avg(fc), "WRITE_SIZE_KiB_per_launch": avg(w) + avg(wc),
       "combine_kernel_KiB_per_launch": {"FETCH_SIZE": avg(fc), "WRITE_SIZE": avg(wc)},
       "hbm_bytes_per_launch_corrected": (2 * (avg(f) + avg(fc)) + avg(w) + avg(wc)) * 1024,
       "algorithmic_bytes_per_launch": 4 * 2 * 17776 * 3072 * 2,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 (gfx950 wide-stream correction)"}
json.dump(out, open(f"profiles/{tag}_attn_traffic.json", "w"), indent=1); json.dump(out, open(f"gpurun_out/traffic/{tag}_attn_traffic.json", "w"), indent=1)  # gpurun merges only gpurun_out/ back
print(json.dumps(out))
PY
