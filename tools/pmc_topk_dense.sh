# developer tool: the one-launch fan-out search (topk_dense_kernel) at BASELINE config #1's size (10 000 x 768 table, 256 queries) under rocprofv3 PMC counters, one
# counter per pass (matrix-pipe busy, wait fraction, HBM bytes; FETCH_SIZE / WRITE_SIZE in KiB, FETCH x2 on gfx950 as the guide prescribes)
#   tools/pmc_topk_dense.sh      -> gpurun_out/pmc_topk_dense/summary.json
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_topk_dense
cat > /tmp/topk_dense_once.py <<PY
import sys
sys.path.insert(0, "$R")
import torch
from motionrag_amd import ops
db = torch.randn(10000, 768, device="cuda"); q = torch.randn(256, 768, device="cuda")
for _ in range(12):
    ops.topk(db, q, 12, order="mfma")
torch.cuda.synchronize()
PY
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_topk_dense/$C -- python3 /tmp/topk_dense_once.py > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json
res = {}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    v = []
    for f in glob.glob(f"gpurun_out/pmc_topk_dense/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "topk_dense_kernel" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    if v:
        res[c] = sum(v[2:]) / max(len(v) - 2, 1)
        res[c + "_launches"] = len(v)
if "GRBM_GUI_ACTIVE" in res:
    cyc = res["GRBM_GUI_ACTIVE"] / 8
    res["kernel_cycles"] = cyc
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if k in res:
            res[k + "_per_unit"] = {"per_simd(1024)": res[k] / (cyc * 1024), "per_cu(256)": res[k] / (cyc * 256)}
if "SQ_WAIT_ANY" in res and "SQ_WAVE_CYCLES" in res:
    res["wait_fraction_of_wave_cycles"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_bytes_per_launch_corrected"] = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
    res["algorithmic_bytes_per_launch"] = {"table + queries read once": 10000 * 768 * 4 + 256 * 768 * 4, "first scores + group minima written and the passing groups read": 256 * 10112 * 4 + 256 * 316 * 4 * 2 + 256 * 20 * 128,
                                           "second scoring: 16 rows + the query per query": 256 * 17 * 768 * 4, "outputs": 256 * 12 * 8}
    res["algorithmic_bytes_per_launch"]["total"] = sum(res["algorithmic_bytes_per_launch"].values())
    res["method"] = "rocprofv3 --pmc, one counter per pass, averages over launches 3..12 of one process; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md)"
json.dump(res, open("gpurun_out/pmc_topk_dense/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
