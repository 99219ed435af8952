# developer tool: matrix-pipe / LDS utilisation of the retrieval fan-out kernel (10^6 x 768 table, 256 queries) from rocprofv3 PMC counters, one counter per pass.
#   tools/pmc_topk.sh [library]      -> gpurun_out/pmc_topk/summary.json
export TMPDIR=/tmp
R=$PWD
LIBV=${1:-}
mkdir -p $R/gpurun_out/pmc_topk
cat > /tmp/topk_once.py <<PY
import os, sys
if "$LIBV":
    os.environ["MRAG_HIP_LIB"] = "$R/$LIBV"; os.environ["MRAG_HIP_LIB_ANY_SOURCE"] = "1"
sys.path.insert(0, "$R")
import torch
from motionrag_amd import ops
db = torch.randn(1000000, 768, device="cuda"); q = torch.randn(256, 768, device="cuda")
for _ in range(4):
    ops.topk(db, q, 12, order="mfma")
torch.cuda.synchronize()
PY
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_topk/$C -- python3 /tmp/topk_once.py > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS"):
    v = []
    for f in glob.glob(f"gpurun_out/pmc_topk/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "topk_mfma_kernel" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    if v:
        res[c] = sum(v[1:]) / max(len(v) - 1, 1)
if "GRBM_GUI_ACTIVE" in res:
    cyc = res["GRBM_GUI_ACTIVE"] / 8
    res["kernel_cycles"] = cyc
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
        if k in res:
            res[k + "_per_unit"] = {"per_simd(1024)": res[k] / (cyc * 1024), "per_cu(256)": res[k] / (cyc * 256)}
if "SQ_WAIT_ANY" in res and "SQ_WAVE_CYCLES" in res:
    res["wait_fraction_of_wave_cycles"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
json.dump(res, open("gpurun_out/pmc_topk/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
