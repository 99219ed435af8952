# developer tool: matrix-pipe utilisation of the two dominant kernels from rocprofv3 PMC counters (one counter per pass, kernel trace only):
#   SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) for the joint attention and the GEMMs at the BASELINE shapes (8-wave tile and, since round 3, the persistent four-wave kernel).
# Writes gpurun_out/pmc_util/summary.json (copy to profiles/).
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_util
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_util/$C -- python3 $R/tools/microbench.py attn gemm > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
res = collections.defaultdict(dict)
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_util/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            n = r["Kernel_Name"]
            if "attn16_kernel<3, 4, 3," in n:
                key = "attn16_kernel<3,4,3,true> B=2 H=48 S=17776"
            elif "gemm_bf16_kernel<2, 4, 8, 4, 0, 0, false>" in n and r["Grid_Size"] == "2562048":
                key = "gemm_bf16_kernel<2,4,8,4,NONE> M=35552 N=9216 K=3072"
            elif "gemm_w4_kernel<" in n:
                key = "gemm_w4_kernel<" + n.split("gemm_w4_kernel<")[1].split(">")[0] + "> (persistent four-wave GEMM, the DiT shapes of microbench.py gemm, averaged over its launches)"
            else:
                continue
            acc[key].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v)
out = {}
for k, d in res.items():
    o = dict(d)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        # MFMA_BUSY sums busy cycles over the 1024 SIMDs; GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs
        o["kernel_cycles"] = d["GRBM_GUI_ACTIVE"] / 8
        o["mfma_busy_fraction_per_simd"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        o["wave_wait_fraction"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"]:
        o["lds_bank_conflict_fraction"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
    out[k] = o
json.dump(out, open("gpurun_out/pmc_util/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
