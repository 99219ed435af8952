# developer tool: PMC passes over the attention micro-benchmark (one kernel), summaries under gpurun_out/
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
export MRAG_ATTN_PIPE=${MRAG_ATTN_PIPE:-0}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p1 -- python3 $R/tools/microbench.py ${MB:-attn} > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p2 -- python3 $R/tools/microbench.py ${MB:-attn} > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/pmc/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in agg.items():
            if True:
                print(d, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
