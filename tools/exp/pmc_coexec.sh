export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/coexec
cd /tmp
for C in SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/coexec/$C -- python3 $R/tools/microbench.py attn > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob
for c in sorted(glob.glob("gpurun_out/coexec/*")):
    name = c.split("/")[-1]
    vals = []
    for f in glob.glob(f"{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "attn16_kernel<3, 4, 1, 3, false," in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    print(f"{name:32s} {sum(vals)/max(1,len(vals)):.4g}  ({len(vals)} launches)")
PY
