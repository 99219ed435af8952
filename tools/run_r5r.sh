# round-5 evidence run: DynamiCrafter step order check, UNet step kernel stats, HBM traffic of the round's memory-bound kernels
mkdir -p gpurun_out/r5_r
python3 - > gpurun_out/r5_r/dc_order.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import microbench as mb
from motionrag_amd import workloads as W
net = W.dynamicrafter1024_unet("cuda")
for p in ("bf16", "fp8", "bf16", "fp8"):
    mb.unet(p, net)
PY
cat gpurun_out/r5_r/dc_order.txt | grep -v amdgpu.ids
bash tools/prof.sh svd r5_r_svd_unet_step
bash tools/prof.sh unet r5_r_dc_unet_step
bash tools/pmc_traffic_r5.sh > gpurun_out/r5_r/traffic.log 2>&1; tail -30 gpurun_out/r5_r/traffic.log
