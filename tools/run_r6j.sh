# round 6, GPU call J: fp8 attention with the V^T fragments requested in front of the row-sum MFMA and sub-tile 1's K fragments under sub-tile 0's P.V MFMAs -- tests, then
# shipped vs fp8head (the kernel of the previous commit) on the DynamiCrafter level-0 attention and the CFG step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6j
python -m pytest tests/test_gpu_fp8.py -q > gpurun_out/r6j/pytest_fp8.txt 2>&1; echo "rc=$?" >> gpurun_out/r6j/pytest_fp8.txt
tail -5 gpurun_out/r6j/pytest_fp8.txt | cut -c1-200
for r in 1 2 3; do for v in shipped fp8head; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 900 python - 2>&1 <<'PY' | grep -E "attention|step" | sed "s/^/$v: /"
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import microbench as mb
from motionrag_amd import ops, workloads as W
B, H, S = 32, 5, 9216
qkv = torch.randn(B, S, 3, H, 64, device="cuda").to(torch.bfloat16)
t8 = mb.timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], fp8=True), iters=20)
t16 = mb.timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]), iters=20)
fl = 4.0 * B * H * S * S * 64
print(f"DC level-0 spatial attention [32 x 9216 x 5 x 64]: fp8 path (amax + quantise + attn8) {t8*1e3:.3f} ms = {fl/t8/1e12:.0f} TFLOP/s, bf16 {t16*1e3:.3f} ms = {fl/t16/1e12:.0f} TFLOP/s")
del qkv
net = W.dynamicrafter1024_unet("cuda")
for prec in ("bf16", "fp8"):
    print("DC step", prec, mb.unet(prec, net)["ms_per_cfg_step"])
PY
done; done > gpurun_out/r6j/fp8_ab.txt 2>&1
cat gpurun_out/r6j/fp8_ab.txt
