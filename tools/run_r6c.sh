# round 6, GPU call C: the round-6 tests (all of them), the multi-process and fp8 tests, then sweeps / A/Bs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_round6.py tests/test_gpu_fp8.py tests/test_gpu_multiproc.py -q > gpurun_out/r6c/pytest_sel.txt 2>&1; echo "rc=$?" >> gpurun_out/r6c/pytest_sel.txt
tail -40 gpurun_out/r6c/pytest_sel.txt | cut -c1-220
python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -q -k "cama or layernorm or groupnorm or resampler or encoder or topk or native" > gpurun_out/r6c/pytest_sel2.txt 2>&1; echo "rc=$?" >> gpurun_out/r6c/pytest_sel2.txt
tail -8 gpurun_out/r6c/pytest_sel2.txt | cut -c1-220
timeout 600 python tools/microbench.py copy_probe 2>&1 | grep copy_probe > gpurun_out/r6c/copy_probe.txt
cat gpurun_out/r6c/copy_probe.txt
for r in 1 2 3; do for v in shipped nolna; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/cama_prof.py 2>&1 | grep "CAMA predict" | sed "s/^/$v: /" | cut -c1-400
done; done > gpurun_out/r6c/cama_ab.txt 2>&1
cat gpurun_out/r6c/cama_ab.txt
timeout 900 python tools/microbench.py unet 2>&1 | grep -E "CFG step" | cut -c1-200 > gpurun_out/r6c/dc_fp8.txt
python - >> gpurun_out/r6c/dc_fp8.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import microbench as mb
from motionrag_amd import workloads as W
net = W.dynamicrafter1024_unet("cuda")
for r in range(3):
    for prec in ("bf16", "fp8"):
        out = mb.unet(prec, net)
        print(r, prec, out)
from motionrag_amd import ops
B, H, S = 32, 5, 9216
qkv = torch.randn(B, S, 3, H, 64, device="cuda").to(torch.bfloat16)
for r in range(3):
    t8 = mb.timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], fp8=True), iters=10)
    t16 = mb.timeit(lambda: ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]), iters=10)
    fl = 4.0 * B * H * S * S * 64
    print(f"DC level-0 spatial attention [32 x 9216 x 5 x 64]: fp8 path (amax + quantise + attn8) {t8*1e3:.3f} ms = {fl/t8/1e12:.0f} TFLOP/s, bf16 {t16*1e3:.3f} ms = {fl/t16/1e12:.0f} TFLOP/s")
PY
tail -12 gpurun_out/r6c/dc_fp8.txt | cut -c1-300
