# round 6, GPU call B: the new tests first, then the whole GPU suite, then same-box A/B of round 6's items
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6b
python -m pytest tests/test_gpu_round6.py -q -x > gpurun_out/r6b/pytest_round6.txt 2>&1; echo "rc=$?" >> gpurun_out/r6b/pytest_round6.txt
tail -15 gpurun_out/r6b/pytest_round6.txt
python -m pytest tests -m gpu -q --deselect tests/test_gpu_round6.py > gpurun_out/r6b/pytest_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r6b/pytest_gpu.txt
tail -15 gpurun_out/r6b/pytest_gpu.txt
for v in shipped r6base shipped r6base; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py r6 2>&1 | grep "^r6" | sed "s/^/$v: /"
done > gpurun_out/r6b/microbench_r6.txt 2>&1
cat gpurun_out/r6b/microbench_r6.txt
timeout 900 python tools/r6_step_ab.py 3 3 > gpurun_out/r6b/step_ab_toggles.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6b/step_ab_toggles.txt | tail -16
bash tools/ab_step.sh shipped r6base > gpurun_out/r6b/step_ab_lib.txt 2>&1
cat gpurun_out/r6b/step_ab_lib.txt
for r in 1 2; do for v in shipped r6base; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 900 python tools/microbench.py svd unet 2>&1 | grep -E "CFG step" | sed "s/^/$v: /" | cut -c1-160
done; done > gpurun_out/r6b/unet_ab.txt 2>&1
cat gpurun_out/r6b/unet_ab.txt
