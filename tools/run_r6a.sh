# round 6, GPU call A: the GPU suite on the tree + same-box A/B of the round-5 library against the shipped one on the convolution-heavy workloads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6a/pytest_gpu.txt
tail -5 gpurun_out/r6a/pytest_gpu.txt
for r in 1 2; do for v in r5 shipped; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  for w in vae svd_vae cogvideox_vae; do
    MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py $w 2>&1 | grep -vi "amdgpu.ids" | sed "s/^/$v: /" | cut -c1-200
  done
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 900 python tools/microbench.py svd unet 2>&1 | grep -E "CFG step" | sed "s/^/$v: /" | cut -c1-160
done; done > gpurun_out/r6a/conv_ab.txt 2>&1
cat gpurun_out/r6a/conv_ab.txt
