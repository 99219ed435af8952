# round 6, GPU call N: the fan-out search in ONE launch (topk_dense_kernel): parity tests, then launch shapes / tile variants at BASELINE config #1's size
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round6.py -m gpu -x -q -k "topk" > gpurun_out/r6n/topk_tests.log 2>&1; tail -15 gpurun_out/r6n/topk_tests.log
timeout 600 python tools/microbench.py topk_sizes > gpurun_out/r6n/topk_sizes.txt 2>&1; cat gpurun_out/r6n/topk_sizes.txt
for r in 1 2; do for v in shipped dense11 dense12 dense21 dense22; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | sed "s/^/$v: /"
done; done > gpurun_out/r6n/topk_dense_tiles.txt 2>&1
cat gpurun_out/r6n/topk_dense_tiles.txt
