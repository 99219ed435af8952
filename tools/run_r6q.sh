# round 6, GPU call Q: one-launch fan-out with group minima / claimed finishing: parity tests, tile variants, time split by diagnostic builds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6q
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round6.py -m gpu -x -q -k "topk" > gpurun_out/r6q/topk_tests.log 2>&1; tail -15 gpurun_out/r6q/topk_tests.log
timeout 600 python tools/microbench.py topk_sizes > gpurun_out/r6q/topk_sizes.txt 2>&1; cat gpurun_out/r6q/topk_sizes.txt
for r in 1 2; do for v in shipped dense11 dense12 dense21 dense22 dense11_d1 dense11_d2 dense22_d1 dense22_d2 dense11_f dense22_f; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | grep -v chain16 | sed "s/^/$v: /"
done; done > gpurun_out/r6q/topk_dense_tiles.txt 2>&1
cat gpurun_out/r6q/topk_dense_tiles.txt
