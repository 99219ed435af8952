# round 6, GPU call T: one-launch fan-out with rank-count ordering in the finishing phase: retrieval tests, sizes, time split
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6t/topk_tests.log 2>&1; tail -5 gpurun_out/r6t/topk_tests.log
timeout 600 python tools/microbench.py topk_sizes > gpurun_out/r6t/topk_sizes.txt 2>&1; cat gpurun_out/r6t/topk_sizes.txt
for r in 1 2; do for v in shipped dense11 dense22_d2; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | grep -v chain16 | sed "s/^/$v: /"
done; done > gpurun_out/r6t/topk_dense_tiles.txt 2>&1
cat gpurun_out/r6t/topk_dense_tiles.txt
