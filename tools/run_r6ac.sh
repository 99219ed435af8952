# round 6, GPU call AC: one-launch fan-out with three-tile workgroups (96 of 128 queries per eight-wave workgroup): retrieval tests, A/B vs four-tile workgroups, stamps, sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6ac
timeout 1800 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6ac/topk_tests.log 2>&1; tail -3 gpurun_out/r6ac/topk_tests.log
for r in 1 2 3; do for v in shipped dense22; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | grep "mfma:" | sed "s/^/$v: /"
done; done > gpurun_out/r6ac/topk_three_tiles_ab.txt 2>&1
cat gpurun_out/r6ac/topk_three_tiles_ab.txt
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6ac/topk_diag.txt
cat gpurun_out/r6ac/topk_diag.txt
timeout 600 python tools/microbench.py topk_sizes 2>&1 | grep -v amdgpu.ids > gpurun_out/r6ac/topk_sizes.txt; cat gpurun_out/r6ac/topk_sizes.txt
