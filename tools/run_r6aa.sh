# round 6, GPU call AA: one-launch fan-out, rank counts on 64-bit keys: retrieval tests, phase stamps, sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6aa
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6aa/topk_tests.log 2>&1; tail -3 gpurun_out/r6aa/topk_tests.log
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6aa/topk_diag.txt
cat gpurun_out/r6aa/topk_diag.txt
timeout 600 python tools/microbench.py topk_sizes 2>&1 | grep -v amdgpu.ids > gpurun_out/r6aa/topk_sizes.txt; cat gpurun_out/r6aa/topk_sizes.txt
timeout 600 python tools/microbench.py topk_small 2>&1 | grep -v amdgpu.ids > gpurun_out/r6aa/topk_small.txt; cat gpurun_out/r6aa/topk_small.txt
