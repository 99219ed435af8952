# round 6, GPU call Y: one-launch fan-out: group minima loaded once per query through LDS -- retrieval tests, phase stamps, sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6y/topk_tests.log 2>&1; tail -3 gpurun_out/r6y/topk_tests.log
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py > gpurun_out/r6y/topk_diag.txt 2>&1
cat gpurun_out/r6y/topk_diag.txt
timeout 600 python tools/microbench.py topk_sizes > gpurun_out/r6y/topk_sizes.txt 2>&1; cat gpurun_out/r6y/topk_sizes.txt
timeout 600 python tools/microbench.py topk_small > gpurun_out/r6y/topk_small.txt 2>&1; cat gpurun_out/r6y/topk_small.txt
