# round 6, GPU call AB: the dense fan-out form in two launches for grids that are not resident at once: retrieval tests, sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6ab
timeout 1800 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6ab/topk_tests.log 2>&1; tail -3 gpurun_out/r6ab/topk_tests.log
timeout 600 python tools/microbench.py topk_sizes 2>&1 | grep -v amdgpu.ids > gpurun_out/r6ab/topk_sizes.txt; cat gpurun_out/r6ab/topk_sizes.txt
timeout 600 python tools/microbench.py topk_small 2>&1 | grep -v amdgpu.ids > gpurun_out/r6ab/topk_small.txt; cat gpurun_out/r6ab/topk_small.txt
