timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -k "gemm" 2>&1 | tail -3
for g in 1 2 4 8 16; do echo "GROUP_M=$g"; MRAG_GEMM_GROUP_M=$g timeout 200 python tools/microbench.py gemm 2>&1 | grep gemm; done
