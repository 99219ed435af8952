timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -k attention 2>&1 | tail -3
echo "--- NW=8"; timeout 120 python tools/microbench.py attn 2>&1 | grep "attn joint"
echo "--- NW=4"; MRAG_ATTN_NW=4 timeout 120 python tools/microbench.py attn 2>&1 | grep "attn joint"
