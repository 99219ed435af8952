timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -k attention 2>&1 | tail -4
echo "--- default"; timeout 120 python tools/microbench.py attn 2>&1 | grep attn
echo "--- PIPE=0"; MRAG_ATTN_PIPE=0 timeout 120 python tools/microbench.py attn 2>&1 | grep attn
