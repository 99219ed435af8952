# round 3, call B: stream-K tests + A/B, and the two failing tests of call A with full tracebacks
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "gemm or qkv" -x 2>&1 | tail -25 > gpurun_out/r3b_gemm_tests.txt
cat gpurun_out/r3b_gemm_tests.txt
ROUNDS=4 timeout 600 python tools/gemm_sk_ab.py > gpurun_out/r3b_gemm_sk_ab.txt 2>&1
cat gpurun_out/r3b_gemm_sk_ab.txt
timeout 600 python -m pytest "tests/test_gpu_kernels.py::test_attention32_family_matches_reference_and_attn16" tests/test_gpu_pipelines.py::test_cogvideox_baseline_pipeline_without_motion_injection -m gpu -q --timeout 600 --tb=short 2>&1 | grep -v "^E    *+\|^E   *where" | tail -60 > gpurun_out/r3b_fail_tb.txt
cat gpurun_out/r3b_fail_tb.txt
