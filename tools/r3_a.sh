# round 3, call A: new parity tests + the 32x32x16 / 16x16x32 attention A/B on one box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "attention" -x 2>&1 | tail -8 > gpurun_out/r3a_attn_tests.txt
cat gpurun_out/r3a_attn_tests.txt
ROUNDS=4 timeout 600 python tools/attn_ab.py > gpurun_out/r3a_attn_ab.txt 2>&1
cat gpurun_out/r3a_attn_ab.txt | head -40
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_pipelines.py tests/test_gpu_multiproc.py -m gpu -q --timeout 900 2>&1 | tail -12 > gpurun_out/r3a_other_tests.txt
cat gpurun_out/r3a_other_tests.txt
