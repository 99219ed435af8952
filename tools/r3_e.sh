mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_svd.py tests/test_gpu_unet.py tests/test_gpu_pipelines.py tests/test_gpu_kernels.py -m gpu -q --timeout 900 -x 2>&1 | grep -v "^E    *+\|^E   *where" | tail -15 > gpurun_out/r3e_tests.txt
cat gpurun_out/r3e_tests.txt
timeout 600 python tools/microbench.py svd > gpurun_out/r3e_svd.txt 2>&1; tail -3 gpurun_out/r3e_svd.txt
