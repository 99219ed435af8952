mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 2>&1 | grep -v "^E    *+\|^E   *where" | tail -40 > gpurun_out/r3f_all_gpu_tests.txt
tail -25 gpurun_out/r3f_all_gpu_tests.txt
