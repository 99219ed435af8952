mkdir -p gpurun_out
ROUNDS=4 timeout 600 python tools/gemm_sk_ab.py > gpurun_out/r3c_gemm_sk_ab.txt 2>&1
cat gpurun_out/r3c_gemm_sk_ab.txt
