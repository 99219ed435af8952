# round 6, GPU call M: persistent GEMM with nontemporal output stores (w4nt) vs shipped: the four DiT GEMMs alone, then the step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6m
for r in 1 2 3; do for v in shipped w4nt; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py gemm 2>&1 | grep "^gemm" | sed "s/^/$v: /"
done; done > gpurun_out/r6m/gemm_nt_ab.txt 2>&1
cat gpurun_out/r6m/gemm_nt_ab.txt
bash tools/ab_step.sh shipped w4nt > gpurun_out/r6m/step_ab_lib.txt 2>&1
cat gpurun_out/r6m/step_ab_lib.txt
