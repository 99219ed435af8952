# round 6, GPU call Z2: phase stamps of the one-launch fan-out form, counters flushed at the end of the kernel (the first version's atomics sat in front of the next vmcnt wait)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6z
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6z/topk_diag2.txt
cat gpurun_out/r6z/topk_diag2.txt
