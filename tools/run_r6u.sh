# round 6, GPU call U: in-kernel phase stamps of the one-launch fan-out form (diagnostic build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6u
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py > gpurun_out/r6u/topk_diag.txt 2>&1
cat gpurun_out/r6u/topk_diag.txt
