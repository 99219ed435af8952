# round 6, GPU call W: one-launch fan-out: pages touched while waiting (vs not), unrolled rank counts -- retrieval tests, phase stamps, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6w
timeout 1500 python -m pytest tests -m gpu -x -q -k "topk or rag or retriev" > gpurun_out/r6w/topk_tests.log 2>&1; tail -3 gpurun_out/r6w/topk_tests.log
MRAG_HIP_LIB=$PWD/tools/lib_topk_stats.so MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/topk_diag.py > gpurun_out/r6w/topk_diag.txt 2>&1
cat gpurun_out/r6w/topk_diag.txt
for r in 1 2 3; do for v in shipped dense_notouch; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | grep "mfma:" | sed "s/^/$v: /"
done; done > gpurun_out/r6w/topk_touch_ab.txt 2>&1
cat gpurun_out/r6w/topk_touch_ab.txt
