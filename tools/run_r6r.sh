# round 6, GPU call R: the one-launch fan-out's grid wait: time between two looks at the `go` word (s_sleep 4 / 32 / 127) at 632 and 158 workgroups
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6r
for r in 1 2; do for v in dense11 dense11_s4 dense11_s127 dense22 dense22_s4 dense22_s127; do
  L=$PWD/tools/lib_$v.so
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 300 python tools/microbench.py topk_small 2>&1 | grep "^topk" | grep "mfma:" | sed "s/^/$v: /"
done; done > gpurun_out/r6r/topk_dense_sleep.txt 2>&1
cat gpurun_out/r6r/topk_dense_sleep.txt
