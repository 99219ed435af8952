# round 6, GPU call I: ip_attn_folded -- shipped (counted waits, 7 waves per SIMD) vs 6 waves / unconstrained registers vs round 6's earlier form (compiler-waited, unconstrained)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6i
for r in 1 2 3; do for v in shipped ipw6 ipw1 ipold; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py ipfold r6 2>&1 | grep -E "ipfold finishing|^r6 ip_attn" | sed "s/^/$v: /"
done; done > gpurun_out/r6i/ipfold_ab.txt 2>&1
cat gpurun_out/r6i/ipfold_ab.txt
