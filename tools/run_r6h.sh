# round 6, GPU call H: ip_attn_folded with counted vmcnt waits over unrolled heads -- tests, then shipped (8 waves per SIMD) vs 7 / 5 waves per SIMD vs the compiler-waited form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6h
python -m pytest tests/test_gpu_round6.py tests/test_gpu_fullwidth_golden.py tests/test_gpu_attn_processor_golden.py tests/test_gpu_kernels.py -q -k "ip_attn or fullwidth or attn_processor or folded or motion" > gpurun_out/r6h/pytest_sel.txt 2>&1; echo "rc=$?" >> gpurun_out/r6h/pytest_sel.txt
tail -6 gpurun_out/r6h/pytest_sel.txt | cut -c1-200
for r in 1 2 3; do for v in shipped ipw7 ipw5 ipold; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py ipfold r6 2>&1 | grep -E "ipfold finishing|^r6 ip_attn" | sed "s/^/$v: /"
done; done > gpurun_out/r6h/ipfold_ab.txt 2>&1
cat gpurun_out/r6h/ipfold_ab.txt
bash tools/ab_step.sh shipped ipold > gpurun_out/r6h/step_ab_lib.txt 2>&1
cat gpurun_out/r6h/step_ab_lib.txt
