# round 6, GPU call E: the tests touched since call D, ip_attn_folded with aligned packed reads (+ nontemporal variant), the step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6e
python -m pytest tests/test_gpu_round6.py tests/test_gpu_models.py tests/test_gpu_pipelines.py tests/test_gpu_fullwidth_golden.py tests/test_gpu_attn_processor_golden.py -q > gpurun_out/r6e/pytest_sel.txt 2>&1; echo "rc=$?" >> gpurun_out/r6e/pytest_sel.txt
tail -25 gpurun_out/r6e/pytest_sel.txt | cut -c1-220
for r in 1 2 3; do for v in shipped ipnt; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py r6 2>&1 | grep -E "^r6 (ip_attn|score)" | sed "s/^/$v: /"
done; done > gpurun_out/r6e/ipfold_ab.txt 2>&1
cat gpurun_out/r6e/ipfold_ab.txt
timeout 600 python tools/r6_step_ab.py 2 3 > gpurun_out/r6e/step_ab_toggles.txt 2>&1
grep -v amdgpu gpurun_out/r6e/step_ab_toggles.txt | tail -10
