# round 6, GPU call D: smoke(), the full GPU suite, ipfold grid A/B, the judged command unprofiled + under rocprofv3, the step alone under rocprofv3, PMC traffic of the attention
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6d
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6d/smoke.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r6d/tests.log
for r in 1 2; do for v in shipped ipwg8; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python tools/microbench.py r6 2>&1 | grep -E "^r6 (ip_attn|score|layernorm|1 GiB)" | sed "s/^/$v: /"
done; done > gpurun_out/r6d/ipfold_ab.txt 2>&1
cat gpurun_out/r6d/ipfold_ab.txt
timeout 900 python bench.py > gpurun_out/r6d/bench_unprofiled.json 2> gpurun_out/r6d/bench_unprofiled.err
cut -c1-600 gpurun_out/r6d/bench_unprofiled.json
bash tools/run_final.sh r6d_final > gpurun_out/r6d/run_final.log 2>&1
tail -8 gpurun_out/r6d/run_final.log | cut -c1-400
bash tools/prof_step_only.sh r6d_step_only > gpurun_out/r6d/step_only.log 2>&1
tail -2 gpurun_out/r6d/step_only.log | cut -c1-300
bash tools/pmc_traffic.sh r6 > gpurun_out/r6d/pmc_traffic.log 2>&1
tail -2 gpurun_out/r6d/pmc_traffic.log | cut -c1-600
