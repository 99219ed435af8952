# round 6, GPU call AI: FINAL evidence on the final tree -- smoke(), the full GPU suite, the judged command unprofiled + under rocprofv3, the step alone under rocprofv3,
# PMC traffic of the attention
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6ai
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6ai/smoke.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r6ai/tests.log
timeout 900 python bench.py > gpurun_out/r6ai/bench_unprofiled.json 2> gpurun_out/r6ai/bench_unprofiled.err
cut -c1-600 gpurun_out/r6ai/bench_unprofiled.json
bash tools/run_final.sh r6ai_final > gpurun_out/r6ai/run_final.log 2>&1
tail -8 gpurun_out/r6ai/run_final.log | cut -c1-400
bash tools/prof_step_only.sh r6ai_step_only > gpurun_out/r6ai/step_only.log 2>&1
tail -2 gpurun_out/r6ai/step_only.log | cut -c1-300
bash tools/pmc_traffic.sh r6 > gpurun_out/r6ai/pmc_traffic.log 2>&1
tail -2 gpurun_out/r6ai/pmc_traffic.log | cut -c1-600
