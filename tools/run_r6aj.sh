# round 6, GPU call AJ: the judged command once more on the final tree after tools/microbench.py's retrieval timing took 50 calls for the small tables (bench.py's
# secondary workloads call it): the unprofiled line that replaces call AI's
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6aj
timeout 900 python bench.py > gpurun_out/r6aj/bench_unprofiled.json 2> gpurun_out/r6aj/bench_unprofiled.err
cut -c1-400 gpurun_out/r6aj/bench_unprofiled.json
timeout 300 python tools/microbench.py topk 2>&1 | grep -v amdgpu.ids > gpurun_out/r6aj/topk_microbench.txt; cat gpurun_out/r6aj/topk_microbench.txt
