mkdir -p gpurun_out
bash tools/run_final.sh r3_a > gpurun_out/r3g_run_final.log 2>&1; tail -8 gpurun_out/r3g_run_final.log
bash tools/pmc_traffic.sh r3 > gpurun_out/r3g_traffic.log 2>&1; tail -2 gpurun_out/r3g_traffic.log
timeout 900 python3 bench.py > gpurun_out/r3g_bench_unprofiled.json 2> gpurun_out/r3g_bench_unprofiled.err; cut -c1-600 gpurun_out/r3g_bench_unprofiled.json
