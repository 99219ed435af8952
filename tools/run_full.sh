# developer tool: full GPU validation = gpu tests + smoke + judged bench + rocprof kernel stats (written under gpurun_out/<tag>)
TAG=${1:-x}
R=$PWD
mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8 | tee gpurun_out/$TAG/tests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
timeout 1500 python bench.py 2>&1 | tail -1 | tee gpurun_out/$TAG/bench.json
export TMPDIR=/tmp
cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/$TAG/prof_bench.log 2>&1
cd $R
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/$TAG/prof_bench.log | cut -c1-300
