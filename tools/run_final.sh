# developer tool: the judged command under rocprofv3 (kernel trace + stats), exactly `python3 bench.py` with its defaults
TAG=${1:-final}
R=$PWD
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py > $R/gpurun_out/$TAG/bench_under_rocprof.log 2>&1
cd $R
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
grep -m1 '^{"metric"' gpurun_out/$TAG/bench_under_rocprof.log > gpurun_out/$TAG/bench.json
cut -c1-400 gpurun_out/$TAG/bench.json
