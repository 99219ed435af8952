# developer tool: kernel statistics of the headline step alone (no secondary workloads, no e2e clip): where the step's wall time is not a kernel
TAG=${1:-step_only}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-e2e --no-shipped-config --no-cpu-baseline --no-ceilings > $R/gpurun_out/$TAG/bench.log 2>&1
cd $R
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
grep -m1 '^{"metric"' gpurun_out/$TAG/bench.log | cut -c1-300
