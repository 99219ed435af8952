# developer tool: rocprofv3 kernel stats of a microbench target:  tools/prof.sh <target> <tag>   -> gpurun_out/<tag>_kernel_stats.csv
T=$1; TAG=${2:-$1}
R=$PWD
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/microbench.py $T > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
tail -2 gpurun_out/prof_$TAG.log
