# developer tool: rocprofv3 kernel stats of tools/cama_prof.py (CAMA predict with fixed random features):  tools/prof_cama.sh <tag>  -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-cama}
R=$PWD
export TMPDIR=/tmp
cd /tmp && ITERS=20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/cama_prof.py > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
grep "CAMA predict" gpurun_out/prof_$TAG.log
