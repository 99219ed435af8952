# interleaved A/B of library variants on ONE device in the STEP context (the chip's sustained power state, not a cold microbenchmark):
#   tools/ab_step.sh <variant> <variant> ...      (variant = tools/lib_<variant>.so from tools/build_variant.sh; "shipped" = motionrag_amd/libmrag_hip.so)
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  echo -n "$v: "; MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --no-e2e --no-shipped-config --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step; attention', d['roofline']['avg_launch_ms'], 'ms =', d['roofline']['frac'])"
done; done
