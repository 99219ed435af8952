# interleaved A/B of library variants on ONE device for the two UNet CFG steps: tools/ab_unet.sh <variant> ... ("shipped" = motionrag_amd/libmrag_hip.so)
for r in 1 2; do for v in "$@"; do
  if [ "$v" = shipped ]; then L=$PWD/motionrag_amd/libmrag_hip.so; else L=$PWD/tools/lib_$v.so; fi
  MRAG_HIP_LIB=$L MRAG_HIP_LIB_ANY_SOURCE=1 timeout 900 python tools/microbench.py svd unet 2>&1 | grep -E "CFG step" | sed "s/^/$v: /" | cut -c1-110
done; done
