# developer tool: build a library variant for interleaved A/B runs:  tools/build_variant.sh <name> [extra hipcc flags for attn_flash.hip / gemm]
cd "$(dirname "$0")/.."
NAME=$1; shift
OBJ=/tmp/mrag_variant_$NAME; mkdir -p $OBJ
for f in api gemm_bf16 attn_flash attn16 attn_fp8 comm norm pointwise preprocess topk unet_ops cama_seq attn_small; do   # motionrag_amd/_lib.py: SOURCES
  EXTRA=""; case $f in attn_flash|attn16) EXTRA="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $EXTRA "$@" -c motionrag_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib_$NAME.so $OBJ/*.o && echo built tools/lib_$NAME.so
