# developer tool: build a library variant for interleaved A/B runs:  tools/build_variant.sh <name> [extra hipcc flags for attn_flash.hip / gemm]
cd "$(dirname "$0")/.."
NAME=$1; shift
STAMP=$(python3 -c "from motionrag_amd._lib import source_hash; print(source_hash())")   # same sources as the product library: the loader accepts it through the explicit MRAG_HIP_LIB override only
OBJ=/tmp/mrag_variant_$NAME; mkdir -p $OBJ
for f in api gemm_bf16 attn_flash attn16 attn_fp8 comm norm pointwise preprocess topk unet_ops cama_seq attn_small probe; do   # motionrag_amd/_lib.py: SOURCES
  EXTRA=""; case $f in attn_flash|attn16|attn_fp8) EXTRA="-fno-slp-vectorize";; api) EXTRA="-DMRAG_SOURCE_HASH=\"$STAMP\"";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $EXTRA "$@" -c motionrag_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib_$NAME.so $OBJ/*.o && echo built tools/lib_$NAME.so
