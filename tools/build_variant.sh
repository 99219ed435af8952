# developer tool: build a library variant for interleaved A/B runs:  tools/build_variant.sh <name> [extra hipcc flags for attn_flash.hip / gemm]
cd "$(dirname "$0")/.."
NAME=$1; shift
# the stamp is the digest of (source digest, extra flags): a variant built with -D flags never carries the product's stamp, so the loader takes it only through
# the explicit MRAG_HIP_LIB + MRAG_HIP_LIB_ANY_SOURCE=1 override (tools/ab*.sh set both)
STAMP=$(python3 -c "import hashlib,sys; from motionrag_amd._lib import source_hash; f=' '.join(sys.argv[1:]); print(hashlib.sha256((source_hash()+'|'+f).encode()).hexdigest()[:16] if f else source_hash())" "$@")
OBJ=/tmp/mrag_variant_$NAME; mkdir -p $OBJ
for f in api gemm_bf16 attn_flash attn16 attn_fp8 comm norm pointwise preprocess topk unet_ops cama_seq attn_small probe; do   # motionrag_amd/_lib.py: SOURCES
  EXTRA=""; case $f in attn_flash|attn16|attn_fp8) EXTRA="-fno-slp-vectorize";; api) EXTRA="-DMRAG_SOURCE_HASH=\"$STAMP\"";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $EXTRA "$@" -c motionrag_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib_$NAME.so $OBJ/*.o && echo built tools/lib_$NAME.so
