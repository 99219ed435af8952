# developer tool: library variants that differ only in topk.hip (other objects reused from motionrag_amd/build):
#   tools/topk_variants.sh <name> <topk source> [-D flags]   -> tools/lib_<name>.so   (load with MRAG_HIP_LIB=... MRAG_HIP_LIB_ANY_SOURCE=1)
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift; shift
mkdir -p /tmp/topkv
cp $SRC /tmp/topkv/topk_$NAME.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Imotionrag_amd/csrc -Iinclude "$@" -c /tmp/topkv/topk_$NAME.hip -o /tmp/topkv/topk_$NAME.o || exit 1
OBJS=$(ls motionrag_amd/build/*.o | grep -v "/topk.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lib_$NAME.so $OBJS /tmp/topkv/topk_$NAME.o -ldl && echo built tools/lib_$NAME.so
