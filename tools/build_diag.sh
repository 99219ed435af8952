# diagnostic build with per-phase s_memtime stamps (never shipped): tools/libmrag_diag.so
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-comment -DMRAG_ATTN_STAMPS -DMRAG_GEMM_STAMPS -shared \
  motionrag_amd/csrc/api.hip motionrag_amd/csrc/attn_flash.hip motionrag_amd/csrc/attn16.hip motionrag_amd/csrc/gemm_bf16.hip -o tools/libmrag_diag.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-comment -DMRAG_ATTN_STAMPS -DMRAG_GEMM_STAMPS -DMRAG_GEMM_SAMEK -shared \
  motionrag_amd/csrc/api.hip motionrag_amd/csrc/attn_flash.hip motionrag_amd/csrc/attn16.hip motionrag_amd/csrc/gemm_bf16.hip -o tools/libmrag_diag_samek.so
