// attn_common.h -- launch parameters and helpers shared by the head_dim-64 attention kernels (attn_flash.hip: 32x32x16 family for short / masked /
// cross-attention launches; attn16.hip: 16x16x32 family for the long unmasked joint attention).
#pragma once
#include "common.h"

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; const bf16_t* resid; const uint8_t* mask;
  long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss;
  int B, H, Sq, Skv, kv_div, n_qtiles;
  float qscale, out_scale;
  // key-split tail (KVSPLIT instantiation): workgroups >= n_main own (b, h, chunk) of the ragged last query tile
  int n_main, kv_splits, chunk_keys, rem_rows, tile_rows;
  const float* bias; long long bias_sh;   // additive score bias [H, Sq, Skv] fp32 (head stride bias_sh) or null: the 32x32x16 kernel's mask path only
  float* part_o;    // [B*H*kv_splits, rem_rows, 64] unnormalised partial outputs
  float2* part_ml;  // [B*H*kv_splits, rem_rows] (running max in log2 units, row sum)
};


// key-split tail of long launches (see attn_flash.hip: plan / merge kernel live there, both kernel families use them)
struct SplitPlan {
  int splits = 1, chunk_keys = 0, rem_rows = 0, n_full = 0;
  size_t bytes = 0;
};
SplitPlan mrag_plan_kv_split(int B, int H, int Sq, int Skv, int tile_rows, int slots);
int mrag_launch_attn_combine(hipStream_t s, const AttnP& p);
// attn16 workgroup shape: query blocks of 16 rows per wave (4 waves per workgroup).  3 -> 192-row workgroups, three per CU (168 VGPRs): shipped;
// 4 -> 256-row workgroups, two per CU (214 VGPRs): fewer K / V fragment bytes and barriers per FLOP.  With the optimistic sweep, QB = 4 is 2.6 % AHEAD
// at S = 17 776 in a cold interleaved microbenchmark (6.05 vs 6.22 ms, profiles/r4_attn_qb4_ab.txt) and 1.5 % BEHIND inside the denoise step, where the
// chip sits in its sustained power state (6.38 vs 6.28 ms per launch, 562 vs 560 ms per step on one box: profiles/r4_attn_step_ab.txt) -- the step is
// what ships, so 3 everywhere.  -DMRAG_ATTN16_QB=4 builds the other shape (tools/build_variant.sh) for A/B runs.
inline int mrag_attn16_qb(int /*Sq*/) {
#ifdef MRAG_ATTN16_QB
  return MRAG_ATTN16_QB;
#else
  return 3;
#endif
}
inline int mrag_attn16_rows(int qb) { return 64 * qb; }
inline int mrag_attn16_slots(int qb) { return (qb == 3 ? 3 : (qb == 2 ? 4 : 2)) * 256; }
// attn16.hip: long unmasked sequences (Sq > 128, Skv >= 256) on v_mfma_f32_16x16x32_bf16; returns MRAG_ENOTSUP for shapes it does not take
int mrag_launch_attn16(hipStream_t s, AttnP p, const SplitPlan* pl, void* workspace, int qb);


// v_max3_f32 through asm: plain fmaxf() on MFMA outputs makes hipcc emit a canonicalising v_max per operand
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

