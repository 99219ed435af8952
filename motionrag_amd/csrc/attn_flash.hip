// attn_flash.hip -- head_dim-64 bf16 attention forward for gfx950 (flash-style, online softmax).
//
//   O[b, q, h, :] = resid + out_scale * softmax(Q K^T * scale  [masked])  V
//
// One kernel family serves every F.scaled_dot_product_attention call site of the MotionRAG hot path
// (include/mrag_hip.h): the 17 776-token CogVideoX joint attention, the 25-key motion
// ("ip") cross-attention with its fused `hidden + scale * ip` update, the Perceiver resampler
// (25 queries x 1 593 keys) and the block-causal CAMA encoder (bool mask).
//
// CDNA4 design:
//   * v_mfma_f32_32x32x16_bf16 with SWAPPED products: S^T = K . Q^T and O^T = V^T . P^T, so a lane
//     owns one query row -- row max / row sum are lane-local plus one v_permlane32_swap, and the S
//     accumulator registers are already the B operand of the PV product (no LDS round trip for P);
//   * each wave owns 32 query rows (Q fragments live in registers, pre-multiplied by
//     scale*log2 e so the softmax is a bare v_exp_f32); a workgroup is NW waves;
//   * K/V tiles of 64 keys are staged by 16-byte LDS-DMA (global_load_lds) into two K stages and two
//     V stages; K is XOR-swizzled on the source side for conflict-free ds_read_b128, V keeps
//     row-major [key][d] with its 64-byte halves swapped on odd key pairs and is consumed through
//     ds_read_b64_tr_b16 (hardware transpose) as the A operand of O^T = V^T . P^T;
//   * variants retired by measurement (round 3 pruned their code; numbers in DESIGN.md section 3): the intra-wave software-pipelined tile
//     (9.9 vs 8.7 ms at S = 17 776), 4-wave workgroups for long sequences, packed fp32 adds, per-32-key-block softmax + PV, static priorities
//     for the younger half, a half-tile stagger of the two halves, vector-pipe max subtraction, f32-add row sums;
//   * the running max is carried as the MFMA's C operand (S' = K.Q^T - m comes out of the chain,
//     no per-score subtract) and only moved when a tile raises it by more than THR (deferred
//     rescale): the rare path rescales O, l, the pending S' and the prefetched S' together;
//   * grid = q-tiles x (b, h) with an XCD-aware order: all q-tiles of one (b, h) run on one XCD,
//     so its K/V (4.5 MB at S = 17 776) streams from that XCD's L2.
#include "common.h"
#include <type_traits>
#include "../../include/mrag_hip.h"

#ifdef MRAG_ATTN_STAMPS
// diagnostic build only (tools/build_diag.sh): per-phase s_memtime sums of the long-sequence loop; never compiled into the product
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int mrag_debug_set_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)); }
#define MRAG_STAMP(T)                                                                  \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#endif



#include "attn_common.h"

namespace {

constexpr float kThr = 5.0f;          // deferred-rescale threshold in log2 units (P <= 32)
constexpr int KVB = 64;               // keys per tile
constexpr int TILE_BYTES = KVB * 128; // one K (or V) tile
constexpr int NS = 4;   // LDS ring stages per operand (DMA runs D = NS-2 tiles ahead); 2 x 64 KB fit two workgroups per CU
constexpr int V_BASE = NS * TILE_BYTES;  // LDS: K stages 0..NS-1, then V stages 0..NS-1

__device__ __forceinline__ bf16x8 scale_frag(u32x4 raw, float s) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(raw[i] << 16) * s;
    const float hi = __uint_as_float(raw[i] & 0xffff0000u) * s;
    r[i] = pack_bf2(lo, hi);
  }
  return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float half_swap_max(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max3_asm(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
}

struct Lane {
  int hh, k_row_off, k_swz, v_lane_off, v_half0, v_half1, head;
  unsigned kb[4], vc0, vc1;   // loop-invariant LDS addresses of this lane's K fragment chunks / V^T reads in ring stage 0 (MRAG_ATTN_IMM_STAGE)
};

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

// S'^T = K . Q^T + C for the two 32-key blocks of one tile (8 MFMAs); C = -running max.
// All 8 K fragments are requested in one asm statement (hipcc otherwise serialises read -> wait -> MFMA pairs and
// exposes the LDS latency four times per tile); `between()` runs while they are in flight (the DMA issue of a later
// tile), then one wait statement that names every destination releases them to the MFMAs.
template <typename Between = NoHook>
__device__ __forceinline__ void qk_tile(const char* kst, const Lane& ln, const bf16x8 (&qf)[4], const f32x16& negm, f32x16& s0, f32x16& s1,
                                        Between between = Between()) {
  // register-lean form (4 waves per SIMD hide the LDS latency): one 32-key block at a time
  const unsigned base = (unsigned)(size_t)(kst + ln.k_row_off);
  const unsigned b0 = base + ((0 + ln.hh) ^ ln.k_swz) * 16, b1 = base + ((2 + ln.hh) ^ ln.k_swz) * 16;
  const unsigned b2 = base + ((4 + ln.hh) ^ ln.k_swz) * 16, b3 = base + ((6 + ln.hh) ^ ln.k_swz) * 16;
  u32x4 kf[4];
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
               : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
  between();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]) :: "memory");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline-constant C
  // The running max is subtracted BY THE MATRIX PIPE: one more k-step whose key fragment is the constant (-1, 0, ..., 0) and whose
  // query fragment is (m, 0, ..., 0) adds -m to every score of the lane's query.  The vector pipe is the saturated one here
  // (per tile 32 v_exp + 32 row-sum adds + 17 v_max3 + 16 cvt_pk against 16 MFMAs); this trades 32 v_add (128 issue cycles)
  // for 2 MFMAs (64 matrix cycles, 16 issue cycles).  m is kept bf16-representable so the product is exact (softmax_tile).
  const u32x4 kneg = {ln.hh == 0 ? 0x0000bf80u : 0u, 0u, 0u, 0u};
  const u32x4 qm = {__float_as_uint(negm[1]), 0u, 0u, 0u};
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), qf[ks], s0, 0, 0, 0);
  asm volatile("ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %5 offset:4096\n\tds_read_b128 %2, %6 offset:4096\n\tds_read_b128 %3, %7 offset:4096\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "memory");
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), qf[ks], s1, 0, 0, 0);
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s1, 0, 0, 0);
}

// O^T += V^T . P^T (8 MFMAs).  The 16 transposed reads go through ONE asm statement: hipcc cannot see that the
// ds_read_tr16 builtin does not alias the in-flight LDS-DMA of later tiles and would drain it (s_waitcnt vmcnt(0))
// in the middle of the tile.  EXEC is all ones here (wave-uniform control flow only).
__device__ __forceinline__ void pv_tile(const char* vst, const Lane& ln, const bf16x8 (&pb)[4], f32x16& o0, f32x16& o1, f32x16& lacc) {
  {
    const unsigned c0 = (unsigned)(size_t)(vst + ln.v_lane_off + ln.v_half0), c1 = (unsigned)(size_t)(vst + ln.v_lane_off + ln.v_half1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      u32x2 u0[4], u1[4];
      const unsigned d0 = c0 + half * 4096, d1 = c1 + half * 4096;
      asm volatile("ds_read_b64_tr_b16 %0, %8 offset:0\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
                   "ds_read_b64_tr_b16 %4, %9 offset:0\n\tds_read_b64_tr_b16 %5, %9 offset:1024\n\t"
                   "ds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
                   "ds_read_b64_tr_b16 %6, %9 offset:2048\n\tds_read_b64_tr_b16 %7, %9 offset:3072\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(u0[0]), "=&v"(u0[1]), "=&v"(u0[2]), "=&v"(u0[3]), "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]), "=&v"(u1[3])
                   : "v"(d0), "v"(d1) : "memory");
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const u32x4 w0 = {u0[2 * k2][0], u0[2 * k2][1], u0[2 * k2 + 1][0], u0[2 * k2 + 1][1]};
        const u32x4 w1 = {u1[2 * k2][0], u1[2 * k2][1], u1[2 * k2 + 1][0], u1[2 * k2 + 1][1]};
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), pb[2 * half + k2], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), pb[2 * half + k2], o1, 0, 0, 0);
      }
    }
  }
}

// The ring stage as an INSTRUCTION IMMEDIATE (ds_read offset field) instead of per-tile address arithmetic: the main loop is unrolled by
// NS, so `t % NS` is a compile-time constant, the lane's base addresses are loop-invariant registers, and ~11 integer vector instructions
// per tile (of ~144) disappear from a loop whose vector pipe is ~77 % busy.
template <int STG, typename Between = NoHook>
__device__ __forceinline__ void qk_tile_imm(const Lane& ln, const bf16x8 (&qf)[4], const f32x16& negm, f32x16& s0, f32x16& s1, Between between = Between()) {
  u32x4 kf[4];
  asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"
               : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3]) : "v"(ln.kb[0]), "v"(ln.kb[1]), "v"(ln.kb[2]), "v"(ln.kb[3]), "n"(STG * TILE_BYTES) : "memory");
  between();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]) :: "memory");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 kneg = {ln.hh == 0 ? 0x0000bf80u : 0u, 0u, 0u, 0u};
  const u32x4 qm = {__float_as_uint(negm[1]), 0u, 0u, 0u};
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), qf[ks], s0, 0, 0, 0);
  asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3]) : "v"(ln.kb[0]), "v"(ln.kb[1]), "v"(ln.kb[2]), "v"(ln.kb[3]), "n"(STG * TILE_BYTES + 4096) : "memory");
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), qf[ks], s1, 0, 0, 0);
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s1, 0, 0, 0);
}

template <int STG>
__device__ __forceinline__ void pv_tile_imm(const Lane& ln, const bf16x8 (&pb)[4], f32x16& o0, f32x16& o1) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u32x2 u0[4], u1[4];
    if (half == 0) {
      asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
                   "ds_read_b64_tr_b16 %4, %9 offset:%10\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
                   "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %6, %9 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(u0[0]), "=&v"(u0[1]), "=&v"(u0[2]), "=&v"(u0[3]), "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]), "=&v"(u1[3])
                   : "v"(ln.vc0), "v"(ln.vc1), "n"(STG * TILE_BYTES), "n"(STG * TILE_BYTES + 1024), "n"(STG * TILE_BYTES + 2048), "n"(STG * TILE_BYTES + 3072) : "memory");
    } else {
      asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
                   "ds_read_b64_tr_b16 %4, %9 offset:%10\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
                   "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %6, %9 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(u0[0]), "=&v"(u0[1]), "=&v"(u0[2]), "=&v"(u0[3]), "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]), "=&v"(u1[3])
                   : "v"(ln.vc0), "v"(ln.vc1), "n"(STG * TILE_BYTES + 4096), "n"(STG * TILE_BYTES + 5120), "n"(STG * TILE_BYTES + 6144), "n"(STG * TILE_BYTES + 7168) : "memory");
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const u32x4 w0 = {u0[2 * k2][0], u0[2 * k2][1], u0[2 * k2 + 1][0], u0[2 * k2 + 1][1]};
      const u32x4 w1 = {u1[2 * k2][0], u1[2 * k2][1], u1[2 * k2 + 1][0], u1[2 * k2 + 1][1]};
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), pb[2 * half + k2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), pb[2 * half + k2], o1, 0, 0, 0);
    }
  }
}

// Both 32-key blocks' fragments requested UP FRONT (8 K reads / 16 V^T reads in flight, released to the MFMAs by counted lgkmcnt waits --
// LDS reads return in order): the second block's LDS latency hides under the first block's MFMAs instead of being paid behind them.  The LDS
// pipe is only ~25 % busy in this loop (256 B/clk per CU, tools/exp/lds_rate.hip; SQ_LDS_IDX_ACTIVE), so what the reads cost is their latency.
template <int STG, typename Between = NoHook>
__device__ __forceinline__ void qk_tile_imm8(const Lane& ln, const bf16x8 (&qf)[4], const f32x16& negm, f32x16& s0, f32x16& s1, Between between = Between()) {
  u32x4 k0[4], k1[4];
  asm volatile("ds_read_b128 %0, %8 offset:%12\n\tds_read_b128 %1, %9 offset:%12\n\tds_read_b128 %2, %10 offset:%12\n\tds_read_b128 %3, %11 offset:%12\n\t"
               "ds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %9 offset:%13\n\tds_read_b128 %6, %10 offset:%13\n\tds_read_b128 %7, %11 offset:%13"
               : "=&v"(k0[0]), "=&v"(k0[1]), "=&v"(k0[2]), "=&v"(k0[3]), "=&v"(k1[0]), "=&v"(k1[1]), "=&v"(k1[2]), "=&v"(k1[3])
               : "v"(ln.kb[0]), "v"(ln.kb[1]), "v"(ln.kb[2]), "v"(ln.kb[3]), "n"(STG * TILE_BYTES), "n"(STG * TILE_BYTES + 4096) : "memory");
  between();
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(k0[0]), "+v"(k0[1]), "+v"(k0[2]), "+v"(k0[3]) :: "memory");
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const u32x4 kneg = {ln.hh == 0 ? 0x0000bf80u : 0u, 0u, 0u, 0u};
  const u32x4 qm = {__float_as_uint(negm[1]), 0u, 0u, 0u};
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k0[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k0[ks]), qf[ks], s0, 0, 0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k1[0]), "+v"(k1[1]), "+v"(k1[2]), "+v"(k1[3]) :: "memory");
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k1[0]), qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k1[ks]), qf[ks], s1, 0, 0, 0);
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kneg), __builtin_bit_cast(bf16x8, qm), s1, 0, 0, 0);
}

template <int STG>
__device__ __forceinline__ void pv_tile_imm16(const Lane& ln, const bf16x8 (&pb)[4], f32x16& o0, f32x16& o1) {
  u32x2 u0[8], u1[8];
  asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
               "ds_read_b64_tr_b16 %4, %9 offset:%10\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
               "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
               "ds_read_b64_tr_b16 %6, %9 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%13"
               : "=&v"(u0[0]), "=&v"(u0[1]), "=&v"(u0[2]), "=&v"(u0[3]), "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]), "=&v"(u1[3])
               : "v"(ln.vc0), "v"(ln.vc1), "n"(STG * TILE_BYTES), "n"(STG * TILE_BYTES + 1024), "n"(STG * TILE_BYTES + 2048), "n"(STG * TILE_BYTES + 3072) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
               "ds_read_b64_tr_b16 %4, %9 offset:%10\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
               "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
               "ds_read_b64_tr_b16 %6, %9 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
               "s_waitcnt lgkmcnt(8)"
               : "=&v"(u0[4]), "=&v"(u0[5]), "=&v"(u0[6]), "=&v"(u0[7]), "=&v"(u1[4]), "=&v"(u1[5]), "=&v"(u1[6]), "=&v"(u1[7])
               : "v"(ln.vc0), "v"(ln.vc1), "n"(STG * TILE_BYTES + 4096), "n"(STG * TILE_BYTES + 5120), "n"(STG * TILE_BYTES + 6144), "n"(STG * TILE_BYTES + 7168) : "memory");
  asm volatile("" : "+v"(u0[0]), "+v"(u0[1]), "+v"(u0[2]), "+v"(u0[3]), "+v"(u1[0]), "+v"(u1[1]), "+v"(u1[2]), "+v"(u1[3]));   // half 0 has landed
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(u0[4]), "+v"(u0[5]), "+v"(u0[6]), "+v"(u0[7]), "+v"(u1[4]), "+v"(u1[5]), "+v"(u1[6]), "+v"(u1[7]) :: "memory");
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int i = half * 4 + 2 * k2;
      const u32x4 w0 = {u0[i][0], u0[i][1], u0[i + 1][0], u0[i + 1][1]};
      const u32x4 w1 = {u1[i][0], u1[i][1], u1[i + 1][0], u1[i + 1][1]};
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), pb[2 * half + k2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), pb[2 * half + k2], o1, 0, 0, 0);
    }
  }
}

struct Run {
  f32x16 o0, o1, negm;
  f32x16 lacc;   // row sums, accumulated by the matrix pipe: lacc = ones . P^T (every register of a lane holds l of its query)
  float m;
};

// softmax bookkeeping of one tile: masks, tile max, deferred rescale, P = exp2(S'), row sum, bf16 B fragments.
// register i of block kb holds key t*64 + kb*32 + (i&3) + 8*(i>>2) + 4*hh for query (lane & 31).
template <bool HAS_MASK, bool HAS_NEXT, typename Mid = NoHook>
__device__ __forceinline__ void softmax_tile(const AttnP& p, const int skv, const Lane& ln, int t, int nt, int qrow_c, f32x16& s0, f32x16& s1,
                                             f32x16& n0, f32x16& n1, Run& r, bf16x8 (&pb)[4], Mid mid = Mid()) {
  // skv = keys this workgroup scans (p.Skv, or its chunk of them in the key-split tail; the mask path is never split)
  // key held by register i of block kb: tile_start + kb*32 + (i&3) + 8*(i>>2) + 4*hh; the last tile of a long sequence starts at
  // Skv-64 (slid back), keys before t*64 were already consumed by the previous tile
  const int tile_start = skv >= KVB ? (t * KVB < skv - KVB ? t * KVB : skv - KVB) : 0;
  const int kbase_idx = tile_start + 4 * ln.hh;
  if (t == nt - 1 && (skv & (KVB - 1))) {
    const int lo = t * KVB;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kbase_idx + (i & 3) + 8 * (i >> 2);
      if (key < lo || key >= skv) s0[i] = -INFINITY;
      if (key + 32 < lo || key + 32 >= skv) s1[i] = -INFINITY;
    }
  }
  if constexpr (HAS_MASK) {   // the instantiation for a byte mask and / or an additive score bias
    if (p.mask) {
      const uint8_t* mrow = p.mask + (long long)qrow_c * p.Skv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_idx + (i & 3) + 8 * (i >> 2);
        if (key < p.Skv && mrow[key]) s0[i] = -INFINITY;
        if (key + 32 < p.Skv && mrow[key + 32]) s1[i] = -INFINITY;
      }
    }
    if (p.bias) {   // softmax(scale q k^T + bias): the scores are in log2 units here (Q carries scale * log2 e), so the bias is too
      const float* brow = p.bias + (long long)ln.head * p.bias_sh + (long long)qrow_c * p.Skv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase_idx + (i & 3) + 8 * (i >> 2);
        if (key < p.Skv) s0[i] = fmaf(brow[key], 1.4426950408889634f, s0[i]);
        if (key + 32 < p.Skv) s1[i] = fmaf(brow[key + 32], 1.4426950408889634f, s1[i]);
      }
    }
  }
  // The v_max3 chains below are inline asm: hipcc pads the MFMA -> VALU read hazard only for instructions it can see, and on the paths without a
  // DMA hook between the score MFMAs and this point nothing else separates them (a stale read only mis-places the running max -- harmless to the
  // result, P is formed from the real scores by compiler-visible code -- but the fp8 kernel showed what a stale NaN pattern does).  19 wait
  // states cover the 16-pass MFMA (< 1 % of a tile).
  asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");
  // tile max: four independent v_max3 chains, then across the two half-waves
  float ma = max3_asm(s0[0], s0[1], s0[2]), mb = max3_asm(s1[0], s1[1], s1[2]);
  float mc = max3_asm(s0[3], s0[4], s0[5]), md = max3_asm(s1[3], s1[4], s1[5]);
#pragma unroll
  for (int i = 6; i < 16; i += 4) {
    ma = max3_asm(ma, s0[i], s0[i + 1]);
    mb = max3_asm(mb, s1[i], s1[i + 1]);
    if (i + 3 < 16) {
      mc = max3_asm(mc, s0[i + 2], s0[i + 3]);
      md = max3_asm(md, s1[i + 2], s1[i + 3]);
    }
  }
  float tm = half_swap_max(max3_asm(max3_asm(ma, mb, mc), md, md));
  // deferred rescale: move the running max only on the first tile or when a row grew past THR
  const bool first = (t == 0);
  if (first || __any(tm > kThr)) {
    float delta = first ? fmaxf(tm, -1e30f) : fmaxf(tm, 0.f);
    // the shift lives in a bf16 MFMA operand: round the new running max to bf16 and move by the EXACT difference (both ends are
    // bf16 values, their fp32 difference is exact), so O / l / P all see the same shift.  Rows that do not move keep delta == 0.
    const float m_new = bf_round(r.m + delta);
    delta = m_new - r.m;
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int i = 0; i < 16; ++i) { r.o0[i] *= alpha; r.o1[i] *= alpha; }
      r.lacc[0] *= alpha;
    }
    r.m += delta;
    r.negm[0] = -r.m;
    r.negm[1] = __uint_as_float(ln.hh == 0 ? (unsigned)f2bf(r.m) : 0u);   // query-side fragment (m, 0, ..., 0) of the max-subtracting k-step
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }
    if constexpr (HAS_NEXT) {  // the prefetched S' of tile t+1 was formed against the old max
#pragma unroll
      for (int i = 0; i < 16; ++i) { n0[i] -= delta; n1[i] -= delta; }
    }
  }
  mid();  // staggered waves rendezvous here (between the max / rescale head and the exp body)
  float la = 0.f, lb = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s0[i] = __builtin_amdgcn_exp2f(s0[i]);
    s1[i] = __builtin_amdgcn_exp2f(s1[i]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 w0, w1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w0[j] = pack_bf2(s0[8 * s + 2 * j], s0[8 * s + 2 * j + 1]);
      w1[j] = pack_bf2(s1[8 * s + 2 * j], s1[8 * s + 2 * j + 1]);
      // row sum of the bf16 pairs the PV MFMAs consume: one v_dot2c_f32_bf16 (pair . (1, 1) + acc, fp32) per packed register instead of
      // two v_add_f32 per pair -- 16 instead of 32 vector instructions per tile, and l sums exactly the P values that multiply V
      // (the packed registers pass through an empty asm: hipcc 7.2 otherwise selects sub-register 0 of the u32x4 for all four dot2c)
      unsigned p0 = w0[j], p1 = w1[j];
      asm volatile("" : "+v"(p0), "+v"(p1));
      w0[j] = p0; w1[j] = p1;
      la = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, p0), __builtin_bit_cast(bf16v2, 0x3f803f80u), la, false);
      lb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, p1), __builtin_bit_cast(bf16v2, 0x3f803f80u), lb, false);
    }
    pb[s] = __builtin_bit_cast(bf16x8, w0);
    pb[2 + s] = __builtin_bit_cast(bf16x8, w1);
  }
  r.lacc[0] += la + lb;   // lane-local half of the row sum; the two half-waves are combined in the epilogue
}


// SHORTKV: key sets of at most one tile (motion tokens, text, temporal frames) -- same code, no barrier stagger; a separate
// instantiation so that profiles list the HBM-bound small-KV launches apart from the MFMA-bound long-sequence ones.
// KVSPLIT: the launch's last workgroups (blockIdx >= n_main) each scan ONE CHUNK of the keys for the ragged last query tile of a
// (b, h) pair and leave (unnormalised O, running max, row sum) in the workspace for attn_combine_kernel -- see launch_attn_split.
template <int NW, bool HAS_MASK, bool SHORTKV = false, bool KVSPLIT = false>
__global__ __launch_bounds__(NW * 64, (NW == 8 && !HAS_MASK) ? 4 : 1) void attn_fwd_kernel(const AttnP p) {
  constexpr int PPW = NW >= 8 ? 1 : 8 / NW;  // 1 KiB DMA pieces per wave per K (or V) tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, SGPR LDS bases
  const int r32 = lane & 31;
  Lane ln;
  ln.hh = lane >> 5;

  // ---- XCD-aware block -> (q-tile, b, h)
  const int nbh = p.B * p.H;
  int bh, qt;
  int skv = p.Skv, key0 = 0;          // keys this workgroup scans: [key0, key0 + skv)
  bool split_unit = false;
  if (KVSPLIT && (int)blockIdx.x >= p.n_main) {
    const int u = blockIdx.x - p.n_main;
    bh = u / p.kv_splits;
    qt = p.n_qtiles;                  // the ragged tile after the n_qtiles full ones
    key0 = (u % p.kv_splits) * p.chunk_keys;
    skv = p.Skv - key0 < p.chunk_keys ? p.Skv - key0 : p.chunk_keys;
    split_unit = true;
  } else if ((nbh & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    bh = (j / p.n_qtiles) * 8 + x;
    qt = j % p.n_qtiles;
  } else {
    bh = blockIdx.x / p.n_qtiles;
    qt = blockIdx.x % p.n_qtiles;
  }
  const int b = bh / p.H, h = bh % p.H;
  const int bkv = b / p.kv_div;
  ln.head = h;

  const int q0 = qt * (NW * 32) + wave * 32;
  const bool wave_active = q0 < p.Sq;
  const int qrow = q0 + r32;
  const int qrow_c = qrow < p.Sq ? qrow : p.Sq - 1;

  // ---- Q fragments: B operand of S^T = K.Q^T : lane holds Q[q = lane&31][d = 16 ks + 8 hh + j]
  bf16x8 qf[4];
  {
    const bf16_t* qp = p.Q + (long long)b * p.q_sb + (long long)qrow_c * p.q_ss + (long long)h * p.q_sh + ln.hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 raw = *(const u32x4*)(qp + ks * 16);
      qf[ks] = p.qscale == 1.0f ? __builtin_bit_cast(bf16x8, raw) : scale_frag(raw, p.qscale);
    }
  }

  // ---- LDS-DMA staging: piece = 8 keys x 128 B; lane i -> key (i >> 3), 16-byte position (i & 7)
  const bf16_t* kbase = p.K + (long long)bkv * p.k_sb + (long long)h * p.k_sh + (long long)key0 * p.k_ss;
  const bf16_t* vbase = p.V + (long long)bkv * p.v_sb + (long long)h * p.v_sh + (long long)key0 * p.v_ss;
  const int prow = lane >> 3, ppos = lane & 7;
  unsigned k_loff[PPW], v_loff[PPW];   // loop-invariant per-lane byte offsets inside a tile (source-side swizzles folded in)
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int kit = ((wave + i * NW) & 7) * 8 + prow;
    k_loff[i] = (unsigned)(kit * p.k_ss + (ppos ^ ((kit >> 1) & 7)) * 8) * 2u;
    v_loff[i] = (unsigned)(kit * p.v_ss + (ppos ^ (((kit >> 1) & 1) << 2)) * 8) * 2u;
  }
  // A partial last tile is SLID BACK to keys [Skv-64, Skv) (its already-seen keys are masked in softmax_tile), and prefetches past
  // the end re-read that tile: every DMA of a >= 64-key sequence is a full in-range tile -> scalar tile base + loop-invariant
  // per-lane offset, no per-lane clamping (no 64-bit lane arithmetic, nothing to spill).  Shorter key sets clamp rows instead.
  const int last_start = skv - KVB;
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto issue_k = [&](int stage, int t) {
    if (last_start >= 0) {
      const int start = t * KVB < last_start ? t * KVB : last_start;
      const char* tile = (const char*)kbase + (long long)start * p.k_ss * 2;
#pragma unroll
      for (int i = 0; i < PPW; ++i) glds16_sbase(tile, k_loff[i], lds0 + stage * TILE_BYTES + ((wave + i * NW) & 7) * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int piece = (wave + i * NW) & 7;
        const int kit = piece * 8 + prow;
        const int key = kit < skv ? kit : skv - 1;  // rows past the end re-read a valid row; their scores are masked
        glds16(kbase + (long long)key * p.k_ss + (ppos ^ ((kit >> 1) & 7)) * 8, smem + stage * TILE_BYTES + piece * 1024);
      }
    }
  };
  auto issue_v = [&](int stage, int t) {
    if (last_start >= 0) {
      const int start = t * KVB < last_start ? t * KVB : last_start;
      const char* tile = (const char*)vbase + (long long)start * p.v_ss * 2;
#pragma unroll
      for (int i = 0; i < PPW; ++i) glds16_sbase(tile, v_loff[i], lds0 + V_BASE + stage * TILE_BYTES + ((wave + i * NW) & 7) * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int piece = (wave + i * NW) & 7;
        const int kit = piece * 8 + prow;
        const int key = kit < skv ? kit : skv - 1;
        glds16(vbase + (long long)key * p.v_ss + (ppos ^ (((kit >> 1) & 1) << 2)) * 8, smem + V_BASE + stage * TILE_BYTES + piece * 1024);
      }
    }
  };

  // ---- fragment read addresses (bytes, relative to the tile base)
  // K (A operand of K.Q^T): lane reads row key = kb*32 + r32, 16-byte chunk (2 ks + hh) ^ ((key>>1)&7)
  ln.k_row_off = r32 * 128;
  ln.k_swz = (r32 >> 1) & 7;
  // V^T (A operand of V^T.P^T) through ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of
  // key row (k0 + q), columns 4p..4p+3 of the block; it receives column (lane & 15).
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  ln.v_lane_off = (4 * ln.hh + q4) * 128 + (g16 & 1) * 32 + p4 * 8;
  ln.v_half0 = (q4 >> 1) * 64;        // d-tile 0: 64-byte half index 0 ^ ((key >> 1) & 1)
  ln.v_half1 = (1 - (q4 >> 1)) * 64;  // d-tile 1
  {
    const unsigned kbase = (unsigned)(size_t)(smem + ln.k_row_off);
#pragma unroll
    for (int c = 0; c < 4; ++c) ln.kb[c] = kbase + ((2 * c + ln.hh) ^ ln.k_swz) * 16;
    ln.vc0 = (unsigned)(size_t)(smem + NS * TILE_BYTES + ln.v_lane_off + ln.v_half0);
    ln.vc1 = (unsigned)(size_t)(smem + NS * TILE_BYTES + ln.v_lane_off + ln.v_half1);
  }

  Run r;
#pragma unroll
  for (int i = 0; i < 16; ++i) { r.o0[i] = 0.f; r.o1[i] = 0.f; r.negm[i] = 0.f; r.lacc[i] = 0.f; }
  r.m = 0.f;

  // ---- main loop.  K and V each own a ring of NS stages; the DMA runs D = NS-1 tiles ahead of the compute and is
  // retired by a COUNTED vmcnt (every wave issues exactly 2*PPW DMA instructions per tile pair, tiles past the end
  // re-read clamped rows so the count never changes) + a raw s_barrier: __syncthreads() would drain the queue.
  const int nt = (skv + KVB - 1) / KVB;
  constexpr int D = NS - 1;   // without the half-tile stagger the stage refilled after barrier #t is the one read in iteration t-1
#ifdef MRAG_ATTN_STAMPS
  unsigned long long vm_wait = 0, bar_wait = 0;
#endif
  auto wait_pair = [&]() {   // all but the (D-1) youngest tile pairs of this wave have landed; then rendezvous
#ifdef MRAG_ATTN_STAMPS
    unsigned long long w0, w1, w2;
    MRAG_STAMP(w0);
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW * (D - 1)) : "memory");
#ifdef MRAG_ATTN_STAMPS
    MRAG_STAMP(w1);
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef MRAG_ATTN_STAMPS
    MRAG_STAMP(w2);
    vm_wait += w1 - w0; bar_wait += w2 - w1;
#endif
  };
  {
    // iteration t reads K(t), V(t) from stage t % NS.  Barrier #j guarantees tile j has landed for every wave; after it each
    // wave issues tile j + D into the stage of tile j - 2 (ring of NS = D + 2).  The "late" half of the workgroup (waves
    // NW/2..NW-1, the SIMD partners of the early half) runs HALF A TILE BEHIND: it takes barrier #j in the middle of its
    // softmax(j-1), so while one wave of a SIMD is in its MFMA phase (QK^T / PV) its partner is in its exp/convert phase,
    // instead of both queueing on the same pipe right after a common barrier.
    const bool late = false;
#pragma unroll
    for (int i = 0; i < D; ++i) { issue_k(i, i); issue_v(i, i); }
    if (late) { wait_pair(); issue_k(D % NS, D); issue_v(D % NS, D); }   // barrier #0
#ifdef MRAG_ATTN_STAMPS
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, ta, tb, tc, td, te, tg;
#endif
    auto iter = [&](int t, auto split_c, auto stage_c) {
      constexpr int STG = decltype(stage_c)::value;   // ring stage as a compile-time constant, or -1
#ifdef MRAG_ATTN_STAMPS
      MRAG_STAMP(ta);
#endif
      if (!late) wait_pair();   // barrier #t
#ifdef MRAG_ATTN_STAMPS
      MRAG_STAMP(tb);
#endif
      auto early_issue = [&]() {
        if (!late) { issue_k((t + D) % NS, t + D); issue_v((t + D) % NS, t + D); }
      };
      auto mid = [&]() {
#ifdef MRAG_ATTN_STAMPS
        MRAG_STAMP(td);
#endif
        if (late) { wait_pair(); issue_k((t + 1 + D) % NS, t + 1 + D); issue_v((t + 1 + D) % NS, t + 1 + D); }   // barrier #(t+1)
#ifdef MRAG_ATTN_STAMPS
        MRAG_STAMP(te);
#endif
      };
      if (!wave_active) { early_issue(); mid(); return; }
      f32x16 s0, s1;
      bf16x8 pb[4];
      if constexpr (STG >= 0) qk_tile_imm8<STG>(ln, qf, r.negm, s0, s1, early_issue);
      else
      qk_tile(smem + (t % NS) * TILE_BYTES, ln, qf, r.negm, s0, s1, early_issue);
#ifdef MRAG_ATTN_STAMPS
      MRAG_STAMP(tc);
#endif
      softmax_tile<HAS_MASK, false>(p, skv, ln, t, nt, qrow_c, s0, s1, s0, s1, r, pb, mid);
      if constexpr (STG >= 0) pv_tile_imm<STG>(ln, pb, r.o0, r.o1);
      else
      pv_tile(smem + V_BASE + (t % NS) * TILE_BYTES, ln, pb, r.o0, r.o1, r.lacc);
#ifdef MRAG_ATTN_STAMPS
      MRAG_STAMP(tg);
      acc_t[0] += tb - ta; acc_t[1] += tc - tb; acc_t[2] += td - tc; acc_t[3] += te - td; acc_t[4] += tg - te; acc_t[5] += tg - ta;
#endif
        };
    // full unmasked tiles take the per-block pipeline; the ragged last tile (and the masked instantiation) the one-softmax path,
    // in separate loops so that neither path's live state burdens the other
    const int n_split = 0;
    using RT = std::integral_constant<int, -1>;
    for (int t = 0; t < n_split; ++t) iter(t, std::true_type{}, RT{});
    int t = n_split;
    if constexpr (NS == 4 && NW == 8 && !HAS_MASK && !SHORTKV) {
      // unrolled by the ring depth: stage = t % NS is an immediate (n_split is 0 here, so t starts at a multiple of NS)
      for (; t + NS <= nt; t += NS) {
        iter(t, std::false_type{}, std::integral_constant<int, 0>{});
        iter(t + 1, std::false_type{}, std::integral_constant<int, 1>{});
        iter(t + 2, std::false_type{}, std::integral_constant<int, 2>{});
        iter(t + 3, std::false_type{}, std::integral_constant<int, 3>{});
      }
    }
    for (; t < nt; ++t) iter(t, std::false_type{}, RT{});
#ifdef MRAG_ATTN_STAMPS
    if (g_stamp_buf && lane == 0 && blockIdx.x < 2048) {
      for (int k = 0; k < 6; ++k) g_stamp_buf[((long long)blockIdx.x * NW + wave) * 8 + k] = acc_t[k];
      g_stamp_buf[((long long)blockIdx.x * NW + wave) * 8 + 6] = nt;
      g_stamp_buf[((long long)blockIdx.x * NW + wave) * 8 + 7] = (vm_wait << 32) | (bar_wait & 0xffffffffull);
    }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the clamped tail DMAs before the LDS is released

  if (!wave_active) return;
  // ---- epilogue: normalise by the MFMA-accumulated row sum (already complete over both half-waves' keys), fused residual
  if (qrow >= p.Sq) return;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r.lacc[0]), __float_as_uint(r.lacc[0]), false, false);
    r.lacc[0] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (KVSPLIT && split_unit) {   // partial result of this key chunk; attn_combine_kernel merges the chunks
    const long long prow_i = (long long)(blockIdx.x - p.n_main) * p.rem_rows + wave * 32 + r32;
    float* po = p.part_o + prow_i * 64 + 4 * ln.hh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = dt ? f32x4{r.o1[4 * g], r.o1[4 * g + 1], r.o1[4 * g + 2], r.o1[4 * g + 3]}
                           : f32x4{r.o0[4 * g], r.o0[4 * g + 1], r.o0[4 * g + 2], r.o0[4 * g + 3]};
        *(f32x4*)(po + dt * 32 + 8 * g) = v;
      }
    }
    if (ln.hh == 0) p.part_ml[prow_i] = make_float2(r.m, r.lacc[0]);
    return;
  }
  const float inv = p.out_scale / r.lacc[0];
  const long long obase = (long long)b * p.o_sb + (long long)qrow * p.o_ss + h * 64 + 4 * ln.hh;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (dt ? r.o1[4 * g + e] : r.o0[4 * g + e]) * inv;
      const long long off = obase + dt * 32 + 8 * g;
      if (p.resid) {
        const u32x2 rr = *(const u32x2*)(p.resid + off);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 out;
      out[0] = pack_bf2(v[0], v[1]);
      out[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.O + off) = out;
    }
  }
}


template <int NW, bool SHORTKV = false>
int launch_attn(hipStream_t s, AttnP p) {
  p.n_qtiles = (p.Sq + NW * 32 - 1) / (NW * 32);
  const dim3 grid(p.n_qtiles * p.B * p.H), block(NW * 64);
  const size_t lds = 2 * NS * TILE_BYTES;
  {
    const void* kf = (p.mask || p.bias) ? (const void*)attn_fwd_kernel<NW, true, SHORTKV> : (const void*)attn_fwd_kernel<NW, false, SHORTKV>;
    const hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (p.mask || p.bias) MRAG_LAUNCH((attn_fwd_kernel<NW, true, SHORTKV>), grid, block, lds, s, p);
  else MRAG_LAUNCH((attn_fwd_kernel<NW, false, SHORTKV>), grid, block, lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_FLASH);
  return MRAG_OK;
}

// ---------------------------------------------------------------------------------------------- key-split tail
// Long-sequence launches are many rounds of 512 resident workgroups (2 per CU) and the LAST round is what the ragged query tile of each
// (b, h) pair leaves: B*H workgroups that each scan ALL keys for Sq % 256 rows.  At the BASELINE shape (B*H = 96, Sq = 17 776 =
// 69 x 256 + 112) those 96 workgroups are 1.4 % of the launch's work and 4 % of its time (tools/tail_probe.py: 7.33 ms for the 6624
// full tiles, 7.63 ms with the ragged ones).  Splitting THEIR keys into `kv_splits` chunks turns the 96 long stragglers into ~512 short
// workgroups that drain into the slots the last full round frees; each leaves (unnormalised O, running max, row sum) in a caller-provided
// workspace and attn_combine_kernel merges the chunks, applies out_scale / resid and writes bf16 -- the same online-softmax algebra as
// between two key tiles, carried through HBM instead of registers.
}  // namespace

SplitPlan mrag_plan_kv_split(int B, int H, int Sq, int Skv, int tile_rows, int slots) {
  // tile_rows: query rows per workgroup (256: attn_flash.hip; 192: attn16.hip); slots: workgroups resident on the chip (2 or 3 per CU)
  SplitPlan pl;
  const int rem = Sq % tile_rows, n_full = Sq / tile_rows, nt = (Skv + KVB - 1) / KVB;
  const long long nbh = (long long)B * H;
  int want = (int)(slots / nbh);
  if (want > 8) want = 8;
  if (rem == 0 || nbh * n_full < 1024 || want < 2 || nt < 8 * want) return pl;
  const int tpc = (nt + want - 1) / want;
  const int chunk = tpc * KVB, splits = (Skv + chunk - 1) / chunk;
  if (splits < 2 || Skv - (splits - 1) * chunk < KVB) return pl;       // every chunk must hold one whole (slid-back) tile
  pl.splits = splits; pl.chunk_keys = chunk; pl.rem_rows = rem; pl.n_full = n_full;
  pl.bytes = (size_t)nbh * splits * rem * (64 * sizeof(float) + sizeof(float2));
  return pl;
}

namespace {

__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnP p) {
  const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);   // (b*H + h, row of the ragged tile); 16 lanes x 4 columns per row
  if (idx >= (long long)p.B * p.H * p.rem_rows) return;
  const int bh = (int)(idx / p.rem_rows), row = (int)(idx % p.rem_rows), d = (threadIdx.x & 15) * 4;
  const long long u0 = (long long)bh * p.kv_splits;
  float M = -INFINITY;
  for (int c = 0; c < p.kv_splits; ++c) M = fmaxf(M, p.part_ml[(u0 + c) * p.rem_rows + row].x);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int c = 0; c < p.kv_splits; ++c) {
    const long long pr = (u0 + c) * p.rem_rows + row;
    const float2 ml = p.part_ml[pr];
    const float w = __builtin_amdgcn_exp2f(ml.x - M);
    l += w * ml.y;
    acc += *(const f32x4*)(p.part_o + pr * 64 + d) * w;
  }
  const float inv = p.out_scale / l;
  const int b = bh / p.H, h = bh % p.H;
  const long long off = (long long)b * p.o_sb + (long long)(p.n_qtiles * p.tile_rows + row) * p.o_ss + h * 64 + d;
  float v[4] = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
  if (p.resid) {
    const u32x2 rr = *(const u32x2*)(p.resid + off);
    v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
    v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
  }
  u32x2 out;
  out[0] = pack_bf2(v[0], v[1]);
  out[1] = pack_bf2(v[2], v[3]);
  *(u32x2*)(p.O + off) = out;
}

int launch_attn_split(hipStream_t s, AttnP p, const SplitPlan& pl, void* workspace) {
  const int nbh = p.B * p.H;
  p.n_qtiles = pl.n_full;
  p.n_main = pl.n_full * nbh;
  p.kv_splits = pl.splits; p.chunk_keys = pl.chunk_keys; p.rem_rows = pl.rem_rows; p.tile_rows = 256;
  p.part_o = (float*)workspace;
  p.part_ml = (float2*)((char*)workspace + (size_t)nbh * pl.splits * pl.rem_rows * 64 * sizeof(float));
  const size_t lds = 2 * NS * TILE_BYTES;
  const void* kf = (const void*)attn_fwd_kernel<8, false, false, true>;
  const hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH((attn_fwd_kernel<8, false, false, true>), dim3(p.n_main + nbh * pl.splits), dim3(512), lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_FLASH_KSPLIT);
  MRAG_LAUNCH(attn_combine_kernel, dim3((unsigned)(((long long)nbh * pl.rem_rows + 15) / 16)), dim3(256), 0, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_COMBINE);
  return MRAG_OK;
}

}  // namespace

int mrag_launch_attn_combine(hipStream_t s, const AttnP& p) {
  MRAG_LAUNCH(attn_combine_kernel, dim3((unsigned)(((long long)p.B * p.H * p.rem_rows + 15) / 16)), dim3(256), 0, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_COMBINE);
  return MRAG_OK;
}

extern "C" int64_t mrag_attn_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Skv) {
  if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return 0;
  const size_t a = mrag_plan_kv_split(B, H, Sq, Skv, 256, 512).bytes, b = mrag_plan_kv_split(B, H, Sq, Skv, mrag_attn16_rows(mrag_attn16_qb(Sq)), mrag_attn16_slots(mrag_attn16_qb(Sq))).bytes;
  return (int64_t)(a > b ? a : b);      // either kernel family's tail plan fits
}

// ---------------------------------------------------------------------------------------------- tiny sequences
// Temporal attention of the UNets (TemporalTransformer / TemporalBasicTransformerBlock: 14-16 frames per pixel, head_dim 64, one
// (pixel, head) pair = 6 KB of Q/K/V) is pure data movement: the 64-key tiles of the flash kernel above would pad every pair 4-8x.
// Here one wavefront owns a pair: Q and K go from global memory straight into 16x16x32 MFMA operand registers (row = lane & 15,
// 16-byte chunk = lane >> 4), S^T = K.Q^T is 2 MFMAs, the softmax over <= 16 keys is 4 registers x 4 lane groups, P^T is re-laid into
// the B operand with two cross-lane moves, V is staged (2 KB, row pitch 144 B) in a per-wave LDS slot and read column-wise for the
// V^T operand of the four O^T = V^T.P^T MFMAs.  No barrier: every LDS byte is written and read by the same wave.
__global__ __launch_bounds__(256) void attn_tiny_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) char vsm[4][16 * 144];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  char* vs = vsm[wave];
  const int total = p.B * p.H;
  for (int pair = blockIdx.x * 4 + wave; pair < total; pair += gridDim.x * 4) {
    const int b = pair / p.H, h = pair - b * p.H, kb = b / p.kv_div;
    const bf16_t* Qp = p.Q + (long long)b * p.q_sb + (long long)h * p.q_sh;
    const bf16_t* Kp = p.K + (long long)kb * p.k_sb + (long long)h * p.k_sh;
    const bf16_t* Vp = p.V + (long long)kb * p.v_sb + (long long)h * p.v_sh;
    const int qrow = r16 < p.Sq ? r16 : p.Sq - 1, krow = r16 < p.Skv ? r16 : p.Skv - 1;   // clamped rows are masked / never stored
    const bf16x8 q0 = *(const bf16x8*)(Qp + (long long)qrow * p.q_ss + kq * 8), q1 = *(const bf16x8*)(Qp + (long long)qrow * p.q_ss + 32 + kq * 8);
    const bf16x8 k0 = *(const bf16x8*)(Kp + (long long)krow * p.k_ss + kq * 8), k1 = *(const bf16x8*)(Kp + (long long)krow * p.k_ss + 32 + kq * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) {                      // V: 16 rows x 8 chunks of 16 bytes
      const int c = lane + 64 * j, row = c >> 3, col = c & 7;
      const int vrow = row < p.Skv ? row : p.Skv - 1;
      *(u32x4*)(vs + row * 144 + col * 16) = *(const u32x4*)(Vp + (long long)vrow * p.v_ss + col * 8);
    }
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q0, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q1, st, 0, 0, 0);      // st[i] = S[q = r16][key = 4 kq + i]
    float sv[4], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sv[i] = (4 * kq + i < p.Skv) ? st[i] * p.qscale : -INFINITY;
      m = fmaxf(m, sv[i]);
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - m); l += sv[i]; }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const unsigned own0 = pack_bf2(sv[0], sv[1]), own1 = pack_bf2(sv[2], sv[3]);
    const unsigned a0 = __shfl(own0, lane + 16), a1 = __shfl(own1, lane + 16), c0 = __shfl(own0, lane + 32), c1 = __shfl(own1, lane + 32);
    u32x4 pw = {0u, 0u, 0u, 0u};                        // P^T as B operand: lane (q, kq) holds keys 8 kq .. 8 kq + 7
    if (kq == 0) pw = u32x4{own0, own1, a0, a1};
    else if (kq == 1) pw = u32x4{a0, a1, c0, c1};
    const bf16x8 pb = __builtin_bit_cast(bf16x8, pw);
    const float inv = p.out_scale / l;
    bf16_t* Op = p.O + (long long)b * p.o_sb + (long long)r16 * p.o_ss + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      u32x4 vw = {0u, 0u, 0u, 0u};                      // V^T as A operand: lane (d = 16 dt + r16, kq) holds keys 8 kq .. 8 kq + 7
      if (kq < 2) {
        const unsigned short* col = (const unsigned short*)(vs + (8 * kq) * 144 + (16 * dt + r16) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) vw[e] = (unsigned)col[(2 * e) * 72] | ((unsigned)col[(2 * e + 1) * 72] << 16);
      }
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vw), pb, o, 0, 0, 0);   // o[i] = O[q = r16][d = 16 dt + 4 kq + i]
      if (r16 < p.Sq) {
        u32x2 out = {pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv)};
        *(u32x2*)(Op + 16 * dt + 4 * kq) = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- folded motion-adapter branch
// hidden += scale * softmax(scores / 8) . V_ip with scores = hidden . M^T already produced by the GEMM, M = K_ip . W_q_ip per head
// (see include/mrag_hip.h: mrag_ip_attn_folded_bf16).  One wavefront owns 16 rows x one head at a time: the 32 (25 valid) scores of a
// (row, head) are 64 contiguous bytes = the 16x16x32 B-operand layout (row = lane & 15, 8 keys per lane >> 4) straight from global
// memory, the softmax is 8 registers x 4 lane groups, V^T of the block's heads sits transposed in LDS as the A operand, and
// O^T = V^T . P^T.  HBM-bound (546 MB per launch at the DiT shape: 109 MB of scores, 218 MB of `hidden` in and out), so the layout is chosen for the
// MEMORY side (round 5): the four MFMAs of a head take V^T rows in the PERMUTED feature order
//     d(m, t) = 8 (m >> 2) + 4 (t & 1) + (m & 3) + 32 (t >> 1)          m = A row of MFMA t
// so that lane (row, kq) ends up with features 8 kq .. 8 kq + 7 (MFMAs 0, 1) and 32 + 8 kq .. 32 + 8 kq + 7 (MFMAs 2, 3): `hidden` is read and written
// as TWO 16-byte accesses per lane, each instruction covering 64 contiguous bytes per row, instead of the four 8-byte accesses (32 contiguous bytes per
// row and instruction) the natural order d = 16 t + m gives -- half the vector-memory instructions and twice the bytes per cache-line request.  The V^T
// image is XOR-swizzled by (d >> 3) & 3 on its 16-byte key chunks: the permuted rows of one MFMA then hit 16 distinct bank groups (the plain
// image is 4-way conflicted for either order).
struct IpFoldP {
  const bf16_t* scores; const bf16_t* v; bf16_t* o;
  long long rows, s_ld, o_ld, v_bs, v_ks, rows_per_batch;
  int H, keys, kv_div, ks;   // ks: elements between the heads' score blocks (32: 64-byte aligned blocks; 26: the packed form of the one-launch score GEMM)
  float qscale, out_scale;
};

#ifndef MRAG_IPFOLD_HG
#define MRAG_IPFOLD_HG 4    // heads whose V^T image a workgroup keeps in LDS (4 KB each); MI355X, DiT shape: 16 -> 196 us, 8 -> 194, 4 -> 185 (more workgroups in flight)
#endif
__global__ __launch_bounds__(256) void ip_attn_folded_kernel(const IpFoldP p) {
  constexpr int HG = MRAG_IPFOLD_HG;                       // heads per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* vt = (bf16_t*)smem;                               // [HG][64 d][4 swizzled chunks of 8 keys]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int h0 = blockIdx.y * HG, kb = blockIdx.z;          // blockIdx.z = K/V batch (one V^T image per block)
  const int nh = min(HG, p.H - h0);
  for (int i = threadIdx.x; i < HG * 64 * 32 / 8; i += 256) ((u32x4*)vt)[i] = u32x4{0u, 0u, 0u, 0u};   // padding keys / heads
  __syncthreads();
  // Key SLOTS.  The four lanes of a (row, head) read the four ALIGNED 16-byte chunks that cover the head's score block: with the packed blocks (ks = 26: a
  // block starts every 52 bytes) the block begins s = (ks h) mod 8 elements into its first chunk, so key k of head h sits in MFMA slot k + s -- the contraction
  // does not care which slot a key occupies as long as the V^T image uses the same one, and the slots in front of / behind the block (neighbouring heads'
  // scores) are masked.  ks = 32: s = 0, the round-5 layout.  (The first packed form read 16 bytes from 4-byte aligned addresses: + 9 us per launch.)
  for (int i = threadIdx.x; i < p.keys * nh * 8; i += 256) {     // 16-byte chunk (key, head, 8 features) -> 8 transposed LDS elements
    const int c8 = i & 7, hl = (i >> 3) % nh, key = (i >> 3) / nh;
    const int slot = key + ((p.ks * (h0 + hl)) & 7);
    const u32x4 raw = *(const u32x4*)(p.v + (long long)kb * p.v_bs + (long long)key * p.v_ks + (h0 + hl) * 64 + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = c8 * 8 + e;                                  // (d >> 3) & 3 == c8 & 3
      vt[(hl * 64 + d) * 32 + ((((slot >> 3) ^ (c8 & 3)) << 3) | (slot & 7))] = (bf16_t)((e & 1) ? (raw[e >> 1] >> 16) : (raw[e >> 1] & 0xffffu));
    }
  }
  __syncthreads();
  // A-operand addresses of the four MFMAs: V^T row d(r16, t), key chunk kq (swizzled by (d >> 3) & 3 = r16 >> 2)
  const int dbase = 8 * (r16 >> 2) + (r16 & 3), kchunk = (kq ^ (r16 >> 2)) << 3;
  // this K/V batch covers q batches [kb * kv_div, (kb + 1) * kv_div): rows [row_lo, row_hi)
  const long long row_lo = (long long)kb * p.kv_div * p.rows_per_batch, row_hi = row_lo + (long long)p.kv_div * p.rows_per_batch;
  for (long long g0 = row_lo + ((long long)blockIdx.x * 4 + wave) * 16; g0 < row_hi; g0 += (long long)gridDim.x * 64) {
    const long long row = g0 + r16;
    const long long rc = row < row_hi ? row : row_hi - 1;
    // the row's scores and current values of the NEXT head are requested while this head is processed (latency-bound kernel)
    u32x4 raw_n = *(const u32x4*)(p.scores + rc * p.s_ld + ((h0 * p.ks) & ~7) + kq * 8);   // (nontemporal loads of the scores measured no different: profiles/r6_ip_attn_folded_packed_aligned.txt)
    u32x4 old_n[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) old_n[hf] = *(const u32x4*)(p.o + rc * p.o_ld + h0 * 64 + 32 * hf + 8 * kq);
    for (int hl = 0; hl < nh; ++hl) {
      const int h = h0 + hl;
      const u32x4 raw = raw_n;
      bf16_t* op = p.o + rc * p.o_ld + h * 64;
      const u32x4 old[2] = {old_n[0], old_n[1]};
      if (hl + 1 < nh) {
        raw_n = *(const u32x4*)(p.scores + rc * p.s_ld + (((h + 1) * p.ks) & ~7) + kq * 8);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) old_n[hf] = *(const u32x4*)(op + 64 + 32 * hf + 8 * kq);
      }
      float sv[8], m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sv[2 * e] = __uint_as_float(raw[e] << 16) * p.qscale;
        sv[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u) * p.qscale;
      }
      const int shift = (p.ks * h) & 7;                           // slot of key 0 (wave-uniform)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if ((unsigned)(kq * 8 + e - shift) >= (unsigned)p.keys) sv[e] = -INFINITY;
        m = fmaxf(m, sv[e]);
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sv[e] = __builtin_amdgcn_exp2f(sv[e] - m); l += sv[e]; }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const u32x4 pw = {pack_bf2(sv[0], sv[1]), pack_bf2(sv[2], sv[3]), pack_bf2(sv[4], sv[5]), pack_bf2(sv[6], sv[7])};
      const bf16x8 pb = __builtin_bit_cast(bf16x8, pw);
      const float inv = p.out_scale / l;
      const bf16_t* vh = vt + hl * 64 * 32 + kchunk;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {                            // features 32 hf + 8 kq .. + 7 of this lane's row
        u32x4 nw;
#pragma unroll
        for (int q = 0; q < 2; ++q) {                             // MFMA t = 2 hf + q: features 32 hf + 8 kq + 4 q .. + 3
          const bf16x8 va = *(const bf16x8*)(vh + (dbase + 4 * q + 32 * hf) * 32);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, acc, 0, 0, 0);     // acc[i] = out[row = r16][d = 32 hf + 8 kq + 4 q + i]
          const unsigned o0 = old[hf][2 * q], o1 = old[hf][2 * q + 1];
          nw[2 * q] = pack_bf2(__uint_as_float(o0 << 16) + acc[0] * inv, __uint_as_float(o0 & 0xffff0000u) + acc[1] * inv);
          nw[2 * q + 1] = pack_bf2(__uint_as_float(o1 << 16) + acc[2] * inv, __uint_as_float(o1 & 0xffff0000u) + acc[3] * inv);
        }
        if (row < row_hi) *(u32x4*)(op + 32 * hf + 8 * kq) = nw;
      }
    }
  }
}

extern "C" int mrag_ip_attn_folded_bf16(void* stream, const void* scores, const void* v, void* hidden, int32_t B, int64_t S, int32_t H, int32_t keys,
                                        int32_t kv_batch_div, int64_t scores_ld, int64_t hidden_ld, int64_t v_batch_stride, int64_t v_key_stride,
                                        float scale, float out_scale, int32_t key_stride) {
  if (!scores || !v || !hidden || B <= 0 || S <= 0 || H <= 0 || keys <= 0 || keys > 32 || kv_batch_div <= 0 || B % kv_batch_div) return MRAG_EINVAL;
  const int ks = key_stride == 0 ? 32 : key_stride;
  if (ks < keys || ks > 32) return MRAG_EINVAL;
  for (int h = 0; h < H && h < 8; ++h)                                            // a block's keys must fit the 32 slots of the aligned chunks that cover it ((ks h) mod 8 repeats with period <= 8)
    if (((ks * h) & 7) + keys > 32) return MRAG_EINVAL;
  if (scores_ld % 8 || hidden_ld % 8 || scores_ld < (((int64_t)(H - 1) * ks) & ~7LL) + 32 || hidden_ld < (int64_t)H * 64) return MRAG_EINVAL;   // a lane group reads the 32 elements from the aligned chunk that holds a block's start
  if (((uintptr_t)scores & 15) || ((uintptr_t)hidden & 15) || ((uintptr_t)v & 15) || v_batch_stride % 8 || v_key_stride % 8) return MRAG_EINVAL;
  IpFoldP p{};
  p.scores = (const bf16_t*)scores; p.v = (const bf16_t*)v; p.o = (bf16_t*)hidden;
  p.rows = (long long)B * S; p.s_ld = scores_ld; p.o_ld = hidden_ld; p.v_bs = v_batch_stride; p.v_ks = v_key_stride; p.rows_per_batch = S;
  p.H = H; p.keys = keys; p.kv_div = kv_batch_div; p.ks = ks; p.qscale = scale * 1.4426950408889634f; p.out_scale = out_scale;
  constexpr int HG = MRAG_IPFOLD_HG;
  const size_t lds = HG * 64 * 32 * sizeof(bf16_t);
  const long long groups = ((long long)kv_batch_div * S + 63) / 64;
  hipError_t e = hipFuncSetAttribute((const void*)ip_attn_folded_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  // ONE round of resident workgroups (each pays a V^T fill and then walks its share of the rows): the count the RUNTIME reports for this kernel's registers
  // and LDS.  Until round 6 the grid assumed eight workgroups per CU from the LDS size alone; the kernel's 72 VGPRs allow seven -- 2 040 workgroups on 1 792
  // slots, i.e. a second round for an eighth of them.
  static int wg_per_cu = 0;                                  // (a property of the code object; the benign race writes the same value)
  if (wg_per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)ip_attn_folded_kernel, 256, lds) != hipSuccess || n <= 0) n = 4;
#ifdef MRAG_IPFOLD_WG8          // developer A/B build: round 5's grid (eight per CU from the LDS size alone)
    n = (int)(128 * 1024 / lds);
#endif
    wg_per_cu = n;
  }
  const long long per = (256LL * wg_per_cu) / (((H + HG - 1) / HG) * (long long)(B / kv_batch_div));
  const unsigned gx = (unsigned)(groups < (per > 1 ? per : 1) ? groups : (per > 1 ? per : 1));
  MRAG_LAUNCH(ip_attn_folded_kernel, dim3(gx, (unsigned)((H + HG - 1) / HG), (unsigned)(B / kv_batch_div)), dim3(256), lds, (hipStream_t)stream, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_IP_ATTN_FOLDED);
  return MRAG_OK;
}

extern "C" int mrag_attn_fwd_bf16(void* stream, const mrag_attn_args* a) {
  if (!a || !a->Q || !a->K || !a->V || !a->O) return MRAG_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Skv <= 0 || a->kv_batch_div <= 0) return MRAG_EINVAL;
  if (a->B % a->kv_batch_div != 0) return MRAG_EINVAL;
  // 16-byte fragment / DMA loads and 8-byte stores
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V) & 15) return MRAG_EINVAL;
  if ((a->q_sb | a->q_ss | a->q_sh | a->k_sb | a->k_ss | a->k_sh | a->v_sb | a->v_ss | a->v_sh) % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)a->O & 7) || (a->o_sb | a->o_ss) % 4 != 0) return MRAG_EINVAL;
  if (a->resid && ((uintptr_t)a->resid & 7)) return MRAG_EINVAL;
  // the scalar-base LDS-DMA path folds a tile's per-lane K / V offsets (up to 63 rows) into 32-bit unsigned byte offsets
  if (a->k_ss < 0 || a->v_ss < 0 || a->k_ss > 0x1ffffff || a->v_ss > 0x1ffffff) return MRAG_ENOTSUP;
  AttnP p{};
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V;
  p.O = (bf16_t*)a->O; p.resid = (const bf16_t*)a->resid; p.mask = a->mask;
  p.bias = a->bias; p.bias_sh = a->bias_sh;
  if (a->bias && (((uintptr_t)a->bias & 3) || a->bias_sh < 0)) return MRAG_EINVAL;
  const bool masked = a->mask || a->bias;   // either one takes the 32x32x16 kernel's per-score path
  p.q_sb = a->q_sb; p.q_ss = a->q_ss; p.q_sh = a->q_sh;
  p.k_sb = a->k_sb; p.k_ss = a->k_ss; p.k_sh = a->k_sh;
  p.v_sb = a->v_sb; p.v_ss = a->v_ss; p.v_sh = a->v_sh;
  p.o_sb = a->o_sb; p.o_ss = a->o_ss;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv; p.kv_div = a->kv_batch_div;
  p.qscale = a->q_prescaled ? 1.0f : a->scale * 1.4426950408889634f;
  p.out_scale = a->out_scale;
  hipStream_t s = (hipStream_t)stream;
  if (a->Sq <= 16 && a->Skv <= 16 && !masked && !a->resid && !(a->tuning & MRAG_ATTN_TUNE_NO_TINY)) {   // temporal attention of the UNets
    const long long pairs = (long long)a->B * a->H;
    long long blocks = (pairs + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    MRAG_LAUNCH(attn_tiny_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_ATTN_TINY);
    return MRAG_OK;
  }
  if (a->Sq > 128 && a->Skv <= KVB) return launch_attn<8, true>(s, p);
  const bool legacy = (a->tuning & MRAG_ATTN_TUNE_LEGACY) != 0;
  const bool may_split = a->Sq > 128 && !masked && a->workspace;   // key-split tail for the ragged last query tile (plan_kv_split)
  if (may_split && ((uintptr_t)a->workspace & 15) != 0) return MRAG_EINVAL;
  if (a->Sq > 128 && !masked && !legacy) {        // long unmasked sequences: the 16x16x32 family (attn16.hip), 192-row workgroups
    SplitPlan pl;
    bool split = false;
    if (may_split) {
      pl = mrag_plan_kv_split(a->B, a->H, a->Sq, a->Skv, mrag_attn16_rows(mrag_attn16_qb(a->Sq)), mrag_attn16_slots(mrag_attn16_qb(a->Sq)));
      split = pl.splits > 1 && a->workspace_bytes >= (int64_t)pl.bytes;
    }
    const int rc = mrag_launch_attn16(s, p, split ? &pl : nullptr, a->workspace, mrag_attn16_qb(a->Sq));
    if (rc != MRAG_ENOTSUP) return rc;
  }
  if (may_split) {
    const SplitPlan pl = mrag_plan_kv_split(a->B, a->H, a->Sq, a->Skv, 256, 512);
    if (pl.splits > 1 && a->workspace_bytes >= (int64_t)pl.bytes) return launch_attn_split(s, p, pl, a->workspace);
  }
  if (a->Sq > 128) return launch_attn<8>(s, p);
  if (a->Sq > 32) return launch_attn<2>(s, p);
  return launch_attn<1>(s, p);
}
