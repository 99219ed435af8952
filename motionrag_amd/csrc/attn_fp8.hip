// attn_fp8.hip -- head_dim-64 attention with OCP e4m3 Q / K / V / P on the block-scaled fp8 MFMA (BASELINE config "DynamiCrafter-1024 UNet
// 16x576x1024 + CAMA, fp8 MFMA attention path"): the spatial self-attention of the UNets, F.scaled_dot_product_attention at
// src/projects/dynamicrafter/DynamiCrafter/lvdm/modules/attention.py:189 (9 216 tokens x 5 heads x 32 frames at level 0).
//
//   mrag_attn_fwd_fp8 takes the SAME arguments as mrag_attn_fwd_bf16 (bf16 Q / K / V views, bf16 O) plus a required workspace:
//     1. amax_kernel      |Q|, |K|, |V| maxima per (batch, head)                              ("per-head amax scales")
//     2. quant_kernel     e4m3 copies with power-of-two per-head scales, laid out for the MFMA operands:
//                           Q8 [B, H, Sq, 64]      Q * (scale * log2 e) * 2^y           (y: amax lands in (224, 448])
//                           K8 [B, H, Skv, 64]     K * 2^ek, 16-byte chunks of a row XORed by (key >> 2) & 3   (LDS bank swizzle, baked in)
//                           V8 [B, H, Skv/64, 64 d, 64 slots]  V * 2^ev, TRANSPOSED per 64-key tile, key slots in the order the score
//                                                  accumulators leave them, chunks XORed by (d >> 2) & 3
//     3. attn8_kernel     v_mfma_scale_f32_32x32x64_f8f6f4 for both products: 2x the bf16 MFMA rate and half the LDS / DMA bytes.
//        S^T = K8 . Q8^T * 2^-(y+ek)   the E8M0 scale operand of the MFMA undoes both power-of-two scales for free; C = -m
//        P'  = exp2(S') * 8            (the shift rides in m; it keeps small probabilities out of e4m3's subnormals and cancels in O / l)
//        O^T = V8^T . P8^T             a lane's 32 packed P' bytes of one 64-key tile ARE its B operand (no cross-lane movement)
//        O   = O^T / l * 2^-ev * out_scale
//      fp32 softmax with a LAZY running max: a row is centred on its first tile's maximum and moves only when a 64-key row sum of P' reaches e4m3's range.
//      Round 6: the row sums ride the matrix pipe -- l^T = ONES . P8^T, one more MFMA per 64 keys whose every output register is the lane's complete tile sum
//      (of the e4m3 values that multiply V) -- instead of 32 v_add per lane and sub-tile: the loop was bound by vector issue (32 v_exp + 32 v_add + 16 v_cvt_pk per
//      four 64-cycle MFMAs), now it issues 32 v_exp + 16 v_cvt_pk against five MFMAs; waves switch priority by phase as in attn16.hip.
// Operand maps were measured, not assumed: tools/exp/fp8_layout_probe.hip -> A[row = l & 31][k = 32 (l >> 5) + byte], B likewise, the lane's
// scale byte applies to its own 32 k's (profiles/r2_fp8_layout_probe.txt).
// Precision: e4m3 carries 3 mantissa bits: expect ~3-6 % relative Frobenius error against fp32 attention (tests state the tolerance);
// this path is opt-in (config #5) and never used for the bf16 headline workload.
#include "attn_common.h"
#include "../../include/mrag_hip.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

namespace {

constexpr int KT = 128;               // keys per LDS stage (two 64-key sub-tiles)
constexpr int STAGE_K = KT * 64;      // 8 KB of K8
constexpr int STAGE = 2 * STAGE_K;    // + 8 KB of V8
constexpr int NS8 = 4;
constexpr float kPShift = 2.0f;       // P' = 4 P: a re-centred row's 64-key tile sums to at most 256, below the trigger
constexpr float kBig8 = 448.0f;       // a 64-key row sum of P' at or above e4m3's largest finite value -> some P' may not have been representable: re-centre
#ifndef MRAG_ATTN8_SETPRIO
#define MRAG_ATTN8_SETPRIO 1          // wave priority by phase as in attn16.hip (score MFMAs 1, exp / convert block 0, P.V + row-sum MFMAs 2)
#endif

struct Fp8P {
  AttnP a;
  uint8_t* q8; uint8_t* k8; uint8_t* v8;
  unsigned* amax;        // [B * H][4]: fp32 bit patterns of max |Q|, |K|, |V| (atomicMax on the bits)
};

// largest e with amax * 2^e <= 448 (e4m3's largest finite value), clamped; amax == 0 -> 0
__device__ __forceinline__ int pow2_fit(float amax) {
  if (!(amax > 0.f)) return 0;
  int e = (int)floorf(log2f(448.0f / amax));
  if (ldexpf(amax, e) > 448.0f) --e;
  if (ldexpf(amax, e + 1) <= 448.0f) ++e;
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}

__global__ __launch_bounds__(256) void amax_kernel(const AttnP p, unsigned* amax) {
  const int which = blockIdx.z, bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const bf16_t* base; long long ss; int S;
  if (which == 0) { base = p.Q + (long long)b * p.q_sb + (long long)h * p.q_sh; ss = p.q_ss; S = p.Sq; }
  else if (which == 1) { base = p.K + (long long)b * p.k_sb + (long long)h * p.k_sh; ss = p.k_ss; S = p.Skv; }
  else { base = p.V + (long long)b * p.v_sb + (long long)h * p.v_sh; ss = p.v_ss; S = p.Skv; }
  unsigned m = 0;
  for (int row = blockIdx.x * 32 + (threadIdx.x >> 3); row < S; row += gridDim.x * 32) {
    const u32x4 v = *(const u32x4*)(base + (long long)row * ss + (threadIdx.x & 7) * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = (v[i] << 16) & 0x7fff0000u, hi = v[i] & 0x7fff0000u;   // |x| as fp32 bits: integer order == float order
      m = lo > m ? lo : m;
      m = hi > m ? hi : m;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + bh * 4 + which, m);
}

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0u, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

// one workgroup = 64 rows x 64 features of Q, K or V of one (b, h): thread t holds row t / 4, features 16 (t % 4) .. + 15
__global__ __launch_bounds__(256) void quant_kernel(const Fp8P fp) {
  __shared__ __attribute__((aligned(16))) uint8_t vt[64 * 64];
  const AttnP& p = fp.a;
  const int which = blockIdx.z, bh = blockIdx.y, b = bh / p.H, h = bh % p.H, tile = blockIdx.x;
  const int S = which == 0 ? p.Sq : p.Skv;
  if (tile * 64 >= S) return;
  const float aq = __uint_as_float(fp.amax[bh * 4 + 0]), ak = __uint_as_float(fp.amax[bh * 4 + 1]), av = __uint_as_float(fp.amax[bh * 4 + 2]);
  float mul;
  if (which == 0) mul = ldexpf(p.qscale, pow2_fit(aq * p.qscale));     // Q * (scale log2 e) * 2^y
  else if (which == 1) mul = ldexpf(1.0f, pow2_fit(ak));
  else mul = ldexpf(1.0f, pow2_fit(av));
  const int t = threadIdx.x, row = t >> 2, seg = t & 3, grow = tile * 64 + row;
  const bf16_t* src;
  if (which == 0) src = p.Q + (long long)b * p.q_sb + (long long)h * p.q_sh + (long long)(grow < S ? grow : S - 1) * p.q_ss;
  else if (which == 1) src = p.K + (long long)b * p.k_sb + (long long)h * p.k_sh + (long long)grow * p.k_ss;
  else src = p.V + (long long)b * p.v_sb + (long long)h * p.v_sh + (long long)grow * p.v_ss;
  float f[16];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32x4 v = *(const u32x4*)(src + seg * 16 + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[8 * i + 2 * j] = __uint_as_float(v[j] << 16) * mul;
      f[8 * i + 2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u) * mul;
    }
  }
  if (which != 2) {
    const u32x4 o = {pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]), pack4_fp8(f[8], f[9], f[10], f[11]), pack4_fp8(f[12], f[13], f[14], f[15])};
    uint8_t* dst = (which == 0 ? fp.q8 : fp.k8) + ((long long)bh * S + grow) * 64;
    const int chunk = which == 0 ? seg : (seg ^ ((row >> 2) & 3));     // K8: LDS bank swizzle baked into the row
    if (grow < S) *(u32x4*)(dst + chunk * 16) = o;
    return;
  }
  // V: transpose the 64-key tile to [d][slot]; slot of key k: lane half hh = (k >> 2) & 1 takes bytes 32 hh .., block k >> 5 its 16-byte half,
  // register r = (k & 3) + 4 ((k & 31) >> 3) -- the order in which the 32x32 score accumulators hold a lane's keys
  const int k = row, kk = k & 31;
  const int slot = ((kk >> 2) & 1) * 32 + (k >> 5) * 16 + (kk & 3) + 4 * (kk >> 3);
  const unsigned w[4] = {pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]), pack4_fp8(f[8], f[9], f[10], f[11]), pack4_fp8(f[12], f[13], f[14], f[15])};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = seg * 16 + i;
    const int pos = (((slot >> 4) ^ ((d >> 2) & 3)) << 4) | (slot & 15);
    vt[d * 64 + pos] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
  }
  __syncthreads();
  *(u32x4*)(fp.v8 + ((long long)bh * S + tile * 64) * 64 + t * 16) = *(const u32x4*)(vt + t * 16);
}

__device__ __forceinline__ float half_swap_max8(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max3_asm(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
}

// 8 waves x 32 query rows; Skv % 128 == 0; no mask; K/V batch == Q batch.
__global__ __launch_bounds__(512, 4) void attn8_kernel(const Fp8P fp) {
  const AttnP& p = fp.a;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  const int nbh = p.B * p.H;
  int bh, qt;
  if ((nbh & 7) == 0) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    bh = (j / p.n_qtiles) * 8 + x;
    qt = j % p.n_qtiles;
  } else {
    bh = blockIdx.x / p.n_qtiles;
    qt = blockIdx.x % p.n_qtiles;
  }
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qt * 256 + wave * 32;
  const bool wave_active = q0 < p.Sq;
  const int qrow = q0 + r32, qc = qrow < p.Sq ? qrow : p.Sq - 1;

  // per-head power-of-two scales (the same arithmetic as quant_kernel)
  const float aq = __uint_as_float(fp.amax[bh * 4 + 0]), ak = __uint_as_float(fp.amax[bh * 4 + 1]), av = __uint_as_float(fp.amax[bh * 4 + 2]);
  const int y = pow2_fit(aq * p.qscale), ek = pow2_fit(ak), ev = pow2_fit(av);
  const int sb = 127 - (y + ek);                         // E8M0 exponent of the Q-side scale: the MFMA multiplies by 2^-(y + ek)
  const int scale_q = __builtin_amdgcn_readfirstlane((sb & 0xff) * 0x01010101), scale_one = 0x7f7f7f7f;

  const i32x8 qf = *(const i32x8*)(fp.q8 + ((long long)bh * p.Sq + qc) * 64 + hh * 32);

  // LDS-DMA: a stage = 128 keys = 8 KB of K8 + 8 KB of V8, both stored in their LDS image order: 16 linear 1-KiB pieces, 2 per wave
  const char* k8 = (const char*)fp.k8 + (long long)bh * p.Skv * 64;
  const char* v8 = (const char*)fp.v8 + (long long)bh * p.Skv * 64;
  const unsigned loff = wave * 1024 + lane * 16;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int nt = p.Skv / KT;
  auto issue_kv = [&](int stage, int t) {
    const int tt = t < nt ? t : nt - 1;                 // prefetches past the end re-read the last tile (never consumed)
    glds16_sbase(k8 + (long long)tt * STAGE_K, loff, lds0 + stage * STAGE + wave * 1024);
    glds16_sbase(v8 + (long long)tt * STAGE_K, loff, lds0 + stage * STAGE + STAGE_K + wave * 1024);
  };
  // fragment read addresses: row (key or d) = r32 (+ 32 per block), bytes 32 hh .. + 31 as two 16-byte chunks XORed by (row >> 2) & 3
  const int swz = (r32 >> 2) & 3;
  const unsigned fa0 = lds0 + r32 * 64 + (((2 * hh) ^ swz) << 4), fa1 = lds0 + r32 * 64 + (((2 * hh + 1) ^ swz) << 4);

  f32x16 o0, o1, negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; negm[i] = 0.f; }
  float m = 0.f, l = 0.f;
  const i32x8 ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};   // e4m3 1.0 in every byte: the A operand of the row-sum MFMA

  constexpr int D = NS8 - 1;
  auto wait_pair = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int i = 0; i < D; ++i) issue_kv(i, i);

  auto sub_tile = [&](int t, int sub, auto stage_c, auto sub_c) {
    constexpr int OFF = decltype(stage_c)::value * STAGE + decltype(sub_c)::value * 4096;
    f32x16 s0, s1;
    bool recentre = (t == 0 && sub == 0);
    float tile_sum;
    i32x8 pb;
    for (int pass = 0;; ++pass) {
      u32x4 ka[2], kb[2];
      asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %5 offset:%7\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(ka[0]), "=&v"(ka[1]), "=&v"(kb[0]), "=&v"(kb[1])
                   : "v"(fa0), "v"(fa1), "n"(OFF), "n"(OFF + 2048) : "memory");
      const i32x8 kf0 = {(int)ka[0][0], (int)ka[0][1], (int)ka[0][2], (int)ka[0][3], (int)ka[1][0], (int)ka[1][1], (int)ka[1][2], (int)ka[1][3]};
      const i32x8 kf1 = {(int)kb[0][0], (int)kb[0][1], (int)kb[0][2], (int)kb[0][3], (int)kb[1][0], (int)kb[1][1], (int)kb[1][2], (int)kb[1][3]};
      s0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf0, qf, negm, 0, 0, 0, scale_one, 0, scale_q);
      s1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf1, qf, negm, 0, 0, 0, scale_one, 0, scale_q);
      if (recentre) {
        // plain fmaxf, NOT the inline-asm v_max3 helper: hipcc pads the MFMA -> VALU read hazard (16-pass MFMA: many wait states) only for
        // instructions it can see; an asm statement reading s0 / s1 straight after the MFMAs read stale registers (found the hard way)
        float ma = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) ma = fmaxf(ma, fmaxf(s0[i], s1[i]));
        const float tm = half_swap_max8(ma);
        const bool first = (t == 0 && sub == 0);
        // S' is relative to m: the row moves so that its largest score of this tile maps to P' = 2^kPShift
        const float delta = first ? fmaxf(tm, -1e30f) - kPShift : fmaxf(tm - kPShift, 0.f);
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        m += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) { negm[i] = -m; s0[i] -= delta; s1[i] -= delta; }
      }
#if MRAG_ATTN8_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s0[i] = __builtin_amdgcn_exp2f(s0[i]);
        s1[i] = __builtin_amdgcn_exp2f(s1[i]);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        pb[v] = (int)pack4_fp8(s0[4 * v], s0[4 * v + 1], s0[4 * v + 2], s0[4 * v + 3]);
        pb[4 + v] = (int)pack4_fp8(s1[4 * v], s1[4 * v + 1], s1[4 * v + 2], s1[4 * v + 3]);
      }
#if MRAG_ATTN8_SETPRIO
      __builtin_amdgcn_s_setprio(2);
#endif
      // l^T[., q] = ONES . P8^T: every register of lane (q, .) is the sum of the tile's 64 e4m3 values of its query (the values that multiply V below)
      f32x16 zero16;
#pragma unroll
      for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
      const f32x16 lt = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones8, pb, zero16, 0, 0, 0, scale_one, 0, scale_one);
      tile_sum = lt[0];
      const bool blown = !(tile_sum < kBig8);                        // a saturated (or non-finite) P' makes the sum reach 448; a re-centred tile sums to <= 256
      if (__builtin_expect(!__any(blown), 1) || recentre) break;
      recentre = true;
#if MRAG_ATTN8_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
    }
    l += tile_sum;
    u32x4 va[2], vb[2];
    asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %5 offset:%7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(va[0]), "=&v"(va[1]), "=&v"(vb[0]), "=&v"(vb[1])
                 : "v"(fa0), "v"(fa1), "n"(OFF + STAGE_K), "n"(OFF + STAGE_K + 2048) : "memory");
    const i32x8 vf0 = {(int)va[0][0], (int)va[0][1], (int)va[0][2], (int)va[0][3], (int)va[1][0], (int)va[1][1], (int)va[1][2], (int)va[1][3]};
    const i32x8 vf1 = {(int)vb[0][0], (int)vb[0][1], (int)vb[0][2], (int)vb[0][3], (int)vb[1][0], (int)vb[1][1], (int)vb[1][2], (int)vb[1][3]};
    o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf0, pb, o0, 0, 0, 0, scale_one, 0, scale_one);
    o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf1, pb, o1, 0, 0, 0, scale_one, 0, scale_one);
#if MRAG_ATTN8_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
  };
  auto iter = [&](int t, auto stage_c) {
    constexpr int STG = decltype(stage_c)::value;
    wait_pair();
    issue_kv((STG + D) % NS8, t + D);
    if (!wave_active) return;
    sub_tile(t, 0, stage_c, std::integral_constant<int, 0>{});
    sub_tile(t, 1, stage_c, std::integral_constant<int, 1>{});
  };
  int t = 0;
  for (; t + NS8 <= nt; t += NS8) {
    iter(t, std::integral_constant<int, 0>{});
    iter(t + 1, std::integral_constant<int, 1>{});
    iter(t + 2, std::integral_constant<int, 2>{});
    iter(t + 3, std::integral_constant<int, 3>{});
  }
  if (t < nt) { iter(t, std::integral_constant<int, 0>{}); ++t; }
  if (t < nt) { iter(t, std::integral_constant<int, 1>{}); ++t; }
  if (t < nt) { iter(t, std::integral_constant<int, 2>{}); ++t; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!wave_active || qrow >= p.Sq) return;

  // (l is complete in every lane: the row-sum MFMA contracts over all 64 keys of a tile, both lane halves included)
  const float inv = ldexpf(p.out_scale, -ev) / l;
  // D of O^T = V^T . P^T: lane (query r32, half hh) holds d = 32 db + (i & 3) + 8 (i >> 2) + 4 hh
  const long long obase = (long long)b * p.o_sb + (long long)qrow * p.o_ss + h * 64 + 4 * hh;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (db ? o1[4 * g + e] : o0[4 * g + e]) * inv;
      const long long off = obase + db * 32 + 8 * g;
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

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int64_t mrag_attn_fp8_workspace_bytes(int32_t B, int32_t H, int32_t Sq, int32_t Skv) {
  if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return 0;
  const size_t bh = (size_t)B * H;
  return (int64_t)(align256(bh * 16) + align256(bh * Sq * 64) + 2 * align256(bh * Skv * 64));
}

extern "C" int mrag_attn_fwd_fp8(void* stream, const mrag_attn_args* a) {
  if (!a || !a->Q || !a->K || !a->V || !a->O || !a->workspace) return MRAG_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Skv <= 0) return MRAG_EINVAL;
  if (a->mask || a->kv_batch_div != 1 || a->Skv % KT != 0 || a->Skv < 4 * KT || a->q_prescaled) return MRAG_ENOTSUP;
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->workspace) & 15) return MRAG_EINVAL;
  if ((a->q_sb | a->q_ss | a->q_sh | a->k_sb | a->k_ss | a->k_sh | a->v_sb | a->v_ss | a->v_sh) % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)a->O & 7) || (a->o_sb | a->o_ss) % 4 != 0 || (a->resid && ((uintptr_t)a->resid & 7))) return MRAG_EINVAL;
  if (a->workspace_bytes < mrag_attn_fp8_workspace_bytes(a->B, a->H, a->Sq, a->Skv)) return MRAG_EINVAL;
  Fp8P fp{};
  AttnP& p = fp.a;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V;
  p.O = (bf16_t*)a->O; p.resid = (const bf16_t*)a->resid;
  p.q_sb = a->q_sb; p.q_ss = a->q_ss; p.q_sh = a->q_sh;
  p.k_sb = a->k_sb; p.k_ss = a->k_ss; p.k_sh = a->k_sh;
  p.v_sb = a->v_sb; p.v_ss = a->v_ss; p.v_sh = a->v_sh;
  p.o_sb = a->o_sb; p.o_ss = a->o_ss;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv; p.kv_div = 1;
  p.qscale = a->scale * 1.4426950408889634f;
  p.out_scale = a->out_scale;
  p.n_qtiles = (a->Sq + 255) / 256;
  const size_t bh = (size_t)a->B * a->H;
  char* ws = (char*)a->workspace;
  fp.amax = (unsigned*)ws; ws += align256(bh * 16);
  fp.q8 = (uint8_t*)ws; ws += align256(bh * a->Sq * 64);
  fp.k8 = (uint8_t*)ws; ws += align256(bh * a->Skv * 64);
  fp.v8 = (uint8_t*)ws;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(fp.amax, 0, bh * 16, s);
  if (e != hipSuccess) return (int)e;
  const int smax = a->Sq > a->Skv ? a->Sq : a->Skv;
  int chunks = (smax + 31) / 32;
  if (chunks > 64) chunks = 64;
  MRAG_LAUNCH(amax_kernel, dim3(chunks, (unsigned)bh, 3), dim3(256), 0, s, p, fp.amax);
  MRAG_LAUNCH_CHECK();
  MRAG_LAUNCH(quant_kernel, dim3((smax + 63) / 64, (unsigned)bh, 3), dim3(256), 0, s, fp);
  MRAG_LAUNCH_CHECK();
  const size_t lds = NS8 * STAGE;
  e = hipFuncSetAttribute((const void*)attn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH(attn8_kernel, dim3(p.n_qtiles * (unsigned)bh), dim3(512), lds, s, fp);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_FP8);
  return MRAG_OK;
}
