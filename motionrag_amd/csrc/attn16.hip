// attn16.hip -- head_dim-64 bf16 flash attention for LONG UNMASKED sequences (the 17 776-token CogVideoX joint attention,
// src/projects/condition/attn_processor.py:233-235; the spatial self-attention of the UNets, lvdm/modules/attention.py:189) on
// v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel family.  attn_flash.hip (v_mfma_f32_32x32x16_bf16, a lane owns a query row) is bound on this chip by power and by vector
// issue slots.  Here:
//   * the 16x16x32 shape -- the guide measures 1.12-1.15x the FLOP/s of 32x32x16 in power-limited loops at equal cycles per FLOP
//     (MI355X_MICROARCH.md, DVFS give-back item 7);
//   * (round 4) the OPTIMISTIC sweep: no running maximum at all -- P = exp2(S) straight from the score MFMAs, row sums on the matrix pipe
//     (l^T += ONES . P^T), one range check per sweep; a workgroup whose check fails repeats the pass in the SAFE form (rounds 2-3: the first
//     tile's maximum carried into the score MFMAs as their C operand, per-tile partial sums, exact re-centring of a tile whose sum explodes).
//     The fast loop issues per 16 x 64 scores exactly 16 v_exp + 8 v_cvt_pk + 18 MFMAs; see the comment above `sweep` in the kernel;
//   * (round 4) wave priorities by phase (s_setprio: score MFMAs 1, exp block 0, P.V MFMAs 2) so that the three waves of a SIMD feed the
//     matrix pipe first and exponentiate in its shadow.
//
// Layout (QB = 16-query blocks per wave, 2 -> 32 rows per wave, 256 per 8-wave workgroup):
//   S^T[key, q] = K . Q^T   A = K fragment  (row = key l & 15, k = d 32 ks + 8 (l >> 4) + j)   <- ds_read_b128 of the XOR-swizzled K tile
//                           B = Q^T fragment (col = q   l & 15, same k)                          <- registers, pre-multiplied by scale * log2 e
//                           C = 0 (fast sweep) | (-m, -m, -m, -m) of the lane's query (safe sweep)
//     D: lane (q = l & 15, g = l >> 4) holds keys 16 kb + 4 g + r  (r = register 0..3)
//   O^T[d, q] = V^T . P^T   A = V^T fragment (row = d 16 db + (l & 15), k-slot 8 g + j)         <- 2 x ds_read_b64_tr_b16 (keys 4 g.., 16 + 4 g..)
//                           B = P^T fragment: k-slot 8 g + j = key 32 s + 4 g + j (j < 4) | 32 s + 16 + 4 g + j - 4 -- exactly the packed
//                               accumulators of key blocks 2 s and 2 s + 1: P never leaves registers and needs no cross-lane move
//     D: lane (q, g) holds d = 16 db + 4 g + r -> 8-byte stores.
// K / V tiles of 64 keys ride a 3-stage LDS-DMA ring like attn_flash.hip's (scalar-base global_load_lds, counted vmcnt, raw s_barrier);
// V's 32-byte chunks are XOR-swizzled by (key >> 1) & 3 on the DMA source so the transposed reads are conflict-free.
#include "attn_common.h"
#include "../../include/mrag_hip.h"

namespace {

constexpr int KVB = 64;                // keys per compute tile
constexpr int TILE_BYTES = KVB * 128;
// LDS: K ring of NS stages, then the V ring
constexpr float kBig = 16777216.0f;    // lazy-max trigger: a lane's partial row sum of one tile above 2^24


// K fragment reads per group of the score MFMAs: 2 (one key block: 8 VGPRs of fragments -- what lets the optimistic sweep's row-sum accumulators fit
// the 168 VGPRs of three workgroups per CU) or 4 (two key blocks: faster where registers allow, i.e. at QB = 4).  MRAG_ATTN16_KGROUP pins one for A/B builds.
#ifdef MRAG_ATTN16_KGROUP
#define KGROUP_OF(QB) MRAG_ATTN16_KGROUP
#else
#define KGROUP_OF(QB) ((QB) == 3 ? 2 : 4)
#endif
// Wave priority by phase (fast sweep): three workgroups per CU put three waves on every SIMD, each cycling through score MFMAs -> a vector-only block
// (48 v_exp + 24 v_cvt_pk, ~480 issue cycles) -> P.V MFMAs.  With equal priorities the oldest wave wins the issue port, whatever it is doing; with
// s_setprio (QK, EXP, PV) = (1, 0, 2) a wave in its exp block yields to waves that have MFMAs to issue, so the matrix pipe is fed first and the
// exponentials fill the MFMAs' shadow.  Same-box interleaved, inside the denoise step (profiles/r4_attn_step_ab4.txt): 6.38 -> 6.04 ms per launch,
// 565.6 -> 551.9 ms per step; (1,0,2) = (2,0,3) = (1,0,3) = (2,1,3); exp at the QK level (1,1,2) keeps only a fifth of the gain; exp ABOVE the MFMA
// phases loses 7 %.  -DMRAG_ATTN16_SETPRIO=0 builds the loop without priorities.
#ifndef MRAG_ATTN16_SETPRIO
#define MRAG_ATTN16_SETPRIO 1
#endif
#ifndef MRAG_ATTN16_PRIO_QK
#define MRAG_ATTN16_PRIO_QK 1
#define MRAG_ATTN16_PRIO_EXP 0
#define MRAG_ATTN16_PRIO_PV 2
#endif
#ifndef MRAG_ATTN16_OPTIMISTIC
#define MRAG_ATTN16_OPTIMISTIC 1       // 0: every pass in the safe (checked) form -- the round-3 loop, kept buildable for A/B runs (tools/build_variant.sh)
#endif

struct Lane16 {
  unsigned ka[2];     // LDS byte address (stage 0, key block 0) of this lane's K fragment for k-step 0 / 1
  unsigned va[4];     // LDS byte address (stage 0, keys 4 g + q) of this lane's V^T read for d-block 0..3
  int g;
};

// 8 K fragments of one 64-key tile -> 8 QB score MFMAs, in two groups of 4 reads: one group (16 VGPRs of fragments) is live at a time.
// (Requesting both groups up front -- 150 -> 168 VGPRs forced -- lost 4 % when it cost the third workgroup per CU; pruned in round 3.)
template <int OFF, int QB, typename Between>
__device__ __forceinline__ void qk16(const Lane16& ln, const bf16x8 (&qf)[QB][2], const f32x4 (&negm)[QB], f32x4 (&s)[4][QB], Between between) {
  if constexpr (KGROUP_OF(QB) == 2) {
  // one key block (2 fragment reads, 2 QB MFMAs) at a time: 8 VGPRs of K fragments live instead of 16 -- the room the row-sum accumulators of the
  // optimistic sweep need at 168 VGPRs (three workgroups per CU)
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    u32x4 kf[2];
    if (kb == 0) {
      asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" : "=&v"(kf[0]), "=&v"(kf[1]) : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF) : "memory");
      between();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]) :: "memory");
    } else if (kb == 1) {
      asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(kf[0]), "=&v"(kf[1]) : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF + 2048) : "memory");
    } else if (kb == 2) {
      asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(kf[0]), "=&v"(kf[1]) : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF + 4096) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(kf[0]), "=&v"(kf[1]) : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF + 6144) : "memory");
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[0]), qf[qb][0], negm[qb], 0, 0, 0);
      s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[1]), qf[qb][1], s[kb][qb], 0, 0, 0);
    }
  }
  } else {
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // key blocks (0, 1), then (2, 3): 4 fragment reads, 4 QB MFMAs each
    u32x4 kf[4];                           // [2 * kbl + ks]
    if (half == 0) {
      asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %5 offset:%7"
                   : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3])
                   : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF), "n"(OFF + 2048) : "memory");
      between();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]) :: "memory");
    } else {
      asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %5 offset:%7\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(kf[0]), "=&v"(kf[1]), "=&v"(kf[2]), "=&v"(kf[3])
                   : "v"(ln.ka[0]), "v"(ln.ka[1]), "n"(OFF + 4096), "n"(OFF + 6144) : "memory");
    }
#pragma unroll
    for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s[2 * half + kbl][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[2 * kbl]), qf[qb][0], negm[qb], 0, 0, 0);
        s[2 * half + kbl][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[2 * kbl + 1]), qf[qb][1], s[2 * half + kbl][qb], 0, 0, 0);
      }
  }
  }
}

// LSUM: the row sums ride the matrix pipe too -- l^T[., q] += ONES . P^T, one more MFMA per 32-key step and query block whose A operand is the
// constant all-ones fragment: every register of lane (q, g) then holds the COMPLETE sum over the tile's keys of the bf16 P values that multiply V
// (2 MFMAs = 16 issue cycles per 16 x 64 scores instead of 8 v_dot2c + 2 v_add on the busier vector pipe).
template <int VOFF, int QB, bool LSUM = false>
__device__ __forceinline__ void pv16(const Lane16& ln, const bf16x8 (&pb)[2][QB], f32x4 (&o)[4][QB], f32x4 (*lacc)[QB] = nullptr) {
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    u32x2 lo[4], hi[4];   // d-block db: keys 32 st + 4 g .. (+3) and 32 st + 16 + 4 g .. (+3)
    if (st == 0)
      asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %1, %9 offset:%12\n\tds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %2, %10 offset:%12\n\tds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %3, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3])
                   : "v"(ln.va[0]), "v"(ln.va[1]), "v"(ln.va[2]), "v"(ln.va[3]), "n"(VOFF), "n"(VOFF + 2048) : "memory");
    else
      asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %1, %9 offset:%12\n\tds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %2, %10 offset:%12\n\tds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
                   "ds_read_b64_tr_b16 %3, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3])
                   : "v"(ln.va[0]), "v"(ln.va[1]), "v"(ln.va[2]), "v"(ln.va[3]), "n"(VOFF + 4096), "n"(VOFF + 6144) : "memory");
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const u32x4 w = {lo[db][0], lo[db][1], hi[db][0], hi[db][1]};
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), pb[st][qb], o[db][qb], 0, 0, 0);
    }
    if constexpr (LSUM) {
      const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) (*lacc)[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), pb[st][qb], (*lacc)[qb], 0, 0, 0);
    }
  }
}

struct NoHook16 {
  __device__ __forceinline__ void operator()() const {}
};

template <int QB, int NW, int NS, bool KVSPLIT>
__global__ __launch_bounds__(NW * 64, QB == 3 ? 3 : (QB == 4 ? 2 : 4)) void attn16_kernel(const AttnP p) {
  constexpr int ROWS = NW * QB * 16;
  constexpr int SK = KVB, STAGE_BYTES = TILE_BYTES, V_BASE = NS * STAGE_BYTES;      // an LDS stage = one 64-key compute tile; one barrier per stage
  constexpr int PPW = 8 / NW;                                                       // 1-KiB DMA pieces per wave per K (and V) stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15;
  Lane16 ln;
  ln.g = lane >> 4;

  // ---- XCD-aware block -> (q-tile, b, h): all q-tiles of one (b, h) on one XCD (its K / V stream from that XCD's L2)
  const int nbh = p.B * p.H;
  int bh, qt;
  int skv = p.Skv, key0 = 0;
  bool split_unit = false;
  if (KVSPLIT && (int)blockIdx.x >= p.n_main) {
    const int u = blockIdx.x - p.n_main;
    bh = u / p.kv_splits;
    qt = p.n_qtiles;
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
  const int q0 = qt * ROWS + wave * (QB * 16);
  const bool wave_active = q0 < p.Sq;

  // ---- Q^T fragments (B operand): lane holds Q[q = 16 qb + r16][d = 32 ks + 8 g + j], pre-multiplied by scale * log2 e
  bf16x8 qf[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + qb * 16 + r16;
    const int qc = qrow < p.Sq ? qrow : p.Sq - 1;
    const bf16_t* qp = p.Q + (long long)b * p.q_sb + (long long)qc * p.q_ss + (long long)h * p.q_sh + ln.g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 raw = *(const u32x4*)(qp + ks * 32);
      if (p.qscale == 1.0f) {
        qf[qb][ks] = __builtin_bit_cast(bf16x8, raw);
      } else {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pack_bf2(__uint_as_float(raw[i] << 16) * p.qscale, __uint_as_float(raw[i] & 0xffff0000u) * p.qscale);
        qf[qb][ks] = __builtin_bit_cast(bf16x8, r);
      }
    }
  }

  // ---- LDS-DMA staging: one 1-KiB piece (8 keys x 128 B) per wave per K (and V) tile; lane i -> key (i >> 3), 16-byte granule (i & 7)
  const bf16_t* kbase = p.K + (long long)bkv * p.k_sb + (long long)h * p.k_sh + (long long)key0 * p.k_ss;
  const bf16_t* vbase = p.V + (long long)bkv * p.v_sb + (long long)h * p.v_sh + (long long)key0 * p.v_ss;
  const int ppos = lane & 7;
  // piece i of a wave holds keys (wave + i NW) * 8 ..: the swizzle terms depend on (key >> 1) & 7 only, so further pieces (NW * 8 = a multiple
  // of 16 keys on) share the lane offset and differ in the SCALAR base -- one VGPR per operand whatever PPW is
  static_assert((NW * 8) % 16 == 0, "pieces of one wave must be a multiple of 16 keys apart");
  const int kit = wave * 8 + (lane >> 3);
  const unsigned k_loff = (unsigned)(kit * p.k_ss + (ppos ^ ((kit >> 1) & 7)) * 8) * 2u;            // K: 16-byte chunks XORed by (key >> 1) & 7
  const unsigned v_loff = (unsigned)(kit * p.v_ss + (ppos ^ (((kit >> 1) & 3) << 1)) * 8) * 2u;     // V: 32-byte chunks XORed by (key >> 1) & 3
  const int last_start = skv - SK;      // >= 0 (the launcher takes only Skv >= 256 and chunks of whole stages); a ragged last stage is slid back
  const unsigned lds0 = (unsigned)(size_t)smem;
  // The DMA requests stage 0, 1, 2, ... in order, so the source offset is WALKED: one 32-bit scalar per operand, advanced by a constant and
  // clamped to the slid-back last stage (requests past the end re-read it; never consumed) -- 2 scalar instructions per request instead of a
  // 64-bit multiply chain (~15).  32-bit byte offsets: a (batch, head) slice of K / V spans Skv * stride * 2 < 4 GB (checked by the launcher).
  const unsigned k_step = (unsigned)(SK * p.k_ss * 2), v_step = (unsigned)(SK * p.v_ss * 2);
  const unsigned k_last = (unsigned)((long long)last_start * p.k_ss * 2), v_last = (unsigned)((long long)last_start * p.v_ss * 2);
  const unsigned k_piece = (unsigned)(NW * 8 * p.k_ss * 2), v_piece = (unsigned)(NW * 8 * p.v_ss * 2);
  unsigned k_next = 0, v_next = 0;
  auto issue_kv = [&](int stage, int /*t: requests come in stage order*/) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      glds16_sbase((const char*)kbase + (k_next + i * k_piece), k_loff, lds0 + stage * STAGE_BYTES + (wave + i * NW) * 1024);
      glds16_sbase((const char*)vbase + (v_next + i * v_piece), v_loff, lds0 + V_BASE + stage * STAGE_BYTES + (wave + i * NW) * 1024);
    }
    k_next = k_next + k_step < k_last ? k_next + k_step : k_last;
    v_next = v_next + v_step < v_last ? v_next + v_step : v_last;
  };

  // ---- fragment read addresses
  {
    const int swz = (r16 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ln.ka[ks] = lds0 + r16 * 128 + (((4 * ks + ln.g) ^ swz) * 16);
    const int q4 = r16 >> 2, p4 = r16 & 3;
    const int key = 4 * ln.g + q4, vs = (key >> 1) & 3;
#pragma unroll
    for (int db = 0; db < 4; ++db) ln.va[db] = lds0 + key * 128 + ((db ^ vs) * 32) + p4 * 8;
  }

  f32x4 o[4][QB], negm[QB];
  f32x4 lacc[QB];          // OPTIMISTIC sweep: row sums accumulated by the matrix pipe (pv16<LSUM>): every register = the complete sum of the lane's query
  float m[QB], l[QB];      // safe sweep: per-lane partial row sums (folded over the 4 lane groups in the epilogue)
  const int nt = (skv + SK - 1) / SK;     // stages
  constexpr int D = NS - 1;
  auto wait_pair = [&]() {   // all but the (D - 1) youngest tile pairs of this wave have landed; then rendezvous
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW * (D - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // exact row maximum of the lane's queries over one tile's scores (first tile and the rare re-centre path only)
  auto tile_max = [&](const f32x4 (&s)[4][QB], float (&tm)[QB]) {
    // plain fmaxf (not the inline-asm v_max3 helper): these values come straight out of MFMAs, and hipcc pads the MFMA -> VALU read hazard
    // only for instructions it can see
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float a = fmaxf(fmaxf(s[0][qb][0], s[0][qb][1]), fmaxf(s[0][qb][2], s[0][qb][3]));
#pragma unroll
      for (int kb = 1; kb < 4; ++kb) a = fmaxf(a, fmaxf(fmaxf(s[kb][qb][0], s[kb][qb][1]), fmaxf(s[kb][qb][2], s[kb][qb][3])));
      a = fmaxf(a, __shfl_xor(a, 16));
      a = fmaxf(a, __shfl_xor(a, 32));
      tm[qb] = a;
    }
  };

  // One pass over the keys, in one of two forms:
  //   FAST (optimistic): NO running maximum.  P = exp2(S) straight from the score MFMAs (C operand = 0), rounded to bf16; O^T += V^T P^T and the row
  //     sums l^T += ONES P^T both on the matrix pipe (pv16<LSUM>); nothing per tile is checked.  Softmax is invariant under a common factor
  //     per row: subtracting a maximum only changes O and l by 2^-m, which cancels in O / l, and fp32 / bf16 keep the same RELATIVE precision at
  //     any magnitude -- so the maximum matters only where exp2 would leave the floating-point range: a row whose largest score (in log2 units,
  //     scale * log2 e folded into Q) is above ~+100 or below ~-100, i.e. a logit beyond +-69 nats.  The END of the sweep checks every row sum of
  //     the workgroup, 2^-100 <= l <= 2^100 (which leaves 2^27 of fp32 headroom for |O| <= l max|V| and catches inf / NaN / all-underflow); if
  //     ANY row fails, the workgroup runs the pass again in the
  //   SAFE form (round 2/3's loop): the first tile fixes the running max m (carried into the score MFMAs as their C operand), per-tile partial
  //     row sums on the vector pipe, a tile whose sum exceeds 2^24 re-centres -- exact tile maximum, m moves, O and l rescaled -- and re-runs
  //     its score MFMAs.
  // Per 16 x 64 scores the fast loop issues 16 v_exp + 8 v_cvt_pk + 18 MFMAs (vector issue ~160 of the 288 matrix-pipe cycles + 144 for the MFMAs'
  // own issue slots); the safe loop 8 v_dot2c + 3 more vector instructions and a ballot on top.  Partial results of the key-split tail carry
  // m = 0 from a fast sweep: attn_combine_kernel merges (m, l, O) triples whatever reference each one used.
  auto sweep = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    k_next = 0; v_next = 0;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      m[qb] = 0.f; l[qb] = 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < D; ++i) issue_kv(i, i);

    auto tile = [&](int t, auto off_c, auto hook) {
      constexpr int OFF = decltype(off_c)::value, SUB = 0;
      f32x4 s[4][QB];
      float ps[QB];
      const bool ragged = (t == nt - 1) && (skv & (SK - 1));
      const int gkey = skv - SK + SUB * KVB + 4 * ln.g;   // slid-back last stage: first key of this lane's group in key block 0 of this tile
      // Pass 1 is the whole story except on the first tile and (safe form) on a tile whose row sums explode: those re-centre -- exact tile
      // maximum, m moves, O and l are rescaled by the exact factor -- and, for an exploded tile, the score MFMAs are simply run again (its K
      // stage is still resident).
      bool recentre = !FAST && (t == 0 && SUB == 0);
      bf16x8 pb[2][QB];
      for (int pass = 0;; ++pass) {
        if constexpr (FAST) {
          f32x4 zero[QB];                       // S = K . Q^T + 0: the C operand is the inline constant
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) zero[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
          qk16<OFF, QB>(ln, qf, zero, s, hook);
        } else if (pass == 0) qk16<OFF, QB>(ln, qf, negm, s, hook);
        else qk16<OFF, QB>(ln, qf, negm, s, NoHook16());
        if (ragged) {   // keys before t * 64 were consumed by the previous tile
          asm volatile("; ragged last tile" ::: "memory");   // keeps hipcc from if-converting this into 48 selects on EVERY tile
          const int lo = t * SK;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (gkey < lo - (16 * kb + r)) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[kb][qb][r] = -INFINITY;
              }
        }
        if (recentre) {
          float tm[QB];
          tile_max(s, tm);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const bool first = (t == 0 && SUB == 0);
            const float delta = first ? fmaxf(tm[qb], -1e30f) : fmaxf(tm[qb], 0.f);   // S' is relative to m already: a row moves by max(0, tile max)
            if (!first) {
              const float alpha = __builtin_amdgcn_exp2f(-delta);
              l[qb] *= alpha;
#pragma unroll
              for (int db = 0; db < 4; ++db) o[db][qb] *= f32x4{alpha, alpha, alpha, alpha};
            }
            m[qb] += delta;
            negm[qb] = f32x4{-m[qb], -m[qb], -m[qb], -m[qb]};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb][qb] -= f32x4{delta, delta, delta, delta};
          }
        }
        if constexpr (FAST) {
#if MRAG_ATTN16_SETPRIO
          __builtin_amdgcn_s_setprio(MRAG_ATTN16_PRIO_EXP);
#endif
          // P = exp2(S') rounded to bf16: nothing else on the vector pipe
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r]);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
              const u32x4 w = {pack_bf2(s[2 * st][qb][0], s[2 * st][qb][1]), pack_bf2(s[2 * st][qb][2], s[2 * st][qb][3]),
                               pack_bf2(s[2 * st + 1][qb][0], s[2 * st + 1][qb][1]), pack_bf2(s[2 * st + 1][qb][2], s[2 * st + 1][qb][3])};
              pb[st][qb] = __builtin_bit_cast(bf16x8, w);
            }
          }
          break;
        } else {
          // P = exp2(S') rounded to bf16 first; the row sum is then one v_dot2c_f32_bf16 (pair . (1, 1) + acc) per packed register -- 8 + 1 instead of
          // 16 vector instructions per 16 scores -- and l sums exactly the P values that multiply V
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[kb][qb][r] = __builtin_amdgcn_exp2f(s[kb][qb][r]);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
              unsigned w0 = pack_bf2(s[2 * st][qb][0], s[2 * st][qb][1]), w1 = pack_bf2(s[2 * st][qb][2], s[2 * st][qb][3]);
              unsigned w2 = pack_bf2(s[2 * st + 1][qb][0], s[2 * st + 1][qb][1]), w3 = pack_bf2(s[2 * st + 1][qb][2], s[2 * st + 1][qb][3]);
              asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));   // hipcc 7.2 otherwise feeds sub-register 0 of the vector to all four dot2c
              a0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, w0), __builtin_bit_cast(bf16v2, 0x3f803f80u), a0, false);
              a1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, w1), __builtin_bit_cast(bf16v2, 0x3f803f80u), a1, false);
              a0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, w2), __builtin_bit_cast(bf16v2, 0x3f803f80u), a0, false);
              a1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, w3), __builtin_bit_cast(bf16v2, 0x3f803f80u), a1, false);
              const u32x4 w = {w0, w1, w2, w3};
              pb[st][qb] = __builtin_bit_cast(bf16x8, w);
            }
            ps[qb] = a0 + a1;
          }
          bool blown = false;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) blown |= !(ps[qb] <= kBig);   // also catches inf / NaN
          if (__builtin_expect(!__any(blown), 1) || recentre) break;     // a re-centred tile has P <= 1: it cannot explode again
          recentre = true;
        }
      }
      if constexpr (FAST) {
#if MRAG_ATTN16_SETPRIO
        __builtin_amdgcn_s_setprio(MRAG_ATTN16_PRIO_PV);
#endif
        pv16<V_BASE + OFF, QB, true>(ln, pb, o, &lacc);
#if MRAG_ATTN16_SETPRIO
        __builtin_amdgcn_s_setprio(MRAG_ATTN16_PRIO_QK);
#endif
      } else {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) l[qb] += ps[qb];
        pv16<V_BASE + OFF, QB>(ln, pb, o);
      }
    };
    auto iter = [&](int t, auto stage_c) {
      constexpr int STG = decltype(stage_c)::value;
      wait_pair();   // barrier #t: stage t has landed for every wave, stage t - 1's buffer is free
      auto early_issue = [&]() { issue_kv((STG + D) % NS, t + D); };
      if (!wave_active) { early_issue(); return; }
      tile(t, std::integral_constant<int, STG * STAGE_BYTES>{}, early_issue);
    };

    int t = 0;
    for (; t + NS <= nt; t += NS) {
      iter(t, std::integral_constant<int, 0>{});
      iter(t + 1, std::integral_constant<int, 1>{});
      if constexpr (NS >= 3) iter(t + 2, std::integral_constant<int, 2>{});
      if constexpr (NS >= 4) iter(t + 3, std::integral_constant<int, 3>{});
    }
    if (t < nt) { iter(t, std::integral_constant<int, 0>{}); ++t; }
    if constexpr (NS >= 3) { if (t < nt) { iter(t, std::integral_constant<int, 1>{}); ++t; } }
    if constexpr (NS >= 4) { if (t < nt) { iter(t, std::integral_constant<int, 2>{}); ++t; } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the clamped tail DMAs before the LDS is released (or re-used by a second sweep)
  };

  bool fast_ok = false;
#if MRAG_ATTN16_OPTIMISTIC
  {
    sweep(std::true_type{});
    bool bad = false;
    if (wave_active) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) bad |= !(lacc[qb][0] <= 1.2676506e30f && lacc[qb][0] >= 7.888609e-31f);   // [2^-100, 2^100]; false for inf / NaN
    }
    // workgroup-uniform verdict: the K / V ring is fed by all four waves, so the pass is repeated by all of them or by none.  The barrier also
    // fences the LDS between the sweeps (every wave's tail DMAs were retired just above)
    fast_ok = !__syncthreads_or((int)bad);
  }
#endif
  if (!fast_ok) sweep(std::false_type{});

  if (!wave_active) return;
  // ---- epilogue: row sums across the 4 lane groups, normalise, fused residual, 8-byte stores (4 consecutive features of a row per lane)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float lt;
    if (fast_ok) {
      lt = lacc[qb][0];          // complete already (the ONES . P^T product sums over every key)
    } else {
      lt = l[qb];
      lt += __shfl_xor(lt, 16);
      lt += __shfl_xor(lt, 32);
    }
    const int qrow = q0 + qb * 16 + r16;
    if (qrow >= p.Sq) continue;
    if (KVSPLIT && split_unit) {   // partial result of this key chunk; attn_combine_kernel merges the chunks
      const long long prow_i = (long long)(blockIdx.x - p.n_main) * p.rem_rows + wave * (QB * 16) + qb * 16 + r16;
      float* po = p.part_o + prow_i * 64 + 4 * ln.g;
#pragma unroll
      for (int db = 0; db < 4; ++db) *(f32x4*)(po + 16 * db) = o[db][qb];
      if (ln.g == 0) p.part_ml[prow_i] = make_float2(m[qb], lt);
      continue;
    }
    const float inv = p.out_scale / lt;
    const long long obase = (long long)b * p.o_sb + (long long)qrow * p.o_ss + h * 64 + 4 * ln.g;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      float v[4] = {o[db][qb][0] * inv, o[db][qb][1] * inv, o[db][qb][2] * inv, o[db][qb][3] * inv};
      const long long off = obase + 16 * db;
      if (p.resid) {
        const u32x2 rr = *(const u32x2*)(p.resid + off);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 out;
      out[0] = pack_bf2(v[0], v[1]);
      out[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.O + off) = out;          // (non-temporal Q loads / O stores measured: fabric reads 1.13 -> 0.98 GiB but writes 0.27 -> 0.60 GiB per launch and +0.5 % time: not kept, profiles/r4_attn_step_ab8.txt)
    }
  }
}

}  // namespace

template <int QB, int NW, int NS>
static int launch16_plain(hipStream_t s, AttnP p) {
  constexpr int ROWS = NW * QB * 16;
  const size_t lds = 2 * NS * TILE_BYTES;
  p.n_qtiles = (p.Sq + ROWS - 1) / ROWS;
  const void* kf = (const void*)attn16_kernel<QB, NW, NS, false>;
  const hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH((attn16_kernel<QB, NW, NS, false>), dim3(p.n_qtiles * p.B * p.H), dim3(NW * 64), lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN16);
  return MRAG_OK;
}

template <int QB, int NW, int NS>
static int launch16_split(hipStream_t s, AttnP p, const SplitPlan* pl, void* workspace) {
  const int nbh = p.B * p.H;
  const size_t lds = 2 * NS * TILE_BYTES;
  p.n_qtiles = pl->n_full;
  p.n_main = pl->n_full * nbh;
  p.kv_splits = pl->splits; p.chunk_keys = pl->chunk_keys; p.rem_rows = pl->rem_rows; p.tile_rows = NW * QB * 16;
  p.part_o = (float*)workspace;
  p.part_ml = (float2*)((char*)workspace + (size_t)nbh * pl->splits * pl->rem_rows * 64 * sizeof(float));
  const void* kf = (const void*)attn16_kernel<QB, NW, NS, true>;
  const hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH((attn16_kernel<QB, NW, NS, true>), dim3(p.n_main + nbh * pl->splits), dim3(NW * 64), lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN16_KSPLIT);
  return mrag_launch_attn_combine(s, p);
}

// Workgroup shapes (interleaved A/B on MI355X; round 2: profiles/r2_attn_ab_variants.txt, round 4 with the optimistic sweep: profiles/r4_attn_*_ab.txt):
//   QB = 3 (shipped): 48 query rows per wave, 4-wave workgroups of 192 rows, THREE per CU (168 VGPRs, 3 waves per SIMD, 48 KB of LDS each);
//   QB = 4 (-DMRAG_ATTN16_QB=4): 64 rows per wave, 256-row workgroups, TWO per CU (214 VGPRs): 3/4 of the K / V fragment bytes and barriers per FLOP.
//           5-7 % behind with the checked loop of round 3; with the optimistic sweep 2.6 % ahead in a cold microbenchmark and 1.5 % behind inside
//           the denoise step (attn_common.h) -- measured in the step, QB = 3 stays.
// Retired by measurement and pruned: 32 rows per wave in 8-wave workgroups, fragment prefetching at 150 VGPRs (-4 %), 128-key LDS stages (-1 %).
int mrag_launch_attn16(hipStream_t s, AttnP p, const SplitPlan* pl, void* workspace, int qb) {
  if (p.mask || p.Sq <= 128 || p.Skv < 4 * KVB) return MRAG_ENOTSUP;
  if ((long long)p.Skv * p.k_ss * 2 >= 0xffffffffLL || (long long)p.Skv * p.v_ss * 2 >= 0xffffffffLL) return MRAG_ENOTSUP;   // walked 32-bit DMA offsets
  if (pl) {
    if (pl->chunk_keys % KVB != 0 || pl->rem_rows >= mrag_attn16_rows(qb)) return MRAG_ENOTSUP;
    if (qb == 2) return launch16_split<2, 4, 2>(s, p, pl, workspace);        // A/B shape: 32 rows per wave, FOUR workgroups per CU, two-stage ring (32 KB of LDS each)
    return qb == 4 ? launch16_split<4, 4, 3>(s, p, pl, workspace) : launch16_split<3, 4, 3>(s, p, pl, workspace);
  }
  if (qb == 2) return launch16_plain<2, 4, 2>(s, p);
  return qb == 4 ? launch16_plain<4, 4, 3>(s, p) : launch16_plain<3, 4, 3>(s, p);
}
