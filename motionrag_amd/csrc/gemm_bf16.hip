// gemm_bf16.hip -- C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias), bf16 in, fp32 accumulate.
//
// Stands behind every nn.Linear on the MotionRAG hot path (see include/mrag_hip.h).
// CDNA4 design (not a port of anything):
//   * v_mfma_f32_16x16x32_bf16, 64-lane wavefronts, wave tile (TM*16) x (TN*16);
//   * both operands are K-contiguous (activations [M,K], nn.Linear weight [N,K]), so A and W
//     tiles use the same LDS image: [rows][64 k] bf16 = 128-byte rows, filled by 16-byte
//     global_load_lds (LDS-DMA, no VGPR round trip), XOR-swizzled on the SOURCE address
//     (chunk ^= row & 7) and un-swizzled on the ds_read_b128 -> conflict-free fragment reads;
//   * two LDS stages; the DMA for K-tile t+1 is issued before the MFMAs of tile t and is
//     retired by the one vmcnt(0)+barrier per K-tile;
//   * operands swapped in the MFMA (W fragment as A-operand) so each lane owns 4 consecutive
//     output columns of one row -> 8-byte bf16 stores and a lane-local fused epilogue;
//   * 1-D grid with a bijective XCD remap so tiles that share an A row-panel sit on one L2.
#include <type_traits>
#include "common.h"
#include "../../include/mrag_hip.h"

#ifdef MRAG_GEMM_STAMPS
// diagnostic build only (tools/build_diag.sh): per-phase s_memtime sums of the 256x256 main loop; never compiled into the product
__device__ unsigned long long* g_gemm_stamp_buf = nullptr;
extern "C" int mrag_debug_set_gemm_stamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp_buf), &p, sizeof(p)); }
#define MRAG_GSTAMP(T)                                                                 \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#else
#define MRAG_GSTAMP(T) do {} while (0)
#endif
#ifndef MRAG_W4_RESID_DEPTH
#define MRAG_W4_RESID_DEPTH 2   // residual row groups in flight in the four-wave kernel's epilogue (developer knob: tools/build_variant.sh)
#endif
#ifndef MRAG_GEMM_TRACE
#define MRAG_GEMM_TRACE 0
#endif
#ifdef MRAG_GEMM_SAMEK   // diagnostic only: every K-tile re-reads tile 0 (always L2-resident) to separate memory latency from sync cost
#define MRAG_DIAG_KSTEP 0
#else
#define MRAG_DIAG_KSTEP BK
#endif

#ifndef MRAG_QK_RING
#define MRAG_QK_RING 4   // row groups of RoPE table rows in flight in the QKNORM_ROPE epilogue (16 registers each; 6+ spill and lose)
#endif
#ifndef MRAG_QK_PRE
#define MRAG_QK_PRE 0    // of which requested before the accumulators are staged (measured equal to 0 on MI355X: 1.925 vs 1.927 ms)
#endif

namespace {

struct GemmP {
  const bf16_t* A; const bf16_t* W; const bf16_t* bias; bf16_t* C; const bf16_t* resid;
  const bf16_t* gate0; const bf16_t* gate1;
  long long M, N, K, lda, ldw, ldc, ldr, rows_per_batch, split, gate_stride;
  int tiles_m, tiles_n, group_m, staged, tuning;
  // MRAG_EPI_QKNORM_ROPE
  const bf16_t* qg; const bf16_t* qb; const bf16_t* kg; const bf16_t* kb; const float* rcos; const float* rsin;
  long long qk_D; int rope_text_len, qk_first; float qk_eps, q_premul;
  // implicit-GEMM convolution (CONV != 0): A is the channels-last activation, rows are gathered per K-tile
  int cv_H, cv_W, cv_Hi, cv_Wi, cv_Ho, cv_Wo, cv_stride, cv_up, cv_ctiles, cv_T, cv_pad;   // cv_pad: zero rows / columns in FRONT of the image (1, or 0 for the bottom/right-only padding)
  long long cv_C, cv_HW;
  // stream-K tail (SK instantiation): logical tiles [0, sk_main) run one per workgroup; the sk_rem tiles behind them are cut into sk_units equal
  // runs of K-tiles, one per workgroup; partial accumulators meet in sk_part, the last arriver of a tile (sk_ticket) sums them in K order
  float acc_scale;    // MRAG_EPI_RESID: C = resid + acc_scale * (acc + bias) (1 unless the caller blends: AlphaBlender folded into a residual branch)
  float* sk_part; unsigned* sk_ticket;
  int sk_main, sk_rem, sk_units, sk_maxparts;
  const bf16_t* lna_g; const bf16_t* lna_b; float lna_eps; int lna;   // gemm_skinny_kernel<.., LNA>: A := LayerNorm_K(A) * lna_g + lna_b in front of the product (either may be null)
  int tile_limit;     // gemm_w4_kernel: tiles [0, tile_limit) of the logical order (all of them, or the whole rounds in front of a tail launch: launch_w4)
  int wb_tiles_m;     // gemm_w4_kernel<EPI, true> (per-sample weights): 256-row tiles per sample -- the row-tile grid restarts at every sample; 0 otherwise
  long long w_bstride;   // elements between the samples' weight matrices
  int cv_lds;         // CONV != 0: byte offset of the parked per-lane tap state in LDS (behind the operand stages / staged-epilogue region)
  int cv_tf;          // CONV == 1 with three temporal taps (causal 3x3x3): output frames per sample (input holds cv_tf + 2 frames per sample); 0 = 2-D
  long long cv_fs;    // elements between consecutive input frames
};

constexpr int SK_FLAG_OFF = 8 * 128 * 144;   // one LDS word behind the staged epilogue's region (stream-K: "this workgroup finishes the tile")

// zero source for the taps that fall outside the image / clip (never written)
__device__ __attribute__((aligned(128))) bf16_t g_zero_row[64];

// sum over the 8 lanes that hold one row (lanes 8g..8g+7) on the vector pipe: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
// (lane i <-> 7 - i of its half row).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip each, six dependent ones per row group.
__device__ __forceinline__ float sum8_dpp(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
  return x;
}

// one row's 8 features of a head in the row layout (the 8 lanes 8g .. 8g + 7 hold the head's 64 features): per-head LayerNorm across those lanes, RoPE on the
// lane's four (even, odd) pairs, Q pre-multiplied -- the arithmetic of qknorm_rope_kernel (norm.hip).  Shared by the 8-wave and the four-wave kernels (same bits).
__device__ __forceinline__ u32x4 qk_row_math(u32x4 val, const bool has_gamma, const bool has_beta, const float (&gam)[8], const float (&bet)[8], const float eps,
                                             const bool has_rope, const bool vid, const f32x4 (&t4)[4], const bool premul_on, const float premul) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(val[e] << 16); v[2 * e + 1] = __uint_as_float(val[e] & 0xffff0000u); }
  if (has_gamma) {
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[e];
    sum = sum8_dpp(sum);
    const float mean = sum * (1.0f / 64.0f);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] -= mean; sq += v[e] * v[e]; }
    sq = sum8_dpp(sq);
    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gam[e];
    if (has_beta) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bet[e];
    }
  }
  if (has_rope) {
    const float cc[8] = {t4[0][0], t4[0][1], t4[0][2], t4[0][3], t4[1][0], t4[1][1], t4[1][2], t4[1][3]};
    const float ss[8] = {t4[2][0], t4[2][1], t4[2][2], t4[2][3], t4[3][0], t4[3][1], t4[3][2], t4[3][3]};
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
      const float a = v[2 * i2], b2 = v[2 * i2 + 1];
      const float oa = a * cc[2 * i2] - b2 * ss[2 * i2];
      const float ob = b2 * cc[2 * i2 + 1] + a * ss[2 * i2 + 1];
      v[2 * i2] = vid ? oa : a;
      v[2 * i2 + 1] = vid ? ob : b2;
    }
  }
  if (premul_on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= premul;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) val[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
  return val;
}

// internal epilogue id: MRAG_EPI_GEGLU with the tanh gate (mrag_gemm_args.geglu_act = 1, T5's gated-gelu): its own instantiation, so the erf kernels of the
// UNets (epilogue-bound at K = 320) carry neither a branch nor the second activation's registers
constexpr int EPI_GEGLU_TANH = 8;
template <int EPI>
constexpr bool is_geglu = (EPI == MRAG_EPI_GEGLU || EPI == EPI_GEGLU_TANH);

template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
  if constexpr (EPI == MRAG_EPI_GELU_TANH) return gelu_tanh_f(v);
  else if constexpr (EPI == MRAG_EPI_GELU_ERF) return gelu_erf_f(v);
  else if constexpr (EPI == MRAG_EPI_SILU) return silu_f(v);
  else return v;
}

// ---- direct epilogue (accumulator layout): lane owns row m = .. + (lane & 15), columns n0 + (lane >> 4) * 4 + {0..3} of every 16x16 tile; 8-byte stores.
// Same rounding points as the LDS-staged epilogue (so a GEMM gives the same bits whichever tile configuration its size selects).
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue_direct(const GemmP& p, f32x4 (&acc)[TM][TN], const long long bm0, const long long bn0, const int wrow0, const int wcol0, const int lane) {
  const int frag_row = lane & 15, frag_q = lane >> 4;
  long long wg_b = 0, wg_pos = 0;
  if constexpr (EPI == MRAG_EPI_GATE_RESID) {
    wg_b = bm0 / p.rows_per_batch;
    wg_pos = bm0 - wg_b * p.rows_per_batch;
  }
  auto row_bp = [&](long long m, long long& b, long long& pos) {
    b = wg_b; pos = wg_pos + (m - bm0);
    while (pos >= p.rows_per_batch) { pos -= p.rows_per_batch; ++b; }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm0 + wrow0 + i * 16 + frag_row;
    if (m >= p.M) continue;
    const bf16_t* gate = nullptr;
    if constexpr (EPI == MRAG_EPI_GATE_RESID) {
      long long b, pos;
      row_bp(m, b, pos);
      gate = (pos < p.split ? p.gate0 : p.gate1) + b * p.gate_stride;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long long n = bn0 + wcol0 + j * 16 + frag_q * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias) {
        const u32x2 bb = *(const u32x2*)(p.bias + n);
        v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
        v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = epi_act<EPI>(v[e]);
      if constexpr (EPI == MRAG_EPI_GATE_RESID) {
        const u32x2 gg = *(const u32x2*)(gate + n);
        v[0] *= __uint_as_float(gg[0] << 16); v[1] *= __uint_as_float(gg[0] & 0xffff0000u);
        v[2] *= __uint_as_float(gg[1] << 16); v[3] *= __uint_as_float(gg[1] & 0xffff0000u);
      }
      if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
      }
      if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID) {
        // the same rounding points as the LDS-staged epilogue above and as the reference's bf16 tensors (`x + gate * linear(.)`: the gated
        // projection is a bf16 tensor before the residual add) -- so a GEMM gives the same bits whichever tile configuration its size selects
        // (a sequence-sharded rank runs smaller problems than the unsharded model)
        const u32x2 rr = *(const u32x2*)(p.resid + m * p.ldr + n);
        v[0] = bf_round(v[0]) + __uint_as_float(rr[0] << 16); v[1] = bf_round(v[1]) + __uint_as_float(rr[0] & 0xffff0000u);
        v[2] = bf_round(v[2]) + __uint_as_float(rr[1] << 16); v[3] = bf_round(v[3]) + __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 out;
      out[0] = pack_bf2(v[0], v[1]);
      out[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.C + m * p.ldc + n) = out;
    }
  }
}

// One output tile (SK: one run of K-tiles [kt0, kt0 + nk) of it).  `wg` = the tile's index in the logical order.
template <int WM, int WN, int TM, int TN, int EPI, int CONV, bool SK>
__device__ __forceinline__ void gemm_tile(const GemmP& p, const int wg, const int kt0, const int nk, const int sk_tile, const int sk_unit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 64;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int PIECES = (BM + BN) / 8;   // 1 KiB LDS-DMA pieces per stage (8 rows x 128 B)
  constexpr int PPW = PIECES / NW;        // pieces per wave
  static_assert(PIECES % NW == 0, "piece split");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MRAG_GEMM_STAMPS
  unsigned long long g_entry, g_loop0 = 0, g_loop1 = 0, g_exit;
  MRAG_GSTAMP(g_entry);
#endif
  const int wm = wave / WN, wn = wave % WN;

  // logical tile order: groups of GROUP_M m-tiles walked n-major, so the ~32 workgroups resident on one XCD (a contiguous
  // run of the logical order after the XCD remap) form a GROUP_M x 8 block that shares GROUP_M A-panels and 8 W-panels
  // per K-step through that XCD's L2 (instead of 1 A-panel and 32 W-panels)
  const int gw = p.group_m * p.tiles_n;
  const int first_m = (wg / gw) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  const int tile_m = first_m + (wg % gw) % gsz, tile_n = (wg % gw) / gsz;
  const long long bm0 = (long long)tile_m * BM, bn0 = (long long)tile_n * BN;

  // ---- per-lane DMA sources (k = 0), one per piece this wave stages.
  // Plain GEMM, and the convolutions on every tile but 256x256: a 64-bit row pointer per piece (+ per A piece of a convolution the tap-independent position and
  // the walked tap cursor: cv_y, cv_x, cv_src, cv_step).  The 8-wave 256x256 tile runs at exactly 256 VGPRs (128 accumulators, two sets of 48 fragment
  // registers): there those 36 registers were 44-51 SPILLED VGPRs -- two scratch reloads per K-tile, each behind an `s_waitcnt vmcnt(0)` that also drains the
  // LDS-DMA ring (round-5 review; tools/check_scratch.py).  SLIM form (256x256 convolutions only; the other tiles have the registers and measured 1 % slower on
  // it: profiles/r6_conv_scratch_ab.txt): per A piece the current tap's source as ONE 32-bit offset in 16-byte units relative to a workgroup-uniform base
  // (`cv_base`, an SGPR pair), per W piece a 32-bit byte offset for the scalar-base form of the DMA: 8 registers.  What a tap change needs to recompute the
  // offsets -- the tap-independent position (y << 16 | x, or the frame index) and the sample's offset, two words per piece -- is parked in LDS (lane-linear
  // words behind the operand stages, GemmP::cv_lds): read back once per tap by ds_read, which counts on lgkmcnt and leaves the DMA ring alone.
  constexpr int APW = BM / 8 / NW;          // a wave's first APW pieces are A rows (piece = wave + i * NW < BM / 8)
  static_assert((BM / 8) % NW == 0, "A pieces split evenly over the waves");
  constexpr bool SLIM = CONV != 0 && TM == 8 && TN == 4 && WM == 2 && WN == 4;
  static_assert(!SLIM || APW <= 4, "the parked conv state is read back by four hand-written statements");
  constexpr int CV_NONE = (int)0x80000000;  // cv_cur: the tap falls outside the image / clip -> the zero row
  const bf16_t* gsrc[SLIM ? 1 : PPW];
  int cv_y[(CONV != 0 && !SLIM) ? APW : 1], cv_x[(CONV != 0 && !SLIM) ? APW : 1];      // legacy conv form: per A piece
  int cv_cur[SLIM ? APW : 1];
  const unsigned cv_park = (unsigned)(size_t)smem + (unsigned)p.cv_lds + (unsigned)tid * 4u;   // word k of this lane at + k * NW * 256: k = 2 i (position), 2 i + 1 (sample offset)
  auto cv_put = [&](int k, int v) { *(int*)(smem + p.cv_lds + (k * NW * 64 + tid) * 4) = v; };
  unsigned cv_woff[SLIM ? PPW - APW : 1];
  const bf16_t* cv_base = p.A;              // workgroup-uniform
  const int cv_c8 = (int)(p.cv_C >> 3);     // 16-byte units per pixel
  if constexpr (SLIM && CONV == 1) {        // the sample (input frame stack position) of the tile's first row
    const long long n0 = bm0 / ((long long)p.cv_Wo * p.cv_Ho);
    const long long n0_in = p.cv_tf ? n0 + 2 * (n0 / p.cv_tf) : n0;
    cv_base = p.A + n0_in * p.cv_H * p.cv_W * p.cv_C;
  } else if constexpr (SLIM && CONV == 2) {
    cv_base = p.A + bm0 * p.cv_C;
  }
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave + i * NW;  // pieces [0, BM/8) are A rows, the rest W rows
    const int r = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (lane >> 3);  // source-side swizzle: row&7 == lane>>3
    if (piece < BM / 8) {
      long long row = bm0 + r;
      row = row < p.M ? row : p.M - 1;  // clamp: tail rows re-read a valid row, stores are masked
      if constexpr (CONV == 1) {         // row = (n, yo, xo) of the output image: keep (yo*stride - pad, xo*stride - pad) and the sample's position
        const int xo = (int)(row % p.cv_Wo);
        const long long r2 = row / p.cv_Wo;
        const int yo = (int)(r2 % p.cv_Ho);
        const long long n = r2 / p.cv_Ho;
        const long long n_in = p.cv_tf ? n + 2 * (n / p.cv_tf) : n;   // 3-D: sample s's output frame t reads input frames s (T + 2) + t + {0, 1, 2}
        if constexpr (SLIM) {
          const long long n0 = bm0 / ((long long)p.cv_Wo * p.cv_Ho);
          const long long n0_in = p.cv_tf ? n0 + 2 * (n0 / p.cv_tf) : n0;
          cv_put(2 * i, (int)(((unsigned)(yo * p.cv_stride - p.cv_pad) << 16) | ((unsigned)(xo * p.cv_stride - p.cv_pad) & 0xffffu)));
          cv_put(2 * i + 1, (int)(n_in - n0_in) * (p.cv_H * p.cv_W * cv_c8) + chunk);
        } else {
          gsrc[i] = p.A + n_in * p.cv_H * p.cv_W * p.cv_C + chunk * 8;
          cv_y[i < APW ? i : 0] = yo * p.cv_stride - p.cv_pad;
          cv_x[i < APW ? i : 0] = xo * p.cv_stride - p.cv_pad;
        }
      } else if constexpr (CONV == 2) {  // row = (b, t, hw): keep the row's position and t
        if constexpr (SLIM) {
          cv_put(2 * i, (int)((row / p.cv_HW) % p.cv_T));
          cv_put(2 * i + 1, (int)(row - bm0) * cv_c8 + chunk);
        } else {
          gsrc[i] = p.A + row * p.cv_C + chunk * 8;
          cv_y[i < APW ? i : 0] = (int)((row / p.cv_HW) % p.cv_T);
        }
      } else {
        gsrc[i] = p.A + row * p.lda + chunk * 8 + (SK ? (long long)kt0 * BK : 0);
      }
    } else {
      long long row = bn0 + (r - BM);
      row = row < p.N ? row : p.N - 1;
      if constexpr (SLIM) cv_woff[i >= APW ? i - APW : 0] = (unsigned)((row * p.ldw + chunk * 8) * 2);   // (< 4 GiB: checked by mrag_conv_bf16)
      else gsrc[i] = p.W + row * p.ldw + chunk * 8 + (SK ? (long long)kt0 * BK : 0);
    }
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a stage: rows of A start at 0, rows of W at BM*128
  const int frag_row = lane & 15, frag_q = lane >> 4, swz = lane & 7;
  const int a_off = (wm * TM * 16 + frag_row) * 128;
  const int w_off = BM * 128 + (wn * TN * 16 + frag_row) * 128;

  // DMA source of piece i for K-tile kt.  Plain GEMM: the row pointer advanced by kt * 64.  Convolutions: K-tile kt is channel
  // block (kt % ctiles) of tap (kt / ctiles); the lane's row is the tap-shifted pixel (or frame), or the zero row outside.
  // The K-tiles are requested in order (0, 1, 2, ...), so the (tap, channel block) pair is WALKED: the tap geometry (bounds test, pixel
  // offset) is redone only when the tap changes -- every Cin / 64 K-tiles -- and leaves one 32-bit offset per piece (cv_cur); a K-tile's
  // request adds the channel block and the workgroup's base to it (a handful of vector instructions per piece, no persistent pointer).
  const bf16_t* cv_src[(CONV != 0 && !SLIM) ? APW : 1];   // legacy form, per A piece: this lane's source at the current (tap, channel block)
  int cv_step[(CONV != 0 && !SLIM) ? APW : 1];            // 64 elements per channel block inside the image, 0 on the zero row
  int cv_kt = -1, cv_tap = 0, cv_cblk = -1;
  auto cv_prepare = [&](int kt) {
    if constexpr (CONV != 0) {
      if (kt == cv_kt) return;
      cv_kt = kt;
      bool new_tap = kt == 0;
      if (++cv_cblk == p.cv_ctiles) { cv_cblk = 0; ++cv_tap; new_tap = true; }
      if (new_tap) {
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          bool ok;
          if constexpr (SLIM) {
            int off, yx, nb;
            // (hand-written reads: a compiler-made LDS load would be ordered behind the LDS-DMA pieces in flight -- `s_waitcnt vmcnt(0)`, the drain this form removes)
            if (i == 0) asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(yx), "=&v"(nb) : "v"(cv_park), "n"(NW * 256) : "memory");
            else if (i == 1) asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(yx), "=&v"(nb) : "v"(cv_park), "n"(2 * NW * 256), "n"(3 * NW * 256) : "memory");
            else if (i == 2) asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(yx), "=&v"(nb) : "v"(cv_park), "n"(4 * NW * 256), "n"(5 * NW * 256) : "memory");
            else asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(yx), "=&v"(nb) : "v"(cv_park), "n"(6 * NW * 256), "n"(7 * NW * 256) : "memory");
            if constexpr (CONV == 1) {
              const int kt3 = p.cv_tf ? cv_tap / 9 : 0, tap9 = cv_tap - 9 * kt3;   // taps in (kt, ky, kx) order; kt3 = 0 for the 2-D convolution
              const int ky = tap9 / 3, kx = tap9 - 3 * ky;
              const int yi = (yx >> 16) + ky, xi = (int)(short)(yx & 0xffff) + kx;
              ok = (unsigned)yi < (unsigned)p.cv_Hi && (unsigned)xi < (unsigned)p.cv_Wi;
              off = nb + ((yi >> p.cv_up) * p.cv_W + (xi >> p.cv_up)) * cv_c8 + kt3 * (int)(p.cv_fs >> 3);
            } else {
              const int t = yx + cv_tap - 1;
              ok = (unsigned)t < (unsigned)p.cv_T;
              off = nb + (cv_tap - 1) * (int)p.cv_HW * cv_c8;
            }
            cv_cur[i] = ok ? off : CV_NONE;
          } else {
            long long off;
            if constexpr (CONV == 1) {
              const int kt3 = p.cv_tf ? cv_tap / 9 : 0, tap9 = cv_tap - 9 * kt3;
              const int ky = tap9 / 3, kx = tap9 - 3 * ky;
              const int yi = cv_y[i] + ky, xi = cv_x[i] + kx;
              ok = (unsigned)yi < (unsigned)p.cv_Hi && (unsigned)xi < (unsigned)p.cv_Wi;
              off = ((long long)(yi >> p.cv_up) * p.cv_W + (xi >> p.cv_up)) * p.cv_C + kt3 * p.cv_fs;
            } else {
              const int t = cv_y[i] + cv_tap - 1;
              ok = (unsigned)t < (unsigned)p.cv_T;
              off = (long long)(cv_tap - 1) * p.cv_HW * p.cv_C;
            }
            cv_src[i] = ok ? gsrc[i] + off : g_zero_row + (lane & 7) * 8;
            cv_step[i] = ok ? 64 : 0;
          }
        }
      } else if constexpr (!SLIM) {
#pragma unroll
        for (int i = 0; i < APW; ++i) cv_src[i] += cv_step[i];
      }
    }
  };
  // request piece i of K-tile kt into `dst` (the piece's 1-KiB slot of a stage).  Convolutions: cv_prepare(kt) ran for this K-tile.
  auto dma_piece = [&](int i, int kt, char* dst) {
    if constexpr (CONV == 0) {
      glds16(gsrc[i] + (long long)kt * MRAG_DIAG_KSTEP, dst);
    } else if constexpr (!SLIM) {
      glds16(i >= APW ? gsrc[i] + (long long)kt * BK : cv_src[i < APW ? i : 0], dst);      // weight rows [Cout, taps * Cin] are plain
    } else {
      if (i >= APW) {                                          // weight rows: scalar base + the lane's byte offset
        glds16_sbase(p.W + (long long)kt * BK, cv_woff[i - APW], (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dst));   // (wave-uniform by construction; the asm wants it in an SGPR)
      } else {
        const int c = cv_cur[i];
        const bf16_t* in_img = cv_base + ((long long)(c + cv_cblk * 8) << 3);
        glds16(c == CV_NONE ? g_zero_row + (lane & 7) * 8 : in_img, dst);
      }
    }
  };
  auto issue = [&](int stage, int kt) {
    char* base = smem + stage * STAGE_BYTES;
    cv_prepare(kt);
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_piece(i, kt, base + (wave + i * NW) * 1024);  // wave-uniform base (+ lane*16 by HW)
  };

  issue(0, 0);
  if constexpr (TM == 8 && TN == 4 && WM == 2 && WN == 4) {
    // ---- 256x256 tile: all 12 fragments of a 32-deep k-step are requested by ONE asm statement and released to the MFMAs by
    // COUNTED s_waitcnt lgkmcnt(N) (LDS reads return in order), the second k-step's 12 reads are issued while the first
    // k-step's MFMAs run -> the LDS latency is paid once per K-tile instead of eight times (hipcc's own schedule: read
    // pair -> lgkmcnt(0) -> 8 MFMAs).  At most 15 LDS reads are outstanding (lgkmcnt is a 4-bit counter).
    //
    // Software pipeline across the per-tile barrier: the fragments of k-step (t, 0) are already in registers when tile t's MFMAs
    // start, the reads of (t, 1) fly under the 32 MFMAs of (t, 0), and the ONE barrier per K-tile sits between the two k-steps:
    // behind it every wave has finished reading stage t (so the DMA of tile t+2 may overwrite it) and tile t+1 has landed (so
    // the reads of (t+1, 0) are issued right there, under the MFMAs of (t, 1)).  No fragment latency is exposed at the tile
    // boundary (measured before: ~350 cycles of first-fragment wait + ~500 of barrier per 2048-cycle MFMA body).
#ifdef MRAG_GEMM_STAMPS
    unsigned long long g_acc[5] = {0, 0, 0, 0, 0}, g0, g1, g2, g3, g4, g5;
#endif
#define MRAG_READ12(W, A, AW, AA)                                                                                   \
      asm volatile(                                                                                                  \
          "ds_read_b128 %0, %12 offset:32768\n\tds_read_b128 %1, %12 offset:34816\n\t"                               \
          "ds_read_b128 %2, %12 offset:36864\n\tds_read_b128 %3, %12 offset:38912\n\t"                               \
          "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:2048\n\t"                                             \
          "ds_read_b128 %6, %13 offset:4096\n\tds_read_b128 %7, %13 offset:6144\n\t"                                 \
          "ds_read_b128 %8, %13 offset:8192\n\tds_read_b128 %9, %13 offset:10240\n\t"                                \
          "ds_read_b128 %10, %13 offset:12288\n\tds_read_b128 %11, %13 offset:14336"                                  \
          : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3]), "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]),    \
            "=&v"(A[4]), "=&v"(A[5]), "=&v"(A[6]), "=&v"(A[7])                                                       \
          : "v"(AW), "v"(AA)                                                                                         \
          : "memory")
#define MRAG_WAIT12(N, W, A)                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                        \
                   : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]),   \
                     "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7])                                                  \
                   :: "memory")
#define MRAG_ROW(I, W, X)                                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[I][j] =                                                                 \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W[j]), __builtin_bit_cast(bf16x8, X), acc[I][j], 0, 0, 0)
    const unsigned smem_u = (unsigned)(size_t)smem;
    const unsigned c0 = ((frag_q + 0) ^ swz) * 16, c1 = ((frag_q + 4) ^ swz) * 16;
    const unsigned offA = a_off, offW = w_off - BM * 128;   // the W reads carry offset:32768 (= BM * 128) in the instruction
    u32x4 w0[4], a0[8], w1[4], a1[8];
    if (nk > 1) {
      issue(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PPW) : "memory");   // tile 0 landed everywhere, tile 1 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    MRAG_READ12(w0, a0, smem_u + offW + c0, smem_u + offA + c0);
#ifdef MRAG_GEMM_STAMPS
    MRAG_GSTAMP(g_loop0);
#endif
    for (int kt = 0; kt < nk; ++kt) {
      MRAG_GSTAMP(g0);
      const unsigned st = smem_u + (kt & 1) * STAGE_BYTES;
      MRAG_WAIT12(0, w0, a0);            // the (t, 0) fragments (requested one k-step ago) are here
      MRAG_GSTAMP(g1);
      MRAG_ROW(0, w0, a0[0]);
      MRAG_READ12(w1, a1, st + offW + c1, st + offA + c1);   // behind the first MFMAs: the 12 KB read burst of 8 waves takes up to ~380 cycles to issue
      MRAG_ROW(1, w0, a0[1]); MRAG_ROW(2, w0, a0[2]); MRAG_ROW(3, w0, a0[3]);
      MRAG_ROW(4, w0, a0[4]); MRAG_ROW(5, w0, a0[5]); MRAG_ROW(6, w0, a0[6]); MRAG_ROW(7, w0, a0[7]);
      __builtin_amdgcn_sched_barrier(0);
      MRAG_WAIT12(0, w1, a1);            // this wave is done reading stage t
      MRAG_GSTAMP(g2);
      const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
      if (more) {
#ifdef MRAG_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MRAG_GSTAMP(g5);
        g_acc[4] += g5 - g2;
        asm volatile("s_barrier" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // tile t+1 landed for every wave; stage t is free
#endif
      }
      MRAG_GSTAMP(g3);
      // the 8 LDS-DMA pieces of tile t+2 go into stage t, ONE PER ROW GROUP between the MFMAs (a burst of 8 costs ~100 cycles
      // each at issue, measured with s_memtime stamps)
      char* nbase = smem + (kt & 1) * STAGE_BYTES;
#define MRAG_PIECE(I) if (more2) dma_piece(I, kt + 2, nbase + (wave + (I) * NW) * 1024)
      MRAG_ROW(0, w1, a1[0]);
      if (more) {
        const unsigned sn = smem_u + ((kt + 1) & 1) * STAGE_BYTES;
        MRAG_READ12(w0, a0, sn + offW + c0, sn + offA + c0);
      }
      if (more2) cv_prepare(kt + 2);
      MRAG_PIECE(0);
      MRAG_ROW(1, w1, a1[1]); MRAG_PIECE(1);
      MRAG_ROW(2, w1, a1[2]); MRAG_PIECE(2);
      MRAG_ROW(3, w1, a1[3]); MRAG_PIECE(3);
      MRAG_ROW(4, w1, a1[4]); MRAG_PIECE(4);
      MRAG_ROW(5, w1, a1[5]); MRAG_PIECE(5);
      MRAG_ROW(6, w1, a1[6]); MRAG_PIECE(6);
      MRAG_ROW(7, w1, a1[7]); MRAG_PIECE(7);
      __builtin_amdgcn_sched_barrier(0);
#undef MRAG_PIECE
#ifdef MRAG_GEMM_STAMPS
      MRAG_GSTAMP(g4);
      g_acc[0] += g1 - g0; g_acc[1] += g2 - g1; g_acc[2] += g3 - g2; g_acc[3] += g4 - g3;
      if (MRAG_GEMM_TRACE && g_gemm_stamp_buf && lane == 0 && blockIdx.x < 4 && kt >= 8 && kt < 12) {   // absolute timeline of 4 K-tiles, after the 64 Ki-word summary
        unsigned long long* tr = g_gemm_stamp_buf + 65536 + (((long long)blockIdx.x * 8 + wave) * 4 + (kt - 8)) * 8;
        tr[0] = g0; tr[1] = g1; tr[2] = g2; tr[3] = g5; tr[4] = g3; tr[5] = g4;
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        tr[6] = hwid;
      }
#endif
    }
#undef MRAG_READ12
#undef MRAG_WAIT12
#undef MRAG_ROW
#ifdef MRAG_GEMM_STAMPS
    MRAG_GSTAMP(g_loop1);
    if (g_gemm_stamp_buf && lane == 0 && blockIdx.x < 1024) {
      for (int k = 0; k < 4; ++k) g_gemm_stamp_buf[((long long)blockIdx.x * 8 + wave) * 8 + k] = g_acc[k];
      g_gemm_stamp_buf[((long long)blockIdx.x * 8 + wave) * 8 + 4] = nk;
      g_gemm_stamp_buf[((long long)blockIdx.x * 8 + wave) * 8 + 5] = g_acc[4];
    }
#endif
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // tile kt landed for every wave; everyone finished reading the other stage
      if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
      const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((frag_q + 4 * ks) ^ swz) * 16;
        bf16x8 wf[TN], af[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(st + w_off + j * 16 * 128 + coff);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(st + a_off + i * 16 * 128 + coff);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  }

  if constexpr (SK) {
    if (nk != (int)(p.K / BK)) {   // a partial run of this tile's K-tiles (workgroup-uniform)
      // Contributors of tail tile T are the units whose iteration range [b(u), b(u + 1)), b(u) = u I / U, meets [T nk_full, (T + 1) nk_full): consecutive
      // units, numbered in K order.  Every contributor parks its fp32 accumulators in its own slot and takes a ticket; the LAST arriver sums the
      // slots in K order -- a fixed order, so the result does not depend on who arrives last (bit-reproducible run to run) -- and runs the epilogue.
      const int nkf = (int)(p.K / BK), I = p.sk_rem * nkf, U = p.sk_units;
      auto owner = [&](int it) {   // the unit whose range holds iteration `it`
        int u = (int)(((long long)it * U) / I);
        while ((int)(((long long)(u + 1) * I) / U) <= it) ++u;
        while ((int)(((long long)u * I) / U) > it) --u;
        return u;
      };
      const int u_first = owner(sk_tile * nkf), u_last = owner(sk_tile * nkf + nkf - 1);
      const int part = sk_unit - u_first, nparts = u_last - u_first + 1;
      float* slot0 = p.sk_part + (size_t)sk_tile * p.sk_maxparts * (BM * BN);
      float* mine = slot0 + (size_t)part * (BM * BN) + ((size_t)wave * (TM * TN) * 64 + lane) * 4;   // lane-linear: every store / load instruction moves 1 KiB
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) *(f32x4*)(mine + (size_t)(i * TN + j) * 256) = acc[i][j];
      unsigned* flag = (unsigned*)(smem + SK_FLAG_OFF);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // every wave's slot stores have left
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back must not be overtaken by the ticket (guide, compiler hazard of the release)
        const unsigned old = __hip_atomic_fetch_add(p.sk_ticket + sk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)(nparts - 1);
        if (last) {
          __hip_atomic_store(p.sk_ticket + sk_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                            // this CU's L1 forgets the other contributors' lines
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last ? 1u : 0u;
      }
      __syncthreads();
      if (*flag == 0u) return;
      const float* src0 = slot0 + ((size_t)wave * (TM * TN) * 64 + lane) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = *(const f32x4*)(src0 + (size_t)(i * TN + j) * 256);
      for (int q = 1; q < nparts; ++q) {
        const float* sq = src0 + (size_t)q * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] += *(const f32x4*)(sq + (size_t)(i * TN + j) * 256);
      }
    }
  }

  // (batch, position-in-batch) of a row without a 64-bit division per row (~100 vector instructions each): ONE division per workgroup for
  // its first row, then rows advance by < 256 -- a short subtract loop (rows_per_batch is 17 776 on the DiT; tiny values still terminate)
  long long wg_b = 0, wg_pos = 0;
  if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_QKNORM_ROPE) {
    wg_b = bm0 / p.rows_per_batch;
    wg_pos = bm0 - wg_b * p.rows_per_batch;
  }
  auto row_bp = [&](long long m, long long& b, long long& pos) {
    b = wg_b; pos = wg_pos + (m - bm0);
    while (pos >= p.rows_per_batch) { pos -= p.rows_per_batch; ++b; }
  };
  // ---- epilogue: lane owns row m = .. + (lane & 15), columns n0 + (lane >> 4) * 4 + {0..3}
  constexpr bool STAGED = (TM == 8 && TN == 4 && WM == 2 && WN == 4);
  if (STAGED && p.staged && !is_geglu<EPI>) {   // GEGLU has its own staged form below ([M, N/2] output)
    // The accumulator layout gives 8-byte pieces of 16 different rows per store instruction (32-byte row segments): the store
    // tail was ~24 % of a K = 3072 workgroup.  Stage the wave's 128 x 64 bf16 tile through LDS (row pitch 144 B) and write
    // whole 128-byte row segments with 16-byte lanes; bias / activation / gate are applied in the accumulator layout, the
    // residual add in the row layout (same rounding points as the reference's bf16 tensors: gate * out, then + residual).
    constexpr int ROWB = 144;
    char* wbase = smem + wave * (128 * ROWB);
    // QKNORM_ROPE: the wave's 64 columns are one head (256-wide tiles, 64-column wave tiles); in the row layout below 8 lanes x 8 features
    // hold a row: per-head LayerNorm across those 8 lanes, RoPE on the lane's 4 (even, odd) pairs, Q pre-multiplied -- the arithmetic of
    // qknorm_rope_kernel (norm.hip).  The fp32 cos / sin rows cost 64 B per lane and row group (1 KB per lane over the tile); issued
    // inside the per-row `is a video row` branch they serialised 16 global-load latencies per workgroup.  They are fetched UNCONDITIONALLY
    // instead (text rows read table row 0 and discard it) through a ring of QK_RING row groups of registers, each slot refilled as it is
    // consumed.  MI355X, M = 35 552, N = 9216, K = 3072 (interleaved A/B): 1.99-2.01 ms before, 1.925 ms with a ring of 3 or 4; rings of
    // 5+ make hipcc spill the table registers and lose the gain again.
    constexpr int QK_RING = MRAG_QK_RING, QK_PRE = MRAG_QK_PRE;
    f32x4 qk_tab[EPI == MRAG_EPI_QKNORM_ROPE ? QK_RING : 1][4];
    unsigned qk_video = 0;              // bit g: row group g's row lies past the text rows (RoPE applies)
    int qk_which = 2;                   // 0 = Q, 1 = K, 2 = V columns (wave-uniform)
    bool has_rope = false;
    const int rsub = lane >> 3, chunk = lane & 7;   // row layout: lane -> row (lane >> 3) of an 8-row group, 16-byte chunk (lane & 7)
    auto qk_fetch = [&](int g) {
      if constexpr (EPI == MRAG_EPI_QKNORM_ROPE) {
        const long long m = bm0 + wm * TM * 16 + g * 8 + rsub;
        long long rb, rpos;
        row_bp(m < p.M ? m : p.M - 1, rb, rpos);
        const int pos = (int)rpos - p.rope_text_len;
        if (pos >= 0) qk_video |= 1u << g;
        const long long ro = (long long)(pos > 0 ? pos : 0) * 64 + chunk * 8;
        f32x4(&dst)[4] = qk_tab[g % QK_RING];
        dst[0] = *(const f32x4*)(p.rcos + ro); dst[1] = *(const f32x4*)(p.rcos + ro + 4);
        dst[2] = *(const f32x4*)(p.rsin + ro); dst[3] = *(const f32x4*)(p.rsin + ro + 4);
      }
    };
    if constexpr (EPI == MRAG_EPI_QKNORM_ROPE) {
      qk_which = p.qk_first + (int)((bn0 + wn * TN * 16) / p.qk_D);
      has_rope = p.rcos != nullptr && qk_which < 2;
      if (has_rope) {
#pragma unroll
        for (int g = 0; g < QK_PRE; ++g) qk_fetch(g);
      }
    }
    __syncthreads();   // every wave is done with the operand stages that these per-wave regions overlay
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long m = bm0 + wm * TM * 16 + i * 16 + frag_row;
      const bf16_t* gate = nullptr;
      if constexpr (EPI == MRAG_EPI_GATE_RESID) {
        const long long mc = m < p.M ? m : p.M - 1;
        long long b, pos;
        row_bp(mc, b, pos);
        gate = (pos < p.split ? p.gate0 : p.gate1) + b * p.gate_stride;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        long long n = bn0 + wn * TN * 16 + j * 16 + frag_q * 4;
        n = n < p.N ? n : p.N - 4;   // clamped columns are never stored
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.bias) {
          const u32x2 bb = *(const u32x2*)(p.bias + n);
          v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
          v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = epi_act<EPI>(v[e]);
        if constexpr (EPI == MRAG_EPI_GATE_RESID) {
          const u32x2 gg = *(const u32x2*)(gate + n);
          v[0] *= __uint_as_float(gg[0] << 16); v[1] *= __uint_as_float(gg[0] & 0xffff0000u);
          v[2] *= __uint_as_float(gg[1] << 16); v[3] *= __uint_as_float(gg[1] & 0xffff0000u);
        }
        if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
        }
        u32x2 out;
        out[0] = pack_bf2(v[0], v[1]);
        out[1] = pack_bf2(v[2], v[3]);
        *(u32x2*)(wbase + (i * 16 + frag_row) * ROWB + (j * 16 + frag_q * 4) * 2) = out;
      }
    }
    // row layout: one instruction = 8 x 128 contiguous bytes
    const long long n = bn0 + wn * TN * 16 + chunk * 8;
    // residual epilogues: all 16 residual vectors of the lane are requested up front (the accumulator registers are free once the tile sits
    // in LDS), so the tail of a workgroup pays ONE memory latency instead of four batches of four
    u32x4 rpre[16];
    if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const long long m = bm0 + wm * TM * 16 + g * 8 + rsub;
        rpre[g] = (m < p.M && n + 8 <= p.N) ? *(const u32x4*)(p.resid + m * p.ldr + n) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    bool qk_done = false;
    if constexpr (EPI == MRAG_EPI_QKNORM_ROPE) {
      if (qk_which < 2) {
        qk_done = true;
        const bf16_t* gm = qk_which ? p.kg : p.qg;
        const bf16_t* bt = qk_which ? p.kb : p.qb;
        const int d0 = chunk * 8;
        float gam[8], bet[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gam[e] = 1.f; bet[e] = 0.f; }
        if (gm) {
          const u32x4 graw = *(const u32x4*)(gm + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gam[2 * e] = __uint_as_float(graw[e] << 16); gam[2 * e + 1] = __uint_as_float(graw[e] & 0xffff0000u); }
          if (bt) {
            const u32x4 braw = *(const u32x4*)(bt + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bet[2 * e] = __uint_as_float(braw[e] << 16); bet[2 * e + 1] = __uint_as_float(braw[e] & 0xffff0000u); }
          }
        }
        if (has_rope) {
#pragma unroll
          for (int g = QK_PRE; g < QK_RING; ++g) qk_fetch(g);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          __builtin_amdgcn_sched_barrier(0);   // one row group at a time: hoisting all 16 LDS reads / address computations spills the ring
          const int row = g * 8 + rsub;
          const long long m = bm0 + wm * TM * 16 + row;
          u32x4 val = *(const u32x4*)(wbase + row * ROWB + chunk * 16);
          val = qk_row_math(val, gm != nullptr, bt != nullptr, gam, bet, p.qk_eps, has_rope, (qk_video >> g) & 1u, qk_tab[g % QK_RING], qk_which == 0 && p.q_premul != 1.0f, p.q_premul);
          if (has_rope && g + QK_RING < 16) qk_fetch(g + QK_RING);   // refill the slot just consumed
          if (m < p.M) *(u32x4*)(p.C + m * p.ldc + n) = val;
        }
      }
    }
    if (!qk_done)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int row = g * 8 + rsub;
      const long long m = bm0 + wm * TM * 16 + row;
      u32x4 val = *(const u32x4*)(wbase + row * ROWB + chunk * 16);
      if (m < p.M && n + 8 <= p.N) {
        if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID) {
          const u32x4 rr = rpre[g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(val[e] << 16) + __uint_as_float(rr[e] << 16);
            const float hi = __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u);
            val[e] = pack_bf2(lo, hi);
          }
        }
        *(u32x4*)(p.C + m * p.ldc + n) = val;
      } else if (m < p.M && n + 4 <= p.N) {   // N % 8 == 4 tail
        u32x2 half = {val[0], val[1]};
        if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID) {
          const u32x2 rr = *(const u32x2*)(p.resid + m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float lo = __uint_as_float(half[e] << 16) + __uint_as_float(rr[e] << 16);
            const float hi = __uint_as_float(half[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u);
            half[e] = pack_bf2(lo, hi);
          }
        }
        *(u32x2*)(p.C + m * p.ldc + n) = half;
      }
    }
#ifdef MRAG_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MRAG_GSTAMP(g_exit);
    if (g_gemm_stamp_buf && lane == 0 && blockIdx.x < 1024) {
      g_gemm_stamp_buf[((long long)blockIdx.x * 8 + wave) * 8 + 6] = g_loop0 - g_entry;
      g_gemm_stamp_buf[((long long)blockIdx.x * 8 + wave) * 8 + 7] = g_exit - g_loop1;
    }
#endif
    return;
  }
  // ---- the 256x320 tile (TN = 5: 80 columns = 160 bytes per wave row; the UNets' level-0 convolutions and linears, N = 320 / 960): the same staging in two
  // halves of 64 rows (8 waves x 64 x 176 B fit the operand stages; 128 rows would not).  In the row layout ten lanes hold a 160-byte row segment, so a store /
  // residual-load instruction moves 6.4 whole segments instead of 8-byte pieces of 16 rows -- the direct form cost the residual convolutions ~100 us each at level 0.
  constexpr bool STAGED5 = (TM == 8 && TN == 5 && WM == 2 && WN == 4 && !SK);
  constexpr bool EPI5 = (EPI == MRAG_EPI_NONE || EPI == MRAG_EPI_GELU_TANH || EPI == MRAG_EPI_GELU_ERF || EPI == MRAG_EPI_SILU || EPI == MRAG_EPI_RESID);
  if constexpr (STAGED5 && EPI5) {
    if (p.staged) {
      __syncthreads();   // every wave is done with the operand stages that the per-wave regions overlay
      const long long n0w = bn0 + wn * 80;
      if (n0w + 80 <= p.N) {
        constexpr int ROWB5 = 176;                      // 160 + 16: the 8-byte writes of 16 rows spread over the banks
        char* wbase = smem + wave * (64 * ROWB5);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              const long long n = n0w + j * 16 + frag_q * 4;
              const f32x4 a4 = acc[half * 4 + i][j];
              float v[4] = {a4[0], a4[1], a4[2], a4[3]};
              if (p.bias) {
                const u32x2 bb = *(const u32x2*)(p.bias + n);
                v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
                v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = epi_act<EPI>(v[e]);
              if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
              }
              u32x2 out;
              out[0] = pack_bf2(v[0], v[1]);
              out[1] = pack_bf2(v[2], v[3]);
              *(u32x2*)(wbase + (i * 16 + frag_row) * ROWB5 + (j * 16 + frag_q * 4) * 2) = out;
            }
          }
          // row layout: chunk index c = t * 64 + lane of the half's 64 x 10 sixteen-byte chunks
          const long long mrow0 = bm0 + wm * 128 + half * 64;
#pragma unroll
          for (int tb = 0; tb < 10; tb += 5) {            // (five residual vectors in flight: ten cost the 160 accumulator registers a spill)
            u32x4 rpre[5];
            if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
              for (int t = 0; t < 5; ++t) {
                const unsigned c = (unsigned)((tb + t) * 64 + lane), row = (c * 6554u) >> 16, ch = c - row * 10u;
                const long long m = mrow0 + row;
                rpre[t] = m < p.M ? *(const u32x4*)(p.resid + m * p.ldr + n0w + ch * 8) : u32x4{0u, 0u, 0u, 0u};
              }
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) {
              const unsigned c = (unsigned)((tb + t) * 64 + lane), row = (c * 6554u) >> 16, ch = c - row * 10u;
              const long long m = mrow0 + row;
              u32x4 val = *(const u32x4*)(wbase + row * ROWB5 + ch * 16);
              if constexpr (EPI == MRAG_EPI_RESID) {
                const u32x4 rr = rpre[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float lo = __uint_as_float(val[e] << 16) + __uint_as_float(rr[e] << 16);
                  const float hi = __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u);
                  val[e] = pack_bf2(lo, hi);
                }
              }
              if (m < p.M) *(u32x4*)(p.C + m * p.ldc + n0w + ch * 8) = val;
            }
          }
        }
      } else {
        epilogue_direct<TM, TN, EPI>(p, acc, bm0, bn0, wm * TM * 16, wn * TN * 16, lane);
      }
      return;
    }
  }
  if constexpr (is_geglu<EPI> && TN % 2 != 0) {
    return;   // never dispatched: the value / gate pairing needs an even number of 16-column tiles per wave
  } else if constexpr (is_geglu<EPI>) {
    // W rows arrive interleaved in 16-row groups: [value 16m..16m+15 | gate 16m..16m+15], so the even 16-column MFMA tile
    // holds the values and the odd one the gates of the SAME 16 outputs in the same lanes: C[m, j] = v * gelu_erf(g),
    // C is [M, N/2].  Removes the [M, N] round trip and the separate GEGLU pass (6 % of an SVD / DynamiCrafter step).
    if constexpr (TM == 8 && TN == 4 && WM == 2 && WN == 4) {
      if (p.staged) {
        // LDS-staged form (as above): the wave's 128 x 32 outputs go through LDS (row pitch 80 B) and leave as 64-byte row segments with
        // 16-byte lanes instead of 8-byte pieces of 16 rows per store.  The UNets' GEGLU projections have K = 320 ... 1280, i.e. 5-20
        // K-tiles per workgroup, so the store tail is most of a workgroup's life there.
        constexpr int ROWB = 80;
        char* wbase = smem + wave * (128 * ROWB);
        __syncthreads();   // every wave is done with the operand stages that these per-wave regions overlay
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; j += 2) {
            long long n = bn0 + wn * TN * 16 + j * 16 + frag_q * 4;   // value columns; gates at n + 16
            n = n + 20 <= p.N ? n : p.N - 20;                          // clamped columns are never stored
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float g[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
            if (p.bias) {
              const u32x2 bv = *(const u32x2*)(p.bias + n), bg = *(const u32x2*)(p.bias + n + 16);
              v[0] += __uint_as_float(bv[0] << 16); v[1] += __uint_as_float(bv[0] & 0xffff0000u);
              v[2] += __uint_as_float(bv[1] << 16); v[3] += __uint_as_float(bv[1] & 0xffff0000u);
              g[0] += __uint_as_float(bg[0] << 16); g[1] += __uint_as_float(bg[0] & 0xffff0000u);
              g[2] += __uint_as_float(bg[1] << 16); g[3] += __uint_as_float(bg[1] & 0xffff0000u);
            }
            geglu4<EPI == EPI_GEGLU_TANH>(v, g);
            u32x2 out;
            out[0] = pack_bf2(v[0], v[1]);
            out[1] = pack_bf2(v[2], v[3]);
            *(u32x2*)(wbase + (i * 16 + frag_row) * ROWB + ((j >> 1) * 16 + frag_q * 4) * 2) = out;
          }
        }
        // row layout: lane -> row (lane >> 2) of a 16-row group, 16-byte chunk (lane & 3): one instruction = 16 x 64 contiguous bytes
        const int rs = lane >> 2, ch = lane & 3;
        const long long no = ((bn0 + wn * TN * 16) >> 1) + ch * 8;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
          const int row = g8 * 16 + rs;
          const long long m = bm0 + wm * TM * 16 + row;
          const u32x4 val = *(const u32x4*)(wbase + row * ROWB + ch * 16);
          if (m < p.M && 2 * (no + 8) <= p.N) *(u32x4*)(p.C + m * p.ldc + no) = val;
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long m = bm0 + wm * TM * 16 + i * 16 + frag_row;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const long long n = bn0 + wn * TN * 16 + j * 16 + frag_q * 4;   // value columns; gates at n + 16
        if (n >= p.N) continue;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        float g[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
        if (p.bias) {
          const u32x2 bv = *(const u32x2*)(p.bias + n), bg = *(const u32x2*)(p.bias + n + 16);
          v[0] += __uint_as_float(bv[0] << 16); v[1] += __uint_as_float(bv[0] & 0xffff0000u);
          v[2] += __uint_as_float(bv[1] << 16); v[3] += __uint_as_float(bv[1] & 0xffff0000u);
          g[0] += __uint_as_float(bg[0] << 16); g[1] += __uint_as_float(bg[0] & 0xffff0000u);
          g[2] += __uint_as_float(bg[1] << 16); g[3] += __uint_as_float(bg[1] & 0xffff0000u);
        }
        // the reference rounds both halves of proj(x) to bf16 before the product (nn.Linear output dtype)
        geglu4<EPI == EPI_GEGLU_TANH>(v, g);
        u32x2 out;
        out[0] = pack_bf2(v[0], v[1]);
        out[1] = pack_bf2(v[2], v[3]);
        const long long no = ((bn0 + wn * TN * 16 + j * 16) >> 1) + frag_q * 4;
        *(u32x2*)(p.C + m * p.ldc + no) = out;
      }
    }
    return;
  }
  epilogue_direct<TM, TN, EPI>(p, acc, bm0, bn0, wm * TM * 16, wn * TN * 16, lane);
}

template <int WM, int WN, int TM, int TN, int EPI, int CONV = 0, bool SK = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const GemmP p) {
  if constexpr (!SK) {
    // the grid is the logical tile range [0, gridDim.x): every tile, or the whole rounds in front of a stream-K tail launch
    gemm_tile<WM, WN, TM, TN, EPI, CONV, false>(p, xcd_remap(blockIdx.x, gridDim.x), 0, (int)(p.K / 64), 0, 0);
  } else {
    // the tail launch: sk_units runs of K-tiles share the sk_rem tiles behind logical tile sk_main evenly.  A run touches at most two tiles, and
    // each of its (at most) two pieces is a workgroup of its own -- blockIdx = 2 unit + piece -- so this wrapper is straight-line code: a loop
    // over the pieces made hipcc keep the whole argument struct in SGPRs across it (106 SGPRs, 40-86 spilled VGPRs, reloads inside the K loop)
    // Runs are dealt to XCDs in contiguous chunks, like the tiles of the main launch: consecutive runs work on neighbouring tiles (shared A / W
    // panels) at nearly the same K offset, so an XCD's L2 serves the panels once instead of every CU streaming its own from HBM
    const int nkf = (int)(p.K / 64), I = p.sk_rem * nkf, unit = xcd_remap(blockIdx.x >> 1, p.sk_units);
    const int it0 = (int)(((long long)unit * I) / p.sk_units), it1 = (int)(((long long)(unit + 1) * I) / p.sk_units);
    const int T0 = it0 / nkf, cut = min(it1, (T0 + 1) * nkf);          // the first piece ends at its tile's last K-tile
    const int it = (blockIdx.x & 1) ? cut : it0, end = (blockIdx.x & 1) ? it1 : cut;
    if (it >= end) return;
    const int T = it / nkf;
    gemm_tile<WM, WN, TM, TN, EPI, CONV, true>(p, p.sk_main + T, it - T * nkf, end - it, T, unit);
  }
}

// UNet widths are multiples of 320: N = 320 / 640 / 960 wastes 38 / 17 / 6 % of a 256-wide tile grid, nothing of a 320-wide one
inline bool wide_n_pays(long long N, int tuning = 0) {
  if (tuning & MRAG_GEMM_TUNE_NO_WIDE) return false;
  const long long w256 = (N + 255) / 256 * 256, w320 = (N + 319) / 320 * 320;
  return w320 * 100 < w256 * 90;                   // at least 10 % fewer padded columns
}

// Round quantisation (round 5, tools/unet_op_table.py): one workgroup per CU means a launch costs ceil(tiles / 256) ROUNDS of one tile's time, however full the
// last round is.  The UNets' level-2 problems (M = 16 128 rows, N = 1 280) are 63 x 5 = 315 tiles of 256x256 -- two rounds, the second 23 % full -- but
// 63 x 4 = 252 tiles of 256x320: ONE round of tiles 1.25x as long, 1.6x less time (the 3x3 convolutions at K = 11 520 .. 23 040 and the K = 5 120 FF2 ran at
// 0.30 of the MFMA peak there).  `rounds x tile width` prices a launch; the 320-wide tile is taken when it is at least 15 % cheaper (same bits: same K order).
inline long long round_cost(long long M, long long N, int BN) {
  const long long tiles = ((M + 255) / 256) * ((N + BN - 1) / BN);
  return ((tiles + 255) / 256) * BN;
}
// (tail_rect: the caller will run a small partial last round of the 256x256 grid as its own launch of 128x128 tiles -- plan_tail_rect -- which costs about half
// a round instead of a whole one)
inline bool wide_rounds_pay(long long M, long long N, int tuning = 0, bool tail_rect = false) {
  if (tuning & MRAG_GEMM_TUNE_NO_WIDE) return false;
  long long c256 = round_cost(M, N, 256);
  if (tail_rect) c256 = (((M + 255) / 256) * ((N + 255) / 256) / 256) * 256 + 128;
  return round_cost(M, N, 320) * 100 < c256 * 85;
}
// DynamiCrafter's level 2 (M = 18 432 rows, N = 1 280) is 72 x 5 = 360 tiles of 256x256 -- two rounds, the second 41 % full -- and 288 of 256x320 (two rounds
// of larger tiles: worse).  A 192-row tile (8 waves of 96 x 64; the generic K loop and the direct epilogue, ~8 % behind the pipelined 256x256 loop per FLOP)
// makes it 96 x 5 = 480 tiles: two nearly full rounds of tiles 3/4 the size.  Taken by the convolutions only (K = 3 840 .. 23 040: 626 -> 537 us at K = 11 520,
// 1 291 -> 1 000 us at K = 23 040, the (3,1,1) one 227 -> 178 us); a K = 5 120 LINEAR measured slower on it (259 vs 248 us on the persistent kernel: the
// direct epilogue's 8-byte stores), so linears keep their kernels.
inline bool short_rows_pay(long long M, long long N, int tuning = 0) {
  if (tuning & MRAG_GEMM_TUNE_NO_WIDE) return false;
  const long long t192 = ((M + 191) / 192) * ((N + 255) / 256), t256 = ((M + 255) / 256) * ((N + 255) / 256);
  const long long c192 = ((t192 + 255) / 256) * 192 * 108, c256 = ((t256 + 255) / 256) * 256 * 100;     // rounds x rows per tile x per-FLOP cost
  return c192 * 100 < c256 * 90 && round_cost(M, N, 320) * 100 >= round_cost(M, N, 256) * 85;
}

// the VAEs' finest levels are 128 channels wide: a 256-wide tile grid computes as many masked columns as real ones there
inline bool narrow_n_pays(long long N) {
  const long long r = N % 256;
  return r != 0 && r <= 128;
}

// Stream-K for the partial last round of the 256x256 tile grid (one workgroup per CU, 256 CUs).  The DiT's to_out / FF2 GEMMs are 1 668 tiles =
// 6.52 rounds: the seventh round runs 132 workgroups on 256 CUs for a whole tile's time.  Here the K-tiles of those `rem` tiles are dealt evenly to
// `units` workgroups (all co-resident: <= 256), so the round ends after rem / units of a tile's time plus the partial-sum exchange.
struct SkPlan {
  bool use = false;
  int n_main = 0, rem = 0, units = 0, maxparts = 0;
  size_t bytes = 0;
};
constexpr int SK_CUS = 256, SK_TICKET_BYTES = 1024;
inline SkPlan plan_streamk(long long M, long long N, long long K) {
  SkPlan pl;
  const long long tiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int nk = (int)(K / 64);
  if (tiles < SK_CUS || tiles > (1 << 24) || nk < 16 || nk > 4096) return pl;
  const int rem = (int)(tiles % SK_CUS);
  if (rem == 0 || rem > 208) return pl;           // a nearly full last round has nothing to win (the exchange costs ~15 us)
  pl.rem = rem; pl.n_main = (int)(tiles - rem);
  pl.units = rem * 4 < SK_CUS ? rem * 4 : SK_CUS;  // at most ~4 contributors per tile (+1 where a run straddles)
  const long long I = (long long)rem * nk;
  for (int t = 0, u = 0; t < rem; ++t) {           // contributors per tile: units meeting [t nk, (t + 1) nk)
    while ((long long)(u + 1) * I / pl.units <= (long long)t * nk) ++u;
    int v = u;
    while ((long long)(v + 1) * I / pl.units < (long long)(t + 1) * nk) ++v;
    pl.maxparts = pl.maxparts > v - u + 1 ? pl.maxparts : v - u + 1;
  }
  pl.bytes = SK_TICKET_BYTES + (size_t)rem * pl.maxparts * 256 * 256 * sizeof(float);
  pl.use = true;
  return pl;
}

// ---- the four-wave kernel's epilogue: 128x128 per wave.
// Fast path (the wave's 128 x 128 outputs all inside the matrix, 16-byte aligned rows, one gate vector for the whole wave tile -- every tile of the DiT but
// the last row of tiles): sixteen rows at a time go from the accumulator layout (lane: row (lane & 15), columns (lane >> 4) * 4 + {0..3} of every 16x16
// tile) through two PRIVATE 4-KB LDS buffers of the wave (the 32 KB the operand stages leave free -- those hold the NEXT tile's first two K-tiles by now;
// 16 rows x 256 B, 16-byte chunk c of row r at ((c ^ r) * 16): the 8-byte writes and the 16-byte reads both spread over every bank) into the row layout
// (lane: row (lane >> 4) of four, chunk (lane & 15)) and leave as whole 256-byte row segments -- 32 sixteen-byte stores per lane instead of 64 eight-byte
// pieces of sixteen rows each (direct form: 20 k cycles per tile; every CU of a round stores at the same time and the L2s take a 32-byte partial-line write
// as a transaction of its own).  No predication, no 64-bit multiplies (pointers step by scalar multiples of the leading dimension), no barrier (the
// buffers are the wave's own; row group i is written while i - 1 is read back).  vmcnt is one in-order counter: the residual vectors of row group i + 1 are
// requested BEFORE the stores of i - 1, so waiting for them never waits for a store.  Bias, activation, gate, the bf16 rounding and the residual add happen
// in the accumulator layout: the rounding points of the other epilogues (bit-equal results).
// General path (edge tiles, a sample or text / video boundary inside the wave's rows, unaligned C): 8-byte predicated stores from the accumulator layout.
template <int EPI>
__device__ __forceinline__ void epilogue_w4(const GemmP& p, char* smem, f32x4 (&acc)[8][8], const long long bm0, const long long bn0, const int wave, const int wrow0,
                                            const int wcol0, const int lane_in, const long long Mend) {   // Mend: rows [.., Mend) exist (p.M, or the end of the tile's sample)
  constexpr bool HAS_R = (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID), HAS_G = (EPI == MRAG_EPI_GATE_RESID), QK = (EPI == MRAG_EPI_QKNORM_ROPE);
  // the lane id is laundered through an empty asm: everything below that depends on the lane only (LDS addresses, column offsets, row pointers) would
  // otherwise be hoisted out of the tile loop and kept in registers ACROSS the K loop, whose 128 fragment registers leave no room -- hipcc then spills
  // around the loop and parks the reload's `s_waitcnt vmcnt(0)` in the loop header, which drains the DMA ring once per K-tile (measured: +33 % K-loop time)
  int lane_e = lane_in;
  asm volatile("" : "+v"(lane_e));
  const int lane = lane_e;
  const int frag_row = lane & 15, frag_q = lane >> 4;
  const long long n0 = bn0 + wcol0 + frag_q * 4;                       // + 16 j
  const long long m0 = bm0 + wrow0;                                    // the wave's first row (wave-uniform)
  // sample / position of the wave's first row (GATE_RESID)
  long long g_b = 0, g_pos = 0;
  if constexpr (HAS_G || QK) {
    g_b = m0 / p.rows_per_batch;
    g_pos = m0 - g_b * p.rows_per_batch;
  }
  if (m0 >= Mend || bn0 + wcol0 >= p.N) return;                         // (a wave tile outside the matrix: nothing to store)
  const int mrows = (int)(Mend - m0 < 128 ? Mend - m0 : 128);            // the wave's valid rows (wave-uniform): < 128 in the last row of tiles only
  bool fast = p.staged && bn0 + wcol0 + 128 <= p.N;
  if constexpr (HAS_G) fast = fast && g_pos + mrows - 1 < p.rows_per_batch && ((g_pos < p.split) == (g_pos + mrows - 1 < p.split));
  auto add_resid = [&](u32x2 out, const u32x2 r2) __attribute__((always_inline)) -> u32x2 {
    out[0] = pack_bf2(__uint_as_float(out[0] << 16) + __uint_as_float(r2[0] << 16), __uint_as_float(out[0] & 0xffff0000u) + __uint_as_float(r2[0] & 0xffff0000u));
    out[1] = pack_bf2(__uint_as_float(out[1] << 16) + __uint_as_float(r2[1] << 16), __uint_as_float(out[1] & 0xffff0000u) + __uint_as_float(r2[1] & 0xffff0000u));
    return out;
  };
  auto acc_math = [&](const int i, const int j, const u32x2 b2, const u32x2 g2) __attribute__((always_inline)) -> u32x2 {   // bias, activation, gate, scale; packed (ONE rounding)
    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
    v[0] += __uint_as_float(b2[0] << 16); v[1] += __uint_as_float(b2[0] & 0xffff0000u);
    v[2] += __uint_as_float(b2[1] << 16); v[3] += __uint_as_float(b2[1] & 0xffff0000u);
    if constexpr (EPI == MRAG_EPI_GELU_TANH) {                          // packed form: same bits, 4.5 instead of 7 issue slots per value
      const f32x2 lo = gelu_tanh_f2(f32x2{v[0], v[1]}), hi = gelu_tanh_f2(f32x2{v[2], v[3]});
      v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = epi_act<EPI>(v[e]);
    }
    if constexpr (HAS_G) {
      v[0] *= __uint_as_float(g2[0] << 16); v[1] *= __uint_as_float(g2[0] & 0xffff0000u);
      v[2] *= __uint_as_float(g2[1] << 16); v[3] *= __uint_as_float(g2[1] & 0xffff0000u);
    }
    if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
    }
    return u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  };
  if (fast) {
    u32x2 bias[8], gate[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bias[j] = p.bias ? *(const u32x2*)(p.bias + n0 + 16 * j) : u32x2{0u, 0u};
      if constexpr (HAS_G) gate[j] = *(const u32x2*)((g_pos < p.split ? p.gate0 : p.gate1) + g_b * p.gate_stride + n0 + 16 * j);
      else gate[j] = u32x2{0u, 0u};
    }
    const int r4 = lane >> 4, chunk = lane & 15;                        // row layout
    const bf16_t* rbase = HAS_R ? p.resid + m0 * p.ldr + n0 : nullptr;  // + row * ldr
    bf16_t* cbase = p.C + (m0 + r4) * p.ldc + (bn0 + wcol0 + chunk * 8);
    char* wput = smem + 131072 + wave * 8192 + frag_row * 256 + (frag_q & 1) * 8;
    const char* wget = smem + 131072 + wave * 8192 + r4 * 256;
    const int xput = frag_q >> 1;
    // QKNORM_ROPE (the fused QKV projection): the wave's 128 columns are two heads of ONE third (q_dmodel % 128 == 0), in the row layout the 8 lanes
    // (r4, chunk >> 3) hold a row of a head: qk_row_math on every 16-byte vector between the LDS read and the store.  The fp32 cos / sin rows (64 B per
    // lane and row) come through a ring of three (row-of-four) steps, each slot refilled as it is consumed
    int qk_which = 2;
    bool has_rope = false;
    const bf16_t *gm = nullptr, *bt = nullptr;
    float gam[8], bet[8];
    f32x4 qk_tab[QK ? 3 : 1][4];
    unsigned qk_video = 0;                                              // bit s: this lane's row of step s lies past the text rows
    auto qk_fetch = [&](const int st) __attribute__((always_inline)) {   // step st = rows 4 st .. 4 st + 3 of the wave tile
      if constexpr (QK) {
        const int row = 4 * st + r4;
        long long pos = g_pos + (row < mrows ? row : mrows - 1);
        while (pos >= p.rows_per_batch) pos -= p.rows_per_batch;
        const int rp = (int)pos - p.rope_text_len;
        if (rp >= 0) qk_video |= 1u << st;
        const long long ro = (long long)(rp > 0 ? rp : 0) * 64 + (chunk & 7) * 8;
        f32x4(&dst)[4] = qk_tab[st % 3];
        dst[0] = *(const f32x4*)(p.rcos + ro); dst[1] = *(const f32x4*)(p.rcos + ro + 4);
        dst[2] = *(const f32x4*)(p.rsin + ro); dst[3] = *(const f32x4*)(p.rsin + ro + 4);
      }
    };
    if constexpr (QK) {
      qk_which = p.qk_first + (int)((bn0 + wcol0) / p.qk_D);
      has_rope = p.rcos != nullptr && qk_which < 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) { gam[e] = 1.f; bet[e] = 0.f; }
      if (qk_which < 2) {
        gm = qk_which ? p.kg : p.qg;
        bt = qk_which ? p.kb : p.qb;
        const int d0 = (chunk & 7) * 8;
        if (gm) {
          const u32x4 graw = *(const u32x4*)(gm + d0);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gam[2 * e] = __uint_as_float(graw[e] << 16); gam[2 * e + 1] = __uint_as_float(graw[e] & 0xffff0000u); }
          if (bt) {
            const u32x4 braw = *(const u32x4*)(bt + d0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bet[2 * e] = __uint_as_float(braw[e] << 16); bet[2 * e + 1] = __uint_as_float(braw[e] & 0xffff0000u); }
          }
        }
      }
      if (has_rope) { qk_fetch(0); qk_fetch(1); qk_fetch(2); }
    }
    constexpr int RD = MRAG_W4_RESID_DEPTH;                                               // residual row groups in flight (requested RD - 1 groups ahead of their use; deeper rings measured no faster and cost registers)
    u32x2 rr[RD][8];
    auto fetch = [&](const int i, const int slot) __attribute__((always_inline)) {
      if constexpr (HAS_R) {
        const int row = 16 * i + frag_row;
        const bf16_t* rrow = rbase + (long long)(row < mrows ? row : mrows - 1) * p.ldr;   // rows below the matrix re-read the last valid one
#pragma unroll
        for (int j = 0; j < 8; ++j) rr[slot][j] = *(const u32x2*)(rrow + 16 * j);
      }
    };
    auto put = [&](const int i) __attribute__((always_inline)) {        // row group i -> LDS buffer i & 1 (accumulator layout)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        u32x2 out = acc_math(i, j, bias[j], gate[j]);
        if constexpr (HAS_R) out = add_resid(out, rr[i % RD][j]);
        *(u32x2*)(wput + (i & 1) * 4096 + (((2 * j + xput) ^ frag_row) * 16)) = out;
      }
    };
    auto get_store = [&](const int i) __attribute__((always_inline)) {  // LDS buffer i & 1 -> global (row layout)
      u32x4 val[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) val[q] = *(const u32x4*)(wget + (i & 1) * 4096 + q * 1024 + ((chunk ^ (q * 4 + r4)) * 16));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (QK) {
          const int st = 4 * i + q;
          if (qk_which < 2) {
            val[q] = qk_row_math(val[q], gm != nullptr, bt != nullptr, gam, bet, p.qk_eps, has_rope, (qk_video >> st) & 1u, qk_tab[st % 3], qk_which == 0 && p.q_premul != 1.0f, p.q_premul);
            if (has_rope && st + 3 < 32) qk_fetch(st + 3);             // refill the slot just consumed
          }
        }
        if (16 * i + 4 * q + r4 < mrows) *(u32x4*)(cbase + (long long)(16 * i + 4 * q) * p.ldc) = val[q];
      }
    };
#pragma unroll
    for (int i = 0; i < RD - 1; ++i) fetch(i, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_barrier(0);   // one row group at a time (keeps the live ranges of a group's 32 accumulator reads short)
      // (row groups below the matrix are computed like the others -- their loads re-read the last valid row, only their stores are masked: every load is
      // issued and consumed unconditionally, so hipcc's vmcnt bookkeeping is exact and carries nothing pending into the K loop)
      if (i + RD - 1 < 8) fetch(i + RD - 1, (i + RD - 1) % RD);
      put(i);
      if (i > 0) get_store(i - 1);     // (one wave, in-order LDS: these reads see the writes of iteration i - 1; the writes of i + 1 come after them)
    }
    __builtin_amdgcn_sched_barrier(0);
    get_store(7);
    // (every load of this path was consumed above, so hipcc's vmcnt bookkeeping carries nothing pending into the K loop: an explicit wait here would only
    // expose the latency of the stores just issued -- tests/test_gemm_w4_isa_cpu.py checks the compiled loop)
    return;
  }
  // ---- general path
  if constexpr (QK) return;             // (never dispatched without the fast path's conditions: launch_w4)
  auto ncol = [&](const int j) __attribute__((always_inline)) { const long long n = n0 + 16 * j; return n < p.N ? n : p.N - 4; };   // (N % 4 == 0)
#pragma unroll
  for (int i = 0; i < 8; ++i) {                                         // (fully unrolled: the accumulators are registers, never indexed at run time)
    __builtin_amdgcn_sched_barrier(0);
    const long long m = m0 + i * 16 + frag_row;
    const bool mok = m < Mend;
    const long long mc = mok ? m : Mend - 1;
    const bf16_t* gate = nullptr;
    if constexpr (HAS_G) {
      long long b = g_b, pos = g_pos + (mc - m0);
      while (pos >= p.rows_per_batch) { pos -= p.rows_per_batch; ++b; }
      gate = (pos < p.split ? p.gate0 : p.gate1) + b * p.gate_stride;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long n = ncol(j);
      const u32x2 b2 = p.bias ? *(const u32x2*)(p.bias + n) : u32x2{0u, 0u};
      u32x2 g2 = u32x2{0u, 0u};
      if constexpr (HAS_G) g2 = *(const u32x2*)(gate + n);
      u32x2 out = acc_math(i, j, b2, g2);
      if constexpr (HAS_R) out = add_resid(out, *(const u32x2*)(p.resid + mc * p.ldr + n));
      if (mok && n0 + 16 * j < p.N) *(u32x2*)(p.C + mc * p.ldc + n) = out;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): see the fast path (this path runs on edge tiles only)
}

// ---- the four-wave kernel's GEGLU epilogue (W rows interleaved in 16-row [value | gate] groups: the even 16-column MFMA tile of a pair holds the values, the
// odd one the gates of the same 16 outputs, in the same lanes; C is [M, N / 2]): the wave's 128 x 64 outputs, sixteen rows at a time through two private
// 2-KB LDS buffers (128-byte rows, 16-byte chunk c of row r at ((c ^ (r & 7)) * 16)) into whole 128-byte row segments.  The arithmetic of the 8-wave GEGLU
// epilogues: both halves rounded to bf16 before the product (nn.Linear's output dtype).  Launched only with whole 128-column wave tiles and 16-byte aligned rows.
template <int EPI>
__device__ __forceinline__ void epilogue_w4_geglu(const GemmP& p, char* smem, f32x4 (&acc)[8][8], const long long bm0, const long long bn0, const int wave, const int wrow0,
                                                  const int wcol0, const int lane_in) {
  int lane_e = lane_in;                                                // laundered: see epilogue_w4
  asm volatile("" : "+v"(lane_e));
  const int lane = lane_e;
  const int frag_row = lane & 15, frag_q = lane >> 4;
  const long long m0 = bm0 + wrow0;
  if (m0 >= p.M || bn0 + wcol0 >= p.N) return;                         // a wave tile outside the matrix (N % 128 == 0: a wave's columns are all in or all out)
  const int mrows = (int)(p.M - m0 < 128 ? p.M - m0 : 128);
  const long long n0 = bn0 + wcol0 + frag_q * 4;                       // value columns of pair jj at n0 + 32 jj, gates 16 further
  u32x2 bv[4], bg[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    bv[jj] = p.bias ? *(const u32x2*)(p.bias + n0 + 32 * jj) : u32x2{0u, 0u};
    bg[jj] = p.bias ? *(const u32x2*)(p.bias + n0 + 32 * jj + 16) : u32x2{0u, 0u};
  }
  const int r8 = lane >> 3, chunk = lane & 7;                          // row layout: 8 rows x 8 chunks per instruction
  bf16_t* cbase = p.C + (m0 + r8) * p.ldc + ((bn0 + wcol0) >> 1) + chunk * 8;
  char* wput = smem + 131072 + wave * 8192 + frag_row * 128 + (frag_q & 1) * 8;
  const char* wget = smem + 131072 + wave * 8192 + r8 * 128 + ((chunk ^ r8) * 16);   // rows r8 and r8 + 8 share (row & 7)
  const int xput = frag_q >> 1, sw = frag_row & 7;
  auto put = [&](const int i) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float v[4] = {acc[i][2 * jj][0], acc[i][2 * jj][1], acc[i][2 * jj][2], acc[i][2 * jj][3]};
      float g[4] = {acc[i][2 * jj + 1][0], acc[i][2 * jj + 1][1], acc[i][2 * jj + 1][2], acc[i][2 * jj + 1][3]};
      v[0] += __uint_as_float(bv[jj][0] << 16); v[1] += __uint_as_float(bv[jj][0] & 0xffff0000u);
      v[2] += __uint_as_float(bv[jj][1] << 16); v[3] += __uint_as_float(bv[jj][1] & 0xffff0000u);
      g[0] += __uint_as_float(bg[jj][0] << 16); g[1] += __uint_as_float(bg[jj][0] & 0xffff0000u);
      g[2] += __uint_as_float(bg[jj][1] << 16); g[3] += __uint_as_float(bg[jj][1] & 0xffff0000u);
      geglu4<EPI == EPI_GEGLU_TANH>(v, g);
      *(u32x2*)(wput + (i & 1) * 2048 + (((2 * jj + xput) ^ sw) * 16)) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    }
  };
  auto get_store = [&](const int i) __attribute__((always_inline)) {
    u32x4 val[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) val[q] = *(const u32x4*)(wget + (i & 1) * 2048 + q * 1024);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (16 * i + 8 * q + r8 < mrows) *(u32x4*)(cbase + (long long)(16 * i + 8 * q) * p.ldc) = val[q];
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    put(i);                                                            // (rows below the matrix: computed, not stored)
    if (i > 0) get_store(i - 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  get_store(7);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// logical tile L -> (tile_m, tile_n): groups of group_m m-tiles walked n-major (see gemm_tile)
__device__ __forceinline__ void tile_coords(const GemmP& p, const int L, int& tile_m, int& tile_n) {
  const int gw = p.group_m * p.tiles_n;
  const int first_m = (L / gw) * p.group_m;
  const int gsz = min(p.tiles_m - first_m, p.group_m);
  tile_m = first_m + (L % gw) % gsz;
  tile_n = (L % gw) / gsz;
}

// ---- 256x256 tiles on FOUR waves (one per SIMD, 128x128 per wave, the 256 accumulator registers pinned in AGPRs), PERSISTENT workgroups (one per CU)
// whose K-tile stream runs across tile boundaries, and an instruction-level hand schedule.
// * Per 64-deep K-tile and wave the matrix pipe sees 128 MFMAs with one memory instruction behind every second one: 32 fragment reads (2/3 of the 8-wave
//   tile's LDS bytes per FLOP) and 16 LDS-DMA pieces (scalar base + loop-invariant 32-bit lane offset: no vector address arithmetic in the loop), two
//   barriers, two counted waits.  Every statement of the K loop is volatile inline asm: hipcc only allocates registers.  (Round-3 attempts with 8-MFMA
//   blocks and bursts of reads / pieces lost 3-15 % to the 8-wave loop: one wave per SIMD has no partner to hide a burst behind.)
// * The DMA cursor runs two K-tiles ahead of the MFMAs and simply walks into the workgroup's NEXT tile: when a tile's last MFMA retires, the first two
//   K-tiles of the next one are in LDS and its first fragments in registers, so the matrix pipe idles only for the epilogue's own instructions -- not for
//   a workgroup launch, an address set-up and a cold first fetch per tile (measured on the non-persistent form of this loop: 17 us per tile, 18 % of a
//   K = 3072 tile).  vmcnt is ONE in-order counter for loads and stores: the epilogue first waits for the (old) DMA pieces, then stores, and the first K-tile
//   behind it runs the variant without a counted wait, so no wait in the loop ever stands behind a store that has just been issued.
// LDS: two 64-KB stages [A rows 0..255 | W rows 0..255], 128-byte rows, 16-byte chunk c of row r at ((c ^ (r & 7)) * 16).
// K-tile g of the stream (stage s = g & 1):  k-step 0 MFMAs | reads of (g, k-step 1) .. lgkmcnt(0), BARRIER (stage s is free) .. DMA of K-tile g + 2 -> stage s
//                                            k-step 1 MFMAs | vmcnt (K-tile g + 1 landed), BARRIER .. reads of (g + 1, k-step 0) .. rest of the DMA
// WB (per-sample weights; EPI_NONE): sample b's rows [b rows_per_batch, (b + 1) rows_per_batch) multiply W + b w_bstride -- the motion branch's folded score GEMM, whose
// weights are built from each CFG sample's own motion tokens (attn_processor.py:250-256).  The row-tile grid restarts at every sample (no tile straddles
// two weight matrices; a sample's last tile is clamped / masked at the sample's end), so both samples ride ONE persistent launch: 700 tiles = 2.73 -> 3
// rounds where two launches of 350 paid 2 + 2.
template <int EPI, bool WB = false>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr unsigned STAGE = 65536, WOFF = 32768;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int tiles = p.tile_limit, nk = (int)(p.K / 64), G = (int)gridDim.x;
  // origin of logical tile (tm, tn): first row, one past the last row that exists for it, element offset of its weight matrix
  auto tile_origin = [&](const int tm, long long& bm0, long long& m_end, long long& w_off) __attribute__((always_inline)) {
    if constexpr (WB) {
      const int b = tm / p.wb_tiles_m;
      bm0 = (long long)b * p.rows_per_batch + (long long)(tm - b * p.wb_tiles_m) * 256;
      m_end = (long long)(b + 1) * p.rows_per_batch;
      w_off = (long long)b * p.w_bstride;
    } else {
      bm0 = (long long)tm * 256; m_end = p.M; w_off = 0;
    }
  };
  const int slot = xcd_remap((int)blockIdx.x, G);   // this workgroup's tiles: slot, slot + G, ... (round r of the grid = what a one-tile-per-workgroup launch dispatches)
  // ---- DMA cursor: (tile d_r of this workgroup, K-tile d_kt).  Piece q = wave + 4 i, i = 0..15 (i < 8: A rows 8 q .. 8 q + 7, else W rows 8 (q - 32) ..);
  // lane -> row (lane >> 3) of the piece, source chunk (lane & 7) ^ row
  unsigned voff[16];
  const bf16_t *baseA = p.A, *baseW = p.W;
  int d_r = 0, d_kt = 0;
  bool d_valid = false;
  auto cursor_set = [&](const int r) __attribute__((always_inline)) {
    const int L = r * G + slot;
    d_valid = L < tiles;
    if (!d_valid) return;                 // the stream has ended: the cursor stays where it is (see `advance`)
    int tm, tn;
    tile_coords(p, L, tm, tn);
    long long bm0, m_end, w_off;
    tile_origin(tm, bm0, m_end, w_off);
    const long long bn0 = (long long)tn * 256;
    int lane_c = lane;                      // laundered (see epilogue_w4): nothing lane-derived of this block may stay live across the K loop
    asm volatile("" : "+v"(lane_c));
    const int prow = lane_c >> 3, pchk = (lane_c & 7) ^ prow;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = wave_s + 4 * i;
      long long r8 = (i < 8) ? 8 * q + prow : 8 * (q - 32) + prow;
      const long long lim = (i < 8) ? m_end - bm0 : p.N - bn0;        // clamp: tail rows re-read the tile's last valid row, stores are masked
      r8 = r8 < lim ? r8 : lim - 1;
      voff[i] = (unsigned)((r8 * ((i < 8) ? p.lda : p.ldw) + pchk * 8) * 2);
    }
    baseA = p.A + bm0 * p.lda;
    baseW = p.W + w_off + bn0 * p.ldw;
  };
  // past the end of the stream the cursor stays on its last K-tile: the loop below has ONE instruction stream (one register allocation for the 256 pinned
  // accumulators -- with one body per stream state hipcc spilled accumulators at the joins), so the last two K-tiles of a workgroup re-request a K-tile
  // into a stage nobody reads again (two redundant L2 reads per workgroup) instead of branching around their DMA
  auto advance = [&]() __attribute__((always_inline)) {
    if (!d_valid) return;
    if (d_kt + 1 < nk) { ++d_kt; return; }
    cursor_set(d_r + 1);
    if (d_valid) { ++d_r; d_kt = 0; }
  };
  const unsigned smem_u = (unsigned)(size_t)smem;
  // fragment reads: lane (row r = lane & 15, k-quarter q = lane >> 4) reads chunk (q [+ 4]) ^ (r & 7) of its row
  const unsigned fr = lane & 15, fq = lane >> 4, swz = lane & 7;
  const unsigned c0 = ((fq + 0) ^ swz) * 16, c1 = ((fq + 4) ^ swz) * 16;
  const unsigned rowA = smem_u + (wm * 128 + fr) * 128, rowW = smem_u + WOFF + (wn * 128 + fr) * 128;
  f32x4 acc[8][8];
  u32x4 a0[8], w0[8], a1[8], w1[8];
#define MRAG_W4_MF(I, J, W, A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(W[J]), "v"(A[I]))
#define MRAG_W4_RD(D, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(D) : "v"(ADDR), "n"(OFF) : "memory")
#define MRAG_W4_LGKM0(W, A)                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                             \
                   : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]),   \
                     "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7])    \
                   :: "memory")
  unsigned aw1, aa1, aw0, aa0, stage_u;   // function scope: clang rejects asm operands that name an enclosing LAMBDA's locals from a nested lambda
  auto dma = [&](auto I) __attribute__((always_inline)) {                // piece wave + 4 i of the cursor's K-tile into the stage at LDS address stage_u
    constexpr int i = decltype(I)::value;
    (void)&voff;                          // (clang does not capture a variable that a generic lambda names only in asm operands)
    const bf16_t* sb = (i < 8 ? baseA : baseW) + (long long)d_kt * 64;
    const unsigned lds = stage_u + (unsigned)(wave_s + 4 * i) * 1024u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff[i]), "s"(sb), "s"(lds) : "memory", "m0");
  };
  // one K-tile of the stream.  `counted`: the wait in front of the second barrier (false for the K-tile right behind an epilogue, which waited for every piece)
  auto kstep = [&](const unsigned g, const bool counted) __attribute__((always_inline)) {
    const unsigned so = (g & 1) ? STAGE : 0u, sn = STAGE - so;        // this K-tile's stage offset, the other stage's
    aw1 = rowW + so + c1; aa1 = rowA + so + c1;                       // (g, k-step 1)
    aw0 = rowW + sn + c0; aa0 = rowA + sn + c0;                       // (g + 1, k-step 0)
    stage_u = smem_u + so;
    MRAG_W4_LGKM0(w0, a0);
    // ---- k-step 0: 64 MFMAs on (w0, a0)
    static_for<32>([&](auto S) __attribute__((always_inline)) {
      constexpr int sl = decltype(S)::value, i = (2 * sl) / 8, j = (2 * sl) % 8;
      (void)&acc; (void)&w0; (void)&a0; (void)&w1; (void)&a1; (void)&aw1; (void)&aa1;
      MRAG_W4_MF(i, j, w0, a0);
      if constexpr (sl < 8) MRAG_W4_RD(w1[sl], aw1, sl * 2048);
      else if constexpr (sl < 16) MRAG_W4_RD(a1[sl - 8], aa1, (sl - 8) * 2048);
      else if constexpr (sl == 22) { MRAG_W4_LGKM0(w1, a1); asm volatile("s_barrier" ::: "memory"); }
      else if constexpr (sl >= 23 && (sl & 1)) dma(std::integral_constant<int, (sl - 23) / 2>{});   // pieces 0..4
      MRAG_W4_MF(i, j + 1, w0, a0);
    });
    // ---- k-step 1: 64 MFMAs on (w1, a1)
    if (counted) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // K-tile g + 1 has landed (5 pieces of g + 2 in flight)
    asm volatile("s_barrier" ::: "memory");                            // ... for every wave
    static_for<32>([&](auto S) __attribute__((always_inline)) {
      constexpr int sl = decltype(S)::value, i = (2 * sl) / 8, j = (2 * sl) % 8;
      (void)&acc; (void)&w0; (void)&a0; (void)&w1; (void)&a1; (void)&aw0; (void)&aa0;
      MRAG_W4_MF(i, j, w1, a1);
      if constexpr (sl < 8) MRAG_W4_RD(w0[sl], aw0, sl * 2048);
      else if constexpr (sl < 16) MRAG_W4_RD(a0[sl - 8], aa0, (sl - 8) * 2048);
      else if constexpr (sl >= 16 && sl < 27) dma(std::integral_constant<int, sl - 11>{});          // pieces 5..15
      MRAG_W4_MF(i, j + 1, w1, a1);
    });
  };
#ifdef MRAG_GEMM_STAMPS
  unsigned long long q_acc[6] = {0, 0, 0, 0, 0, 0}, q0, q1, q2, q3, q4, q_in;
  MRAG_GSTAMP(q_in);
#endif
  // ---- prologue: the stream's first two K-tiles, the first fragments
  cursor_set(0);
  if (!d_valid) return;
  stage_u = smem_u;
  static_for<16>([&](auto I) __attribute__((always_inline)) { dma(I); });
  advance();
  stage_u = smem_u + STAGE;
  static_for<16>([&](auto I) __attribute__((always_inline)) { dma(I); });
  advance();
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  aw0 = rowW + c0; aa0 = rowA + c0;
  static_for<8>([&](auto J) __attribute__((always_inline)) { constexpr int j = decltype(J)::value; (void)&w0; (void)&aw0; MRAG_W4_RD(w0[j], aw0, j * 2048); });
  static_for<8>([&](auto J) __attribute__((always_inline)) { constexpr int j = decltype(J)::value; (void)&a0; (void)&aa0; MRAG_W4_RD(a0[j], aa0, j * 2048); });
  unsigned g = 0;
  for (int r = 0;; ++r) {
    const int L = r * G + slot;
    if (L >= tiles) break;
    MRAG_GSTAMP(q0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 4" ::: "memory");   // accumulator writes -> first MFMA (the asm MFMAs are invisible to hipcc's hazard pass)
    MRAG_GSTAMP(q1);
    for (int t = 0; t < nk; ++t, ++g) {
      kstep(g, !(t == 0 && r > 0));
      advance();
    }
    MRAG_GSTAMP(q2);
    // the MFMAs above are invisible to hipcc's hazard pass: the accumulators are read (v_accvgpr_read) only after the matrix pipe has drained; every DMA piece
    // in flight (issued BEFORE the stores below) is waited for here, so the next counted wait in the loop comes two K-tiles after the stores
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    MRAG_GSTAMP(q3);
    int tm, tn;
    tile_coords(p, L, tm, tn);
    long long e_bm0, e_mend, e_woff;
    tile_origin(tm, e_bm0, e_mend, e_woff);
    if constexpr (is_geglu<EPI>) epilogue_w4_geglu<EPI>(p, smem, acc, e_bm0, (long long)tn * 256, wave, wm * 128, wn * 128, lane);
    else epilogue_w4<EPI>(p, smem, acc, e_bm0, (long long)tn * 256, wave, wm * 128, wn * 128, lane, e_mend);
#ifdef MRAG_GEMM_STAMPS
    MRAG_GSTAMP(q4);
    q_acc[0] += q1 - q0; q_acc[1] += q2 - q1; q_acc[2] += q3 - q2; q_acc[3] += q4 - q3; q_acc[4] += 1;
#endif
  }
#ifdef MRAG_GEMM_STAMPS
  MRAG_GSTAMP(q4);
  if (g_gemm_stamp_buf && lane == 0 && blockIdx.x < 1024) {
    unsigned long long* o = g_gemm_stamp_buf + ((long long)blockIdx.x * 8 + wave) * 8;
    for (int k = 0; k < 5; ++k) o[k] = q_acc[k];
    o[5] = q4 - q_in;
  }
#endif
#undef MRAG_W4_MF
#undef MRAG_W4_RD
#undef MRAG_W4_LGKM0
}

template <int WM, int WN, int TM, int TN, int CONV = 0>
int launch_cfg(hipStream_t s, const GemmP& p0, int epi, const SkPlan* sk = nullptr);

// The partial last round of the persistent grid as a RECTANGLE of small tiles.  One workgroup per CU: a launch costs ceil(tiles / 256) rounds however
// full the last one is -- the DiT's FF1 (139 x 48 = 6 672 tiles = 26.06 rounds) pays a 27th round of 79 us for 16 tiles.  Stream-K over those tiles
// measured slower (EXPERIMENTS.md section 3: the runs lose the lock-step that lets an XCD's L2 serve an operand panel once).  When the remainder is SMALL
// the tail was EXPECTED to be cheaper as its own launch of 128x128 tiles (two workgroups per CU, 72 workgroups for FF1's 3 x 6 tiles) -- and MEASURED equal:
// FF1 + GELU 2.236-2.246 ms against 2.245-2.257 ms, the denoise step 550.41 against 550.40 ms (profiles/r6_microbench_items.txt, r6_step_ab_toggles.txt): sixteen
// tiles on sixteen CUs of an otherwise idle chip run well above the loaded rate, so the 27th round costs far less than a round.  OPT-IN
// (MRAG_GEMM_TUNE_TAIL_RECT), kept with its test like the stream-K tail.  The logical tile order walks
// the last group of row tiles column by column, so the last `rem` tiles lie inside the rectangle [last row group] x [last ceil(rem / gsz) tile
// columns]; the persistent launch stops in front of it (GemmP::tile_limit) and the rectangle runs as a plain sub-problem (pointers advanced).  Same K
// order and rounding points: bit-equal to the one-launch form (test_gemm_w4_tail_rectangle).
struct TailRect { bool use = false; int limit = 0; long long r0 = 0, c0 = 0; };
constexpr int W4_TAIL_MAX = 32;
inline TailRect plan_tail_rect(const GemmP& p, int epi) {
  TailRect t;
  if (p.wb_tiles_m || !(p.tuning & MRAG_GEMM_TUNE_TAIL_RECT)) return t;
  if (!(epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_TANH || epi == MRAG_EPI_RESID)) return t;   // (epilogues whose arithmetic does not depend on a row's absolute index)
  const long long tiles = (long long)p.tiles_m * p.tiles_n;
  const int rem = (int)(tiles % SK_CUS);
  if (tiles < 2 * SK_CUS || rem == 0 || rem > W4_TAIL_MAX) return t;
  const int first_m = ((p.tiles_m - 1) / p.group_m) * p.group_m, gsz = p.tiles_m - first_m;
  const int ncols = (rem + gsz - 1) / gsz;
  if (ncols > p.tiles_n) return t;
  t.use = true;
  t.limit = (int)(tiles - (long long)gsz * ncols);
  t.r0 = (long long)first_m * 256; t.c0 = (long long)(p.tiles_n - ncols) * 256;
  return t;
}

// the persistent four-wave launch: one workgroup per CU, 128 KB of LDS
inline int launch_w4(hipStream_t s, const GemmP& p0, int epi) {
  // the K-tile stream walks A and W with 32-bit byte offsets inside a 256-row panel: (row * ld + chunk) * 2 with row <= 255 must stay below 4 GiB
  // (a view with a huge leading dimension goes to the 8-wave kernel, whose row pointers are 64-bit)
  if (256LL * (p0.lda > p0.ldw ? p0.lda : p0.ldw) * 2 >= (1LL << 32)) return MRAG_ENOTSUP;
  GemmP p = p0;
  const bool wb = p.w_bstride != 0;
  if (wb) {                                     // per-sample weights: the row-tile grid restarts at every sample
    if (epi != MRAG_EPI_NONE || p.rows_per_batch <= 0 || p.M % p.rows_per_batch != 0) return MRAG_ENOTSUP;
    p.wb_tiles_m = (int)((p.rows_per_batch + 255) / 256);
    p.tiles_m = (int)(p.M / p.rows_per_batch) * p.wb_tiles_m;
  } else {
    p.wb_tiles_m = 0;
    p.tiles_m = (int)((p.M + 255) / 256);
  }
  p.tiles_n = (int)((p.N + 255) / 256);
  p.group_m = ((p.tuning >> 8) & 0xff) ? ((p.tuning >> 8) & 0xff) : 4;
  const long long tiles = (long long)p.tiles_m * p.tiles_n;
  // the LDS-staged epilogue needs 16-byte aligned rows of C (and of the residual); otherwise the direct 8-byte store path runs
  p.staged = (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0) && (!p.resid || ((p.ldr % 8 == 0) && (((uintptr_t)p.resid & 15) == 0)));
  if (p.tuning & MRAG_GEMM_TUNE_NO_STAGED) p.staged = 0;
  const TailRect tail = plan_tail_rect(p, epi);
  p.tile_limit = tail.use ? tail.limit : (int)tiles;
  const dim3 grid((unsigned)(p.tile_limit < SK_CUS ? p.tile_limit : SK_CUS)), block(256);
  const size_t lds = 131072 + 32768;   // two operand stages + 8 KB of epilogue staging per wave: all 160 KB of a CU
#define MRAG_W4_LAUNCH(...)                                                                            \
  {                                                                                                    \
    auto kfn = gemm_w4_kernel<__VA_ARGS__>;                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                          \
  }
  switch (epi) {
    case MRAG_EPI_NONE:
      if (wb) MRAG_W4_LAUNCH(MRAG_EPI_NONE, true) else MRAG_W4_LAUNCH(MRAG_EPI_NONE)
      break;
    case MRAG_EPI_GELU_TANH: MRAG_W4_LAUNCH(MRAG_EPI_GELU_TANH) break;
    case MRAG_EPI_RESID: MRAG_W4_LAUNCH(MRAG_EPI_RESID) break;
    case MRAG_EPI_GATE_RESID: MRAG_W4_LAUNCH(MRAG_EPI_GATE_RESID) break;
    case MRAG_EPI_GEGLU:
    case EPI_GEGLU_TANH:                  // whole 128-column wave tiles, aligned rows of C [M, N / 2]
      if (!p.staged || p.N % 128 != 0) return MRAG_ENOTSUP;
      if (epi == MRAG_EPI_GEGLU) MRAG_W4_LAUNCH(MRAG_EPI_GEGLU) else MRAG_W4_LAUNCH(EPI_GEGLU_TANH)
      break;
    case MRAG_EPI_QKNORM_ROPE:            // fast epilogue path only: whole 128-column wave tiles inside one third, aligned rows
      if (!p.staged || p.N % 128 != 0 || p.qk_D % 128 != 0) return MRAG_ENOTSUP;
      MRAG_W4_LAUNCH(MRAG_EPI_QKNORM_ROPE)
      break;
    default: return MRAG_ENOTSUP;
  }
#undef MRAG_W4_LAUNCH
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(epi == MRAG_EPI_QKNORM_ROPE ? MRAG_K_GEMM_W4_QKNORM_ROPE : (epi == MRAG_EPI_GEGLU || epi == EPI_GEGLU_TANH) ? MRAG_K_GEMM_W4_GEGLU : wb ? MRAG_K_GEMM_W4_BATCHED_W : MRAG_K_GEMM_W4);
  if (tail.use) {                               // the rectangle behind the whole rounds: rows [r0, M) x columns [c0, N) on 128x128 tiles
    GemmP t = p0;
    t.A = p0.A + tail.r0 * p0.lda; t.W = p0.W + tail.c0 * p0.ldw; t.C = p0.C + tail.r0 * p0.ldc + tail.c0;
    if (p0.bias) t.bias = p0.bias + tail.c0;
    if (p0.resid) t.resid = p0.resid + tail.r0 * p0.ldr + tail.c0;
    t.M = p0.M - tail.r0; t.N = p0.N - tail.c0;
    const int rc = launch_cfg<2, 2, 4, 4>(s, t, epi);
    if (rc != MRAG_OK) return rc;
    MRAG_COUNT(MRAG_K_GEMM_W4_TAIL_RECT);
  }
  return MRAG_OK;
}


// ---- K = 320, N a multiple of 320: the UNets' level-0 linears (to_q / to_out / proj_in / proj_out: N = 320; the fused QKV: N = 960; the GEGLU projection:
// N = 2 560 -- 75-95 launches per CFG step over 258 048 / 294 912 pixel rows).  0.05-0.5 TFLOP against 0.4-0.9 GB each: memory-bound -- but a 256x320 or
// 256x256 tile re-stages 160-200 KB of weights per tile and runs load, five short K-tiles and store strictly one after the other with one workgroup per
// CU: 2.4-2.5 TB/s on the plain shapes, 1.4 TB/s with GEGLU (tools/unet_op_table.py).  Here the weight never moves.  A workgroup owns ONE 320-column slice of
// W: its ten waves hold it as MFMA operands in REGISTERS (wave w: slice columns 32 w .. 32 w + 31 = 2 column tiles x 10 k-steps = 80 VGPRs) for its
// lifetime, and streams 64-row activation tiles through a two-stage LDS-DMA ring (40 KB per stage, the K-tile-major swizzled image of the other kernels);
// the outputs leave through an LDS staging tile as whole rows of the slice, the residual added in the row layout.  N / 320 slices x G persistent
// workgroups; block id = slice * G + g with G a multiple of 8, so the workgroups that read the SAME activation tiles (equal g) share an XCD's L2 and the
// activations come from HBM once.  GEGLU: a wave's two column tiles are the value and the gate tile of the same 16 outputs (the 16-row [value | gate]
// interleave of the other GEGLU epilogues).  Same K order and rounding points as the other tiles: bit-equal results.
constexpr int SK320_ROWS = 64, SK320_STAGE = SK320_ROWS * 640, SK320_CPITCH = 656;   // C staging: 64 rows x <= 640 B, pitch 656 B (8-byte writes of 16 rows spread over the banks)

template <int EPI>
__global__ __launch_bounds__(640) void gemm_k320_kernel(const GemmP p) {
  constexpr bool GEGLU = is_geglu<EPI>;
  constexpr int CW = GEGLU ? 160 : 320, CH = CW / 8;        // columns / 16-byte chunks of a staged output row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* cst = smem + 2 * SK320_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4, swz = lane & 7;
  const int slices = (int)(p.N / 320), G = (int)gridDim.x / slices;
  const int slice = (int)blockIdx.x / G, g = (int)blockIdx.x - slice * G;
  const int n0 = slice * 320 + wave * 32;                   // the wave's first column of W / bias
  // the wave's weight fragments: W[n0 + 16 j + fr][32 ks + 8 fq .. + 7]
  bf16x8 wf[2][10];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) wf[j][ks] = *(const bf16x8*)(p.W + (long long)(n0 + 16 * j + fr) * p.ldw + 32 * ks + 8 * fq);
  u32x2 bias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias[j] = p.bias ? *(const u32x2*)(p.bias + n0 + 16 * j + 4 * fq) : u32x2{0u, 0u};
  const long long c0 = GEGLU ? slice * 160 : slice * 320;   // the slice's first output column
  const int tiles = (int)((p.M + SK320_ROWS - 1) / SK320_ROWS);
  // DMA: piece q = wave + 10 i (i < 4) of a tile: K-tile q / 8, rows 8 (q % 8) .. + 7; lane -> row (lane >> 3), source chunk (lane & 7) ^ row
  auto issue = [&](const int tile, const int stage) {
    const long long m0 = (long long)tile * SK320_ROWS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave + 10 * i, kt = q >> 3;
      long long row = m0 + 8 * (q & 7) + (lane >> 3);
      row = row < p.M ? row : p.M - 1;                      // tail rows re-read the last valid row; their stores are masked
      glds16(p.A + row * p.lda + kt * 64 + (((lane & 7) ^ (lane >> 3)) * 8), smem + stage * SK320_STAGE + q * 1024);
    }
  };
  int tile = g;
  if (tile < tiles) issue(tile, 0);
  for (int it = 0; tile < tiles; ++it, tile += G) {
    const int stage = it & 1;
    const bool more = tile + G < tiles;
    if (more) {
      issue(tile + G, stage ^ 1);                           // (the other stage was released by the barrier that closed the previous iteration)
      // INVARIANT of the counted wait (as in topk_mfma_kernel): vmcnt retires in order and counts every vector-memory operation of the wave -- the residual loads and C
      // stores of the previous tile's epilogue are all issued BEFORE the next tile's four pieces, so "4 outstanding" means exactly those pieces.  No global access may be
      // moved between `issue` and this wait.  -DMRAG_DIAG_VMCNT0 replaces it by vmcnt(0): the results must not change.
#ifdef MRAG_DIAG_VMCNT0
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // this tile's four pieces have landed, the next tile's four are in flight
#endif
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const char* st = smem + stage * SK320_STAGE;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
      const int off = (ks >> 1) * 8192 + (((fq + 4 * (ks & 1)) ^ swz) * 16);
#pragma unroll
      for (int i = 0; i < 4; i += 2) {                      // two row tiles at a time: 8 fragment registers live (158 VGPRs at three waves per SIMD)
        const bf16x8 a0 = *(const bf16x8*)(st + off + (i * 16 + fr) * 128), a1 = *(const bf16x8*)(st + off + ((i + 1) * 16 + fr) * 128);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], a0, acc[i][j], 0, 0, 0);
          acc[i + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], a1, acc[i + 1][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                    // (keeps hipcc from hoisting the next k-steps' fragment reads: they would spill)
    }
    // ---- epilogue: bias (+ scale | GEGLU), ONE rounding to bf16 in the accumulator layout, staged to rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[j][0] = acc[i][j][0] + __uint_as_float(bias[j][0] << 16); v[j][1] = acc[i][j][1] + __uint_as_float(bias[j][0] & 0xffff0000u);
        v[j][2] = acc[i][j][2] + __uint_as_float(bias[j][1] << 16); v[j][3] = acc[i][j][3] + __uint_as_float(bias[j][1] & 0xffff0000u);
      }
      if constexpr (GEGLU) {                                // column tile 0: values, tile 1: the gates of the same 16 outputs
        geglu4<EPI == EPI_GEGLU_TANH>(v[0], v[1]);
        *(u32x2*)(cst + (i * 16 + fr) * SK320_CPITCH + (wave * 16 + 4 * fq) * 2) = u32x2{pack_bf2(v[0][0], v[0][1]), pack_bf2(v[0][2], v[0][3])};
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (EPI == MRAG_EPI_RESID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] *= p.acc_scale;
          }
          *(u32x2*)(cst + (i * 16 + fr) * SK320_CPITCH + (wave * 32 + 16 * j + 4 * fq) * 2) = u32x2{pack_bf2(v[j][0], v[j][1]), pack_bf2(v[j][2], v[j][3])};
        }
      }
    }
    __syncthreads();
    const long long m0 = (long long)tile * SK320_ROWS;
#pragma unroll 2
    for (int idx = tid; idx < SK320_ROWS * CH; idx += 640) {  // whole rows of the slice: CH sixteen-byte chunks per row (two at a time: the 80 weight registers stay live)
      const int row = idx / CH, ch = idx - row * CH;
      const long long m = m0 + row;
      u32x4 val = *(const u32x4*)(cst + row * SK320_CPITCH + ch * 16);
      if (m < p.M) {
        if constexpr (EPI == MRAG_EPI_RESID) {
          const u32x4 rr = *(const u32x4*)(p.resid + m * p.ldr + c0 + ch * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(val[e] << 16) + __uint_as_float(rr[e] << 16);
            const float hi = __uint_as_float(val[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u);
            val[e] = pack_bf2(lo, hi);
          }
        }
        *(u32x4*)(p.C + m * p.ldc + c0 + ch * 8) = val;
      }
    }
    __syncthreads();                                        // the staging tile and this stage are free again
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Few-row GEMM (M <= 256: CAMA's Perceiver latents and encoder tokens, 25-251 rows; the retrieval query's text embedder, 16 rows).  On the tiled kernels such a
// problem is 8-96 workgroups, each walking the whole K through an LDS ring with a barrier per K-tile: 15-45 us per launch whatever the size.
// Here a workgroup owns a 32 x 64 output tile and its EIGHT waves split K between them (wave w takes the 32-deep K-steps w, w + 8, ...): every wave streams its
// fragments straight from L2 / HBM into MFMA operands -- no LDS staging, no barrier in the loop, dozens of independent 16-byte loads in flight per lane -- and the
// eight partial tiles meet once in LDS, where they are added in wave order (a fixed order: bit-reproducible; NOT the summation order of the tiled kernels, so
// the last bits differ from theirs).  Grid = ceil(M / 32) x ceil(N / 64) workgroups: 128-512 for CAMA's shapes.  Epilogue and rounding points: epilogue_direct's.
// Long K (>= 2 048: the feed-forward's second projection) takes SIXTEEN waves over a 32 x 32 tile instead: half the K-steps per wave, twice the workgroups.
constexpr int SKM_ROWS = 32;

// LNA (round 6): the LayerNorm in FRONT of the projection rides in the A load -- CAMA's Perceiver layers run `to_q(norm2(latents))` and `ff1(ln(latents))` over 250
// rows, where the LayerNorm was a 6 us launch of its own in a chain of dependent launches.  A workgroup reads all of K for its 32 rows anyway: its waves first
// compute the rows' statistics (4 or 2 rows per wave, the arithmetic of layernorm_kernel in norm.hip lane for lane: per-lane sums over idx = (c 64 + lane) 8,
// the wave butterfly, mean, then the squared deviations -- so the normalised bf16 values are the SAME BITS the separate kernel writes), park them in LDS, and every
// A fragment is normalised, scaled, shifted and rounded to bf16 in registers before its MFMAs.  Results are bit-identical to LayerNorm kernel + GEMM.
template <int EPI, int NWV, int COLS, bool LNA = false>
__global__ __launch_bounds__(64 * NWV) void gemm_skinny_kernel(const GemmP p) {
  constexpr int TJ = COLS / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = (float*)smem;                                   // [NWV][SKM_ROWS][COLS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long m0 = (long long)blockIdx.y * SKM_ROWS, n0 = (long long)blockIdx.x * COLS;
  const int r = lane & 15, kc = (lane >> 4) * 8;
  const bf16_t* ap[2];
  const bf16_t* wp[TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long m = m0 + i * 16 + r;
    m = m < p.M ? m : p.M - 1;                                  // (rows / columns past the problem are computed on a clamped copy and never stored)
    ap[i] = p.A + m * p.lda + kc;
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    long long n = n0 + j * 16 + r;
    n = n < p.N ? n : p.N - 1;
    wp[j] = p.W + n * p.ldw + kc;
  }
  f32x4 acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nks = (int)(p.K / 32);
  float ln_mean[2] = {0.f, 0.f}, ln_rstd[2] = {1.f, 1.f};
  if constexpr (LNA) {
    constexpr int RPWV = SKM_ROWS / NWV;                          // rows whose statistics this wave computes
    const int D = (int)p.K;
#pragma unroll
    for (int rr = 0; rr < RPWV; ++rr) {
      const int row_l = wave * RPWV + rr;
      long long m = m0 + row_l;
      m = m < p.M ? m : p.M - 1;
      const bf16_t* x = p.A + m * p.lda;
      float sum = 0.f;
      for (int c = 0; c * 512 < D; ++c) {
        const int idx = (c * 64 + lane) * 8;
        if (idx < D) {
          const u32x4 raw = *(const u32x4*)(x + idx);
#pragma unroll
          for (int e = 0; e < 4; ++e) { sum += __uint_as_float(raw[e] << 16); sum += __uint_as_float(raw[e] & 0xffff0000u); }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum / (float)D;
      float sq = 0.f;
      for (int c = 0; c * 512 < D; ++c) {
        const int idx = (c * 64 + lane) * 8;
        if (idx < D) {
          const u32x4 raw = *(const u32x4*)(x + idx);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d0 = __fsub_rn(__uint_as_float(raw[e] << 16), mean), d1 = __fsub_rn(__uint_as_float(raw[e] & 0xffff0000u), mean);
            sq = __builtin_fmaf(d0, d0, sq); sq = __builtin_fmaf(d1, d1, sq);
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (lane == 0) { part[2 * row_l] = mean; part[2 * row_l + 1] = rsqrtf(sq / (float)D + p.lna_eps); }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) { ln_mean[i] = part[2 * (i * 16 + r)]; ln_rstd[i] = part[2 * (i * 16 + r) + 1]; }
    __syncthreads();                                              // (the partial tiles reuse this LDS behind the K loop)
  }
  auto steps = [&](auto U, const int ks0) __attribute__((always_inline)) {       // U K-steps of this wave from ks0: all loads first ((2 + TJ) U independent 16-byte loads in flight)
    constexpr int u_n = decltype(U)::value;
    bf16x8 a[u_n][2], w[u_n][TJ];
    u32x4 lg[LNA ? u_n : 1], lb[LNA ? u_n : 1];
#pragma unroll
    for (int u = 0; u < u_n; ++u) {
      const int k = (ks0 + u * NWV) * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[u][i] = *(const bf16x8*)(ap[i] + k);
#pragma unroll
      for (int j = 0; j < TJ; ++j) w[u][j] = *(const bf16x8*)(wp[j] + k);
      if constexpr (LNA) {
        lg[u] = p.lna_g ? *(const u32x4*)(p.lna_g + k + kc) : u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        lb[u] = p.lna_b ? *(const u32x4*)(p.lna_b + k + kc) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    if constexpr (LNA) {                                           // o = (v - mean) * rstd [* gamma] [+ beta], ONE rounding to bf16: layernorm_kernel's arithmetic
#pragma unroll
      for (int u = 0; u < u_n; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 raw = __builtin_bit_cast(u32x4, a[u][i]);
          u32x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o0 = __fmul_rn(__fsub_rn(__uint_as_float(raw[e] << 16), ln_mean[i]), ln_rstd[i]), o1 = __fmul_rn(__fsub_rn(__uint_as_float(raw[e] & 0xffff0000u), ln_mean[i]), ln_rstd[i]);
            if (p.lna_g) { o0 = __fmul_rn(o0, __uint_as_float(lg[u][e] << 16)); o1 = __fmul_rn(o1, __uint_as_float(lg[u][e] & 0xffff0000u)); }
            if (p.lna_b) { o0 = __fadd_rn(o0, __uint_as_float(lb[u][e] << 16)); o1 = __fadd_rn(o1, __uint_as_float(lb[u][e] & 0xffff0000u)); }
            o4[e] = pack_bf2(o0, o1);
          }
          a[u][i] = __builtin_bit_cast(bf16x8, o4);
        }
    }
#pragma unroll
    for (int u = 0; u < u_n; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j], a[u][i], acc[i][j], 0, 0, 0);
  };
  int ks = wave;
  constexpr int UB = (LNA && NWV == 16) ? 2 : 4;   // K-steps whose loads are in flight together (sixteen waves leave 128 registers per lane: four steps + the LayerNorm's operands spilled)
  for (; ks + (UB - 1) * NWV < nks; ks += UB * NWV) steps(std::integral_constant<int, UB>{}, ks);
  for (; ks < nks; ks += NWV) steps(std::integral_constant<int, 1>{}, ks);
  // accumulator layout: lane owns row i * 16 + (lane & 15), columns j * 16 + (lane >> 4) * 4 + {0..3}
  float* mine = part + wave * (SKM_ROWS * COLS);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) *(f32x4*)(mine + (i * 16 + r) * COLS + j * 16 + (lane >> 4) * 4) = acc[i][j];
  __syncthreads();
  // one thread per 4 consecutive columns of the 32 x COLS tile
  if (tid >= SKM_ROWS * COLS / 4) return;
  const int row = tid / (COLS / 4), col = (tid % (COLS / 4)) * 4;
  const long long m = m0 + row, n = n0 + col;
  if (m >= p.M || n >= p.N) return;
  f32x4 t = *(const f32x4*)(part + row * COLS + col);
#pragma unroll
  for (int w8 = 1; w8 < NWV; ++w8) {
    const f32x4 u = *(const f32x4*)(part + w8 * (SKM_ROWS * COLS) + row * COLS + col);
    t[0] += u[0]; t[1] += u[1]; t[2] += u[2]; t[3] += u[3];
  }
  float v[4] = {t[0], t[1], t[2], t[3]};
  if (p.bias) {
    const u32x2 bb = *(const u32x2*)(p.bias + n);
    v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
    v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = epi_act<EPI>(v[e]);
  if constexpr (EPI == MRAG_EPI_RESID) {
    const u32x2 rr = *(const u32x2*)(p.resid + m * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
    v[0] = bf_round(v[0]) + __uint_as_float(rr[0] << 16); v[1] = bf_round(v[1]) + __uint_as_float(rr[0] & 0xffff0000u);
    v[2] = bf_round(v[2]) + __uint_as_float(rr[1] << 16); v[3] = bf_round(v[3]) + __uint_as_float(rr[1] & 0xffff0000u);
  }
  u32x2 out;
  out[0] = pack_bf2(v[0], v[1]);
  out[1] = pack_bf2(v[2], v[3]);
  *(u32x2*)(p.C + m * p.ldc + n) = out;
}

// where the few-row kernel runs: measured against the 128 x 128 tile on MI355X (tools/skinny_sweep.py, profiles/r5_gemm_skinny_sweep.txt)
inline bool skinny_applies(const mrag_gemm_args* a, int epi) {
  // (W is read once per 32-row tile, through L2: past ~134 MB of such reads the tiled kernel's larger tiles win again -- [256 x 10240 x 4096] 115 vs 50 us,
  // [256 x 4096 x 4096] 49 vs 47, [128 x 4096 x 4096] 27 vs 47, [250 x 1024 x 4096] 25 vs 46; a single row tile streams any weight once: the UNets' batched
  // time-embedding projection, 32 x 36 480 x 1 280)
  const long long row_tiles = (a->M + SKM_ROWS - 1) / SKM_ROWS;
  return a->M <= 256 && row_tiles * a->N * a->K <= 8LL * 4096 * 2048 && a->K >= 256 && a->N >= 256 &&
         !(a->tuning & MRAG_GEMM_TUNE_NO_SKINNY) && ((a->tuning >> 4) & 0xf) == 0 &&
         (epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_TANH || epi == MRAG_EPI_GELU_ERF || epi == MRAG_EPI_SILU || epi == MRAG_EPI_RESID) &&
         (!a->bias || (((uintptr_t)a->bias) & 7) == 0);
}

inline int launch_skinny(hipStream_t s, const GemmP& p, int epi) {
  if (p.lna && !(epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_ERF)) return MRAG_ENOTSUP;   // the two forms CAMA runs: to_q(norm2(.)), gelu(ff1(ln(.)))
  // 16 waves x 32 columns where K is long and the 8-wave grid would leave most CUs idle ([250 x 1024 x 4096] 25.5 -> 20.1 us, [64 x 4096 x 4096] 26.1 -> 20.7;
  // a grid that already fills the chip loses: [128 x 4096 x 4096] 26.6 -> 34.6; profiles/r5_gemm_skinny_sweep.txt).  MRAG_GEMM_TUNE_SKINNY_8: the 8-wave form always (A/B runs)
  const long long wg8 = ((p.M + SKM_ROWS - 1) / SKM_ROWS) * ((p.N + 63) / 64);
  const bool sixteen = p.K >= 2048 && p.N >= 1024 && wg8 < 256 && !(p.tuning & MRAG_GEMM_TUNE_SKINNY_8);
  const int cols = sixteen ? 32 : 64, nw = sixteen ? 16 : 8;
  const dim3 grid((unsigned)((p.N + cols - 1) / cols), (unsigned)((p.M + SKM_ROWS - 1) / SKM_ROWS)), block(64 * nw);
  const size_t lds = (size_t)nw * SKM_ROWS * cols * sizeof(float);
#define MRAG_SKINNY_LAUNCH(E, W, ...)                                                                  \
  {                                                                                                    \
    auto kfn = gemm_skinny_kernel<E, W, __VA_ARGS__>;                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                          \
  }
#define MRAG_SKINNY_CASE(E)                                                                            \
  case E:                                                                                              \
    if (sixteen) MRAG_SKINNY_LAUNCH(E, 16, 32) else MRAG_SKINNY_LAUNCH(E, 8, 64)                        \
    break;
  if (p.lna) {
    if (epi == MRAG_EPI_NONE) { if (sixteen) MRAG_SKINNY_LAUNCH(MRAG_EPI_NONE, 16, 32, true) else MRAG_SKINNY_LAUNCH(MRAG_EPI_NONE, 8, 64, true) }
    else { if (sixteen) MRAG_SKINNY_LAUNCH(MRAG_EPI_GELU_ERF, 16, 32, true) else MRAG_SKINNY_LAUNCH(MRAG_EPI_GELU_ERF, 8, 64, true) }
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_GEMM_SKINNY_LNA);
    return MRAG_OK;
  }
  switch (epi) {
    MRAG_SKINNY_CASE(MRAG_EPI_NONE)
    MRAG_SKINNY_CASE(MRAG_EPI_GELU_TANH)
    MRAG_SKINNY_CASE(MRAG_EPI_GELU_ERF)
    MRAG_SKINNY_CASE(MRAG_EPI_SILU)
    MRAG_SKINNY_CASE(MRAG_EPI_RESID)
    default: return MRAG_ENOTSUP;
  }
#undef MRAG_SKINNY_CASE
#undef MRAG_SKINNY_LAUNCH
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_GEMM_SKINNY);
  return MRAG_OK;
}

inline bool k320_applies(const mrag_gemm_args* a, int epi) {
  // (the GEGLU projection, N = 2 560 = eight slices, is instantiated and bit-equal but NOT dispatched: 758 vs 683 us -- a tile costs ~12 k cycles here whatever
  // the slice count, so eight passes over the activations lose against the persistent four-wave kernel; N = 960 gains 8 %, N = 320 40 %)
  return a->K == 320 && a->N % 320 == 0 && a->N <= 960 && a->M >= 16384 && (epi == MRAG_EPI_NONE || epi == MRAG_EPI_RESID) &&
         !(a->tuning & (MRAG_GEMM_TUNE_NO_WIDE | MRAG_GEMM_TUNE_NO_STAGED | MRAG_GEMM_TUNE_GEGLU_NO_STAGED | MRAG_GEMM_TUNE_STREAMK)) &&
         a->ldc % 8 == 0 && (((uintptr_t)a->C) & 15) == 0 && (!a->resid || (a->ldr % 8 == 0 && (((uintptr_t)a->resid) & 15) == 0)) && (!a->bias || (((uintptr_t)a->bias) & 7) == 0);
}

inline int launch_k320(hipStream_t s, const GemmP& p, int epi) {
  const int tiles = (int)((p.M + SK320_ROWS - 1) / SK320_ROWS), slices = (int)(p.N / 320);
  int G = (SK_CUS / slices) & ~7;                           // persistent workgroups per slice: a multiple of 8 (block id % 8 = XCD: equal g -> one XCD)
  if (G > tiles) G = tiles >= 8 ? (tiles & ~7) : tiles;
  const dim3 grid((unsigned)(slices * G)), block(640);
  const size_t lds = 2 * SK320_STAGE + SK320_ROWS * SK320_CPITCH;
#define MRAG_K320_CASE(E)                                                                              \
  case E: {                                                                                            \
    auto kfn = gemm_k320_kernel<E>;                                                                    \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                          \
    break;                                                                                             \
  }
  switch (epi) {
    MRAG_K320_CASE(MRAG_EPI_NONE)
    MRAG_K320_CASE(MRAG_EPI_RESID)
    default: return MRAG_ENOTSUP;                           // (the GEGLU form of the template was measured and is not instantiated: k320_applies)
  }
#undef MRAG_K320_CASE
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_GEMM_N320K320);
  return MRAG_OK;
}

template <int WM, int WN, int TM, int TN, int CONV>
int launch_cfg(hipStream_t s, const GemmP& p0, int epi, const SkPlan* sk) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  GemmP p = p0;
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (int)((p.N + BN - 1) / BN);
  p.group_m = ((p.tuning >> 8) & 0xff) ? ((p.tuning >> 8) & 0xff) : 4;
  const dim3 grid(sk ? sk->n_main : p.tiles_m * p.tiles_n), block(WM * WN * 64);   // with a stream-K plan: the whole rounds here, the tail as a second launch
  // the LDS-staged epilogue needs 16-byte aligned rows of C (and of the residual); otherwise the direct 8-byte store path runs
  p.staged = (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0) && (!p.resid || ((p.ldr % 8 == 0) && (((uintptr_t)p.resid & 15) == 0)));
  if ((p.tuning & MRAG_GEMM_TUNE_NO_STAGED) || ((epi == MRAG_EPI_GEGLU || epi == EPI_GEGLU_TANH) && (p.N % 32 != 0 || (p.tuning & MRAG_GEMM_TUNE_GEGLU_NO_STAGED)))) p.staged = 0;
  if (epi == MRAG_EPI_QKNORM_ROPE && !((WM == 2 && WN == 4 && TM == 8 && TN == 4) && p.staged)) return MRAG_ENOTSUP;   // lives in the LDS-staged epilogue
  const size_t lds_stages = 2 * (BM + BN) * 64 * 2;
  size_t lds = (WM == 2 && WN == 4 && TM == 8 && TN == 4 && lds_stages < 8 * 128 * 144) ? 8 * 128 * 144 : lds_stages;
  if constexpr (CONV != 0 && WM == 2 && WN == 4 && TM == 8 && TN == 4) {   // the SLIM form's eight parked words per lane (gemm_tile: cv_park)
    p.cv_lds = (int)lds;
    lds += (size_t)WM * WN * 64 * 32;
  }
  bool sk_ok = false;
  if constexpr (WM == 2 && WN == 4 && TM == 8 && TN == 4 && CONV == 0) sk_ok = sk != nullptr;
  if (sk && !sk_ok) return MRAG_ENOTSUP;
#define MRAG_GEMM_CASE(E)                                                                              \
  case E: {                                                                                            \
    auto kfn = gemm_bf16_kernel<WM, WN, TM, TN, E, CONV>;                                              \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                   \
    break;                                                                                             \
  }
  if constexpr (CONV != 0) {   // convolutions carry bias / residual only
    switch (epi) {
      MRAG_GEMM_CASE(MRAG_EPI_NONE)
      MRAG_GEMM_CASE(MRAG_EPI_RESID)
      default: return MRAG_EINVAL;
    }
  } else {
    switch (epi) {
      MRAG_GEMM_CASE(MRAG_EPI_NONE)
      MRAG_GEMM_CASE(MRAG_EPI_GELU_TANH)
      MRAG_GEMM_CASE(MRAG_EPI_GELU_ERF)
      MRAG_GEMM_CASE(MRAG_EPI_RESID)
      MRAG_GEMM_CASE(MRAG_EPI_GATE_RESID)
      MRAG_GEMM_CASE(MRAG_EPI_SILU)
      MRAG_GEMM_CASE(MRAG_EPI_GEGLU)
      MRAG_GEMM_CASE(EPI_GEGLU_TANH)
      MRAG_GEMM_CASE(MRAG_EPI_QKNORM_ROPE)
      default: return MRAG_EINVAL;
    }
  }
#undef MRAG_GEMM_CASE
  MRAG_LAUNCH_CHECK();
  {
    constexpr int tile = (BM == 256 && BN == 320) ? 1 : (BM == 256 && BN == 128) ? 2 : (BM == 128 && BN == 128) ? 3 : (BM == 192) ? 4 : 0;   // 0: 256x256 (8 or 16 waves)
    constexpr int ids[3][5] = {{MRAG_K_GEMM_256x256, MRAG_K_GEMM_256x320, MRAG_K_GEMM_256x128, MRAG_K_GEMM_128x128, MRAG_K_GEMM_192x256},
                               {MRAG_K_CONV3_256x256, MRAG_K_CONV3_256x320, MRAG_K_CONV3_256x128, MRAG_K_CONV3_128x128, MRAG_K_CONV3_192x256},
                               {MRAG_K_CONVT_256x256, MRAG_K_CONVT_256x320, MRAG_K_CONVT_256x128, MRAG_K_CONVT_128x128, MRAG_K_CONVT_192x256}};
    MRAG_COUNT(ids[CONV][tile]);
  }
  if constexpr (WM == 2 && WN == 4 && TM == 8 && TN == 4 && CONV == 0) {
    if (sk) {   // the partial last round.  Its own launch: the main kernel keeps its register allocation, and whole rounds end together anyway
      const dim3 tgrid(2 * sk->units);
#define MRAG_GEMM_SK_CASE(E)                                                                           \
  case E: {                                                                                            \
    auto kfn = gemm_bf16_kernel<WM, WN, TM, TN, E, 0, true>;                                           \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 16)); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, tgrid, block, lds + 16, s, p);                                                    \
    break;                                                                                             \
  }
      switch (epi) {
        MRAG_GEMM_SK_CASE(MRAG_EPI_NONE)
        MRAG_GEMM_SK_CASE(MRAG_EPI_GELU_TANH)
        MRAG_GEMM_SK_CASE(MRAG_EPI_RESID)
        MRAG_GEMM_SK_CASE(MRAG_EPI_GATE_RESID)
        MRAG_GEMM_SK_CASE(MRAG_EPI_QKNORM_ROPE)
        default: return MRAG_EINVAL;   // mrag_gemm_bf16 plans a tail for these five only
      }
#undef MRAG_GEMM_SK_CASE
      MRAG_LAUNCH_CHECK();
      MRAG_COUNT(MRAG_K_GEMM_STREAMK_TAIL);
    }
  }
  return MRAG_OK;
}

}  // namespace

extern "C" int mrag_gemm_bf16(void* stream, const mrag_gemm_args* a) {
  if (!a || !a->A || !a->W || !a->C) return MRAG_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MRAG_EINVAL;
  if (a->K % 64 != 0 || a->N % 4 != 0) return MRAG_ENOTSUP;
  if (a->epilogue == MRAG_EPI_GEGLU && a->N % 32 != 0) return MRAG_ENOTSUP;
  if (a->lda % 8 != 0 || a->ldw % 8 != 0 || a->ldc % 4 != 0) return MRAG_EINVAL;
  if (((uintptr_t)a->A | (uintptr_t)a->W) & 15) return MRAG_EINVAL;
  if ((uintptr_t)a->C & 7) return MRAG_EINVAL;
  if ((a->epilogue == MRAG_EPI_RESID || a->epilogue == MRAG_EPI_GATE_RESID) &&
      (!a->resid || a->ldr % 4 != 0 || ((uintptr_t)a->resid & 7)))
    return MRAG_EINVAL;
  if (a->epilogue == MRAG_EPI_GATE_RESID &&
      (!a->gate0 || !a->gate1 || a->rows_per_batch <= 0 || a->gate_stride % 4 != 0)) return MRAG_EINVAL;
  GemmP p{};
  p.A = (const bf16_t*)a->A; p.W = (const bf16_t*)a->W; p.bias = (const bf16_t*)a->bias;
  p.C = (bf16_t*)a->C; p.resid = (const bf16_t*)a->resid;
  p.gate0 = (const bf16_t*)a->gate0; p.gate1 = (const bf16_t*)a->gate1;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
  p.rows_per_batch = a->rows_per_batch; p.split = a->split; p.gate_stride = a->gate_stride;
  p.acc_scale = a->acc_scale == 0.0f ? 1.0f : a->acc_scale;
  if (a->w_batch_stride < 0 || a->w_batch_stride % 8 != 0) return MRAG_EINVAL;
  if (a->a_ln != 0 && a->a_ln != 1) return MRAG_EINVAL;
  if (a->a_ln) {
    if ((((uintptr_t)a->a_ln_gamma | (uintptr_t)a->a_ln_beta) & 15) || a->K > 8192) return MRAG_EINVAL;
    p.lna = 1; p.lna_g = (const bf16_t*)a->a_ln_gamma; p.lna_b = (const bf16_t*)a->a_ln_beta; p.lna_eps = a->a_ln_eps;
  }
  p.w_bstride = a->w_batch_stride;
  if (a->epilogue == MRAG_EPI_GEGLU && a->geglu_act != 0 && a->geglu_act != 1) return MRAG_EINVAL;
  if (a->epilogue == MRAG_EPI_QKNORM_ROPE) {
    if (a->qk_dmodel <= 0 || a->qk_dmodel % 64 != 0 || a->N % a->qk_dmodel != 0 || a->qk_first < 0 || a->qk_first + a->N / a->qk_dmodel > 3 ||
        a->rows_per_batch <= 0) return MRAG_EINVAL;
    if ((a->rope_cos != nullptr) != (a->rope_sin != nullptr) || (((uintptr_t)a->rope_cos | (uintptr_t)a->rope_sin) & 15)) return MRAG_EINVAL;
    if (((uintptr_t)a->q_gamma | (uintptr_t)a->q_beta | (uintptr_t)a->k_gamma | (uintptr_t)a->k_beta) & 15) return MRAG_EINVAL;
    p.qg = (const bf16_t*)a->q_gamma; p.qb = (const bf16_t*)a->q_beta; p.kg = (const bf16_t*)a->k_gamma; p.kb = (const bf16_t*)a->k_beta;
    p.rcos = a->rope_cos; p.rsin = a->rope_sin; p.qk_D = a->qk_dmodel; p.rope_text_len = a->rope_text_len; p.qk_first = a->qk_first; p.qk_eps = a->qk_eps; p.q_premul = a->q_premul;
  }
  hipStream_t s = (hipStream_t)stream;
  // big problems: 256x256 tiles, 8 waves (1 workgroup per CU); small ones: 128x128, 4 waves,
  // so that a few hundred rows still spread over the 256 CUs.
  const long long t256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
  p.tuning = a->tuning;
  const int epi = (a->epilogue == MRAG_EPI_GEGLU && a->geglu_act == 1) ? EPI_GEGLU_TANH : a->epilogue;   // the tanh gate is its own instantiation
  if (const int cfg = (a->tuning >> 4) & 0xf) {   // developer knob (tools/microbench.py); 0 = the shipped choice below
    if (cfg == 1 && t256 >= 192) return launch_cfg<4, 4, 4, 4>(s, p, epi);   // 256x256, 16 waves (4 per SIMD)
    if (cfg == 2) return launch_cfg<2, 2, 4, 4>(s, p, epi);                  // 128x128, 4 waves, 2 workgroups per CU
    if (cfg == 4) {                                                          // the persistent kernel whatever the tile count (small-M experiments: T5 11.3 -> 13.6 ms; 256x128 8-wave tiles: 12.9 ms -- the 128x128 tile stays)
      const int rc = launch_w4(s, p, epi);
      if (rc != MRAG_ENOTSUP) return rc;
    }
    if (cfg == 3 && t256 >= 192) {                                           // 256x256, 4 waves, persistent, hand-scheduled
      const int rc = launch_w4(s, p, epi);
      if (rc != MRAG_ENOTSUP) return rc;
    }
  }
  // problems made of whole 128-column wave tiles: the persistent four-wave kernel -- 3-13 % ahead of the 8-wave 256x256 tile on the DiT's shapes, 8-27 %
  // on the UNets' N = 640 / 1280 linears (where it also replaces the 256x320 tile); behind the 8-wave tile where the epilogue of one wave per SIMD outweighs
  // a short K loop (GELU below K = 1536, anything below K = 320), and on shapes that would take its general epilogue path (profiles/r3_gemm_w4_ab.txt)
  if (a->w_batch_stride != 0) {
    // per-sample weights (the motion branch's folded score GEMM): the persistent four-wave kernel only -- whole 128-column wave tiles, aligned rows, no
    // epilogue; anything else is the caller's loop over the samples
    if (epi != MRAG_EPI_NONE || a->N % 128 != 0 || a->K < 320 || a->rows_per_batch <= 0 || a->M % a->rows_per_batch != 0 || a->ldc % 8 != 0 || (((uintptr_t)a->C) & 15) ||
        (a->tuning & (MRAG_GEMM_TUNE_NO_W4 | MRAG_GEMM_TUNE_NO_STAGED)))
      return MRAG_ENOTSUP;
    return launch_w4(s, p, epi);
  }
  if (a->a_ln) return skinny_applies(a, epi) ? launch_skinny(s, p, epi) : MRAG_ENOTSUP;   // the LayerNorm-in-the-A-load form lives in the few-row kernel only: the caller runs LayerNorm + GEMM otherwise
  if (skinny_applies(a, epi)) return launch_skinny(s, p, epi);       // M <= 256: eight waves split K, no LDS ring (gemm_skinny_kernel)
  if (k320_applies(a, epi)) return launch_k320(s, p, epi);           // K = 320, N = 320 .. 2 560: the weight in registers, activations streamed (gemm_k320_kernel)
  const bool w4_ok = t256 >= 192 && !(a->tuning & (MRAG_GEMM_TUNE_NO_W4 | MRAG_GEMM_TUNE_NO_STAGED | MRAG_GEMM_TUNE_STREAMK | MRAG_GEMM_TUNE_NO_WIDE)) && a->N % 128 == 0 &&
      a->K >= (epi == MRAG_EPI_GELU_TANH ? 1536 : 320) &&
      (epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_TANH || epi == MRAG_EPI_RESID || epi == MRAG_EPI_GATE_RESID || epi == MRAG_EPI_QKNORM_ROPE || epi == MRAG_EPI_GEGLU ||
       epi == EPI_GEGLU_TANH) &&
      a->ldc % 8 == 0 && (((uintptr_t)a->C) & 15) == 0 && (!a->resid || (a->ldr % 8 == 0 && (((uintptr_t)a->resid) & 15) == 0));
  const bool w4_tail = w4_ok && t256 >= 2 * SK_CUS && t256 % SK_CUS != 0 && t256 % SK_CUS <= W4_TAIL_MAX && (a->tuning & MRAG_GEMM_TUNE_TAIL_RECT) &&
      (epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_TANH || epi == MRAG_EPI_RESID);                    // (plan_tail_rect's conditions)
  // (first: a problem that the 320-wide tile finishes in fewer rounds -- see wide_rounds_pay; the persistent kernel walks the same 256x256 tile grid)
  if (t256 >= 192 && !wide_n_pays(a->N, a->tuning) && wide_rounds_pay(a->M, a->N, a->tuning, w4_tail) && !(a->tuning & MRAG_GEMM_TUNE_STREAMK) && a->epilogue != MRAG_EPI_GEGLU &&
      a->epilogue != MRAG_EPI_QKNORM_ROPE)
    return launch_cfg<2, 4, 8, 5>(s, p, epi);
  if (w4_ok) {
    const int rc = launch_w4(s, p, epi);
    if (rc != MRAG_ENOTSUP) return rc;
  }
  if (t256 >= 192 && wide_n_pays(a->N, a->tuning) && a->epilogue != MRAG_EPI_GEGLU) return launch_cfg<2, 4, 8, 5>(s, p, epi);   // 256x320 tile
  if (t256 >= 192 && a->workspace && (a->tuning & MRAG_GEMM_TUNE_STREAMK) &&
      (epi == MRAG_EPI_NONE || epi == MRAG_EPI_GELU_TANH || epi == MRAG_EPI_RESID || epi == MRAG_EPI_GATE_RESID || epi == MRAG_EPI_QKNORM_ROPE)) {
    const SkPlan pl = plan_streamk(a->M, a->N, a->K);
    if (pl.use && a->workspace_bytes >= (int64_t)pl.bytes) {
      if ((uintptr_t)a->workspace & 15) return MRAG_EINVAL;
      p.sk_ticket = (unsigned*)a->workspace;
      p.sk_part = (float*)((char*)a->workspace + SK_TICKET_BYTES);
      p.sk_main = pl.n_main; p.sk_rem = pl.rem; p.sk_units = pl.units; p.sk_maxparts = pl.maxparts;
      const hipError_t e = hipMemsetAsync(a->workspace, 0, SK_TICKET_BYTES, s);   // the tickets start at zero whatever an earlier (aborted) launch left
      if (e != hipSuccess) return (int)e;
      const int rc = launch_cfg<2, 4, 8, 4>(s, p, epi, &pl);
      if (rc != MRAG_ENOTSUP) return rc;
    }
  }
  if (t256 >= 192) return launch_cfg<2, 4, 8, 4>(s, p, epi);
  return launch_cfg<2, 2, 4, 4>(s, p, epi);
}

extern "C" int64_t mrag_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 64 != 0) return 0;
  if (wide_n_pays(N)) return 0;
  const SkPlan pl = plan_streamk(M, N, K);
  return pl.use ? (int64_t)pl.bytes : 0;
}

extern "C" int mrag_conv_bf16(void* stream, const mrag_conv_args* a) {
  if (!a || !a->x || !a->W || !a->y) return MRAG_EINVAL;
  if (a->N <= 0 || a->H <= 0 || a->Wd <= 0 || a->Cin <= 0 || a->Cout <= 0) return MRAG_EINVAL;
  if (a->Cin % 64 != 0 || a->Cout % 4 != 0) return MRAG_ENOTSUP;   // one K-tile = 64 channels of one tap
  if (a->mode != MRAG_CONV_3X3 && a->mode != MRAG_CONV_T3) return MRAG_EINVAL;
  if (a->epilogue != MRAG_EPI_NONE && a->epilogue != MRAG_EPI_RESID) return MRAG_EINVAL;
  if (((uintptr_t)a->x | (uintptr_t)a->W) & 15) return MRAG_EINVAL;
  if ((uintptr_t)a->y & 7) return MRAG_EINVAL;
  if (a->epilogue == MRAG_EPI_RESID && (!a->resid || ((uintptr_t)a->resid & 7))) return MRAG_EINVAL;
  GemmP p{};
  p.A = (const bf16_t*)a->x; p.W = (const bf16_t*)a->W; p.bias = (const bf16_t*)a->bias; p.C = (bf16_t*)a->y; p.resid = (const bf16_t*)a->resid;
  p.N = a->Cout; p.ldc = a->Cout; p.ldr = a->Cout; p.cv_C = a->Cin; p.cv_ctiles = a->Cin / 64;
  p.acc_scale = a->acc_scale == 0.0f ? 1.0f : a->acc_scale;
  hipStream_t s = (hipStream_t)stream;
  // the implicit GEMM walks its sources with 32-bit offsets (gemm_tile): positions in 16-byte units relative to the first sample a workgroup touches
  // (at most a few frames apart), weight rows in bytes relative to W
  if ((long long)a->H * a->Wd * (a->Cin / 8) * 6 >= (1LL << 31) || (long long)a->Cout * 27 * a->Cin * 2 >= (1LL << 32)) return MRAG_ENOTSUP;
  if (a->mode == MRAG_CONV_3X3) {
    if ((a->stride != 1 && a->stride != 2) || (a->upsample != 0 && a->upsample != 1)) return MRAG_EINVAL;
    p.cv_H = a->H; p.cv_W = a->Wd; p.cv_up = a->upsample; p.cv_stride = a->stride;
    p.cv_Hi = a->upsample ? 2 * a->H : a->H; p.cv_Wi = a->upsample ? 2 * a->Wd : a->Wd;
    if (a->asym_pad != 0 && (a->asym_pad != 1 || a->stride != 2 || a->upsample)) return MRAG_EINVAL;
    p.cv_pad = a->asym_pad ? 0 : 1;
    // padding 1 / 1: Ho = (Hi + 2 - 3) / stride + 1; padding 0 / 1: Ho = (Hi + 1 - 3) / stride + 1
    p.cv_Ho = (p.cv_Hi + p.cv_pad - 2) / a->stride + 1; p.cv_Wo = (p.cv_Wi + p.cv_pad - 2) / a->stride + 1;
    p.M = (long long)a->N * p.cv_Ho * p.cv_Wo; p.K = 9LL * a->Cin; p.ldw = p.K;
    if (a->t_taps != 0) {   // causal 3x3x3 over frame stacks that already hold the two leading context frames
      if (a->t_taps != 3 || a->t_frames <= 0 || a->N % a->t_frames != 0 || a->stride != 1 || a->upsample || a->asym_pad) return MRAG_EINVAL;
      p.cv_tf = a->t_frames; p.cv_fs = (long long)a->H * a->Wd * a->Cin; p.K = 27LL * a->Cin; p.ldw = p.K;
    }
    const long long t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (t256 >= 192 && (wide_n_pays(p.N) || wide_rounds_pay(p.M, p.N))) return launch_cfg<2, 4, 8, 5, 1>(s, p, a->epilogue);
    if (t256 >= 192 && narrow_n_pays(p.N)) return launch_cfg<4, 2, 4, 4, 1>(s, p, a->epilogue);   // 256x128 tile, 8 waves of 64x64
    if (t256 >= 192 && short_rows_pay(p.M, p.N)) return launch_cfg<2, 4, 6, 4, 1>(s, p, a->epilogue);   // 192x256 tile
    if (t256 >= 192) return launch_cfg<2, 4, 8, 4, 1>(s, p, a->epilogue);
    return launch_cfg<2, 2, 4, 4, 1>(s, p, a->epilogue);
  }
  // (3,1,1) temporal convolution over x [(N = B) x (H = T), Wd = HW, Cin]
  p.cv_T = a->H; p.cv_HW = a->Wd;
  p.M = (long long)a->N * a->H * a->Wd; p.K = 3LL * a->Cin; p.ldw = p.K;
  const long long t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  if (t256 >= 192 && (wide_n_pays(p.N) || wide_rounds_pay(p.M, p.N))) return launch_cfg<2, 4, 8, 5, 2>(s, p, a->epilogue);
  if (t256 >= 192 && short_rows_pay(p.M, p.N)) return launch_cfg<2, 4, 6, 4, 2>(s, p, a->epilogue);      // 192x256 tile
  if (t256 >= 192) return launch_cfg<2, 4, 8, 4, 2>(s, p, a->epilogue);
  return launch_cfg<2, 2, 4, 4, 2>(s, p, a->epilogue);
}
