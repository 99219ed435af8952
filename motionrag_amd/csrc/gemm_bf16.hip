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
#include "common.h"
#include "../../include/mrag_hip.h"

namespace {

struct GemmP {
  const bf16_t* A; const bf16_t* W; const bf16_t* bias; bf16_t* C; const bf16_t* resid;
  const bf16_t* gate0; const bf16_t* gate1;
  long long M, N, K, lda, ldw, ldc, ldr, rows_per_batch, split, gate_stride;
  int tiles_m, tiles_n;
};

template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
  if constexpr (EPI == MRAG_EPI_GELU_TANH) return gelu_tanh_f(v);
  else if constexpr (EPI == MRAG_EPI_GELU_ERF) return gelu_erf_f(v);
  else if constexpr (EPI == MRAG_EPI_SILU) return silu_f(v);
  else return v;
}

template <int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(const GemmP p) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 64;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int PIECES = (BM + BN) / 8;   // 1 KiB LDS-DMA pieces per stage (8 rows x 128 B)
  constexpr int PPW = PIECES / NW;        // pieces per wave
  static_assert(PIECES % NW == 0, "piece split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int nwg = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int tile_m = wg / p.tiles_n, tile_n = wg % p.tiles_n;
  const long long bm0 = (long long)tile_m * BM, bn0 = (long long)tile_n * BN;

  // ---- per-lane DMA source pointers (k = 0), one per piece this wave stages
  const bf16_t* gsrc[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = wave + i * NW;  // pieces [0, BM/8) are A rows, the rest W rows
    const int r = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (lane >> 3);  // source-side swizzle: row&7 == lane>>3
    if (piece < BM / 8) {
      long long row = bm0 + r;
      row = row < p.M ? row : p.M - 1;  // clamp: tail rows re-read a valid row, stores are masked
      gsrc[i] = p.A + row * p.lda + chunk * 8;
    } else {
      long long row = bn0 + (r - BM);
      row = row < p.N ? row : p.N - 1;
      gsrc[i] = p.W + row * p.ldw + chunk * 8;
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

  const int nk = (int)(p.K / BK);

  auto issue = [&](int stage, int kt) {
    char* base = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave + i * NW;
      glds16(gsrc[i] + (long long)kt * BK, base + piece * 1024);  // wave-uniform base (+ lane*16 by HW)
    }
  };

  issue(0, 0);
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

  // ---- epilogue: lane owns row m = .. + (lane & 15), columns n0 + (lane >> 4) * 4 + {0..3}
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long m = bm0 + wm * TM * 16 + i * 16 + frag_row;
    if (m >= p.M) continue;
    const bf16_t* gate = nullptr;
    if constexpr (EPI == MRAG_EPI_GATE_RESID) {
      const long long b = m / p.rows_per_batch, pos = m - b * p.rows_per_batch;
      gate = (pos < p.split ? p.gate0 : p.gate1) + b * p.gate_stride;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long long n = bn0 + wn * TN * 16 + j * 16 + frag_q * 4;
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
      if constexpr (EPI == MRAG_EPI_GATE_RESID || EPI == MRAG_EPI_RESID) {
        const u32x2 rr = *(const u32x2*)(p.resid + m * p.ldr + n);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 out;
      out[0] = pack_bf2(v[0], v[1]);
      out[1] = pack_bf2(v[2], v[3]);
      *(u32x2*)(p.C + m * p.ldc + n) = out;
    }
  }
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(hipStream_t s, const GemmP& p0, int epi) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  GemmP p = p0;
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (int)((p.N + BN - 1) / BN);
  const dim3 grid(p.tiles_m * p.tiles_n), block(WM * WN * 64);
  const size_t lds = 2 * (BM + BN) * 64 * 2;
#define MRAG_GEMM_CASE(E)                                                                              \
  case E: {                                                                                            \
    auto kfn = gemm_bf16_kernel<WM, WN, TM, TN, E>;                                                    \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                   \
    break;                                                                                             \
  }
  switch (epi) {
    MRAG_GEMM_CASE(MRAG_EPI_NONE)
    MRAG_GEMM_CASE(MRAG_EPI_GELU_TANH)
    MRAG_GEMM_CASE(MRAG_EPI_GELU_ERF)
    MRAG_GEMM_CASE(MRAG_EPI_RESID)
    MRAG_GEMM_CASE(MRAG_EPI_GATE_RESID)
    MRAG_GEMM_CASE(MRAG_EPI_SILU)
    default: return MRAG_EINVAL;
  }
#undef MRAG_GEMM_CASE
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

}  // namespace

extern "C" int mrag_gemm_bf16(void* stream, const mrag_gemm_args* a) {
  if (!a || !a->A || !a->W || !a->C) return MRAG_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MRAG_EINVAL;
  if (a->K % 64 != 0 || a->N % 4 != 0) return MRAG_ENOTSUP;
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
  hipStream_t s = (hipStream_t)stream;
  // big problems: 256x256 tiles, 8 waves (1 workgroup per CU); small ones: 128x128, 4 waves,
  // so that a few hundred rows still spread over the 256 CUs.
  const long long t256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
  if (t256 >= 192) return launch_cfg<2, 4, 8, 4>(s, p, a->epilogue);
  return launch_cfg<2, 2, 4, 4>(s, p, a->epilogue);
}
