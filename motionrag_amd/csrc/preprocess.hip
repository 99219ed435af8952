// preprocess.hip -- the pixel side of the frozen feature encoders that feed CAMA:
//   VideoMAEEmbedder.forward / preprocess   src/projects/condition/encoders/condition.py:378-400
//       16 uniformly sampled frames, (x + 1) / 2, Resize(224, bilinear, antialias), CenterCrop(224), ImageNet normalise,
//       then VideoMAE's tubelet embedding Conv3d(3, 768, (2, 16, 16), stride (2, 16, 16))
//   DINOImageEmbedder.forward / CLIPImageEmbedder.preprocess   condition.py:503-507, 561-604
//       (x + 1) / 2, Resize(256, bicubic, antialias), CenterCrop(224), normalise, then Conv2d(3, 1024, 14, stride 14)
// ONE kernel does the frame gather, the antialiased separable resize, the crop, the value map and the normalisation and writes the result
// directly as the ROWS of the patch-embedding GEMM ([tokens, (c, dt, dy, dx)] bf16, K zero-padded to the GEMM's 64-deep tiles): no resized
// image, no permuted copy and no im2col buffer ever exist.  HBM-bound: every source pixel is read ~(taps / scale) times through L2, every
// output element written once.
//
// The resize is torch's `interpolate(..., antialias=True, align_corners=False)` (what torchvision's Resize calls on tensors): per output
// coordinate a span [first, first + count) of source coordinates with normalised triangle / cubic (a = -0.5) weights whose support grows with
// the down-scale factor.  The spans and weights are built on the host once per (source size, target size) (`ops.resize_taps`) and cropped to
// the centre window, so the kernel is a plain two-level weighted sum in fp32 -- sum_y wy (sum_x wx src) -- rounded once to bf16 after the
// affine value map; ATen's own GPU kernel rounds three times on the bf16 path ((x+1)/2, resize, normalise), so this is at least as close to
// the fp32 result.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrag_hip.h"

namespace {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

struct PreP {
  const void* src; const int32_t* frame_idx;
  const float* wy; const int32_t* y0; const int32_t* ny;
  const float* wx; const int32_t* x0; const int32_t* nx;
  bf16_t* out;
  long long s_n, s_t, s_c, ldo;
  int N, T, C, H, W, OH, OW, taps_y, taps_x, pt, ph, pw, src_fp32, Kvalid;
  float a[4], b[4];          // out = a[c] * resized + b[c]   (value map and normalisation folded on the host)
};

// one thread = one output pixel of one (image, frame, channel) plane; x fastest so source reads of neighbouring lanes overlap / coalesce and
// the 14- or 16-pixel patch rows leave as contiguous 28- / 32-byte runs
template <bool FP32>
__global__ __launch_bounds__(256) void resize_patch_kernel(const PreP p) {
  const long long total = (long long)p.N * p.T * p.C * p.OH * p.OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % p.OW);
    long long r = i / p.OW;
    const int oy = (int)(r % p.OH); r /= p.OH;
    const int c = (int)(r % p.C); r /= p.C;
    const int t = (int)(r % p.T);
    const int n = (int)(r / p.T);
    const int ts = p.frame_idx ? p.frame_idx[t] : t;
    const long long base = (long long)n * p.s_n + (long long)ts * p.s_t + (long long)c * p.s_c;
    const int ys = p.y0[oy], yc = p.ny[oy], xs = p.x0[ox], xc = p.nx[ox];
    const float* wyp = p.wy + (long long)oy * p.taps_y;
    const float* wxp = p.wx + (long long)ox * p.taps_x;
    float acc = 0.f;
    for (int j = 0; j < yc; ++j) {
      const long long row = base + (long long)(ys + j) * p.W + xs;
      float h = 0.f;
      if (FP32) {
        const float* s = (const float*)p.src + row;
        for (int k = 0; k < xc; ++k) h = fmaf(wxp[k], s[k], h);
      } else {
        const bf16_t* s = (const bf16_t*)p.src + row;
        for (int k = 0; k < xc; ++k) h = fmaf(wxp[k], bf2f(s[k]), h);
      }
      acc = fmaf(wyp[j], h, acc);
    }
    const float v = fmaf(p.a[c], acc, p.b[c]);
    const int Hp = p.OH / p.ph, Wp = p.OW / p.pw;
    const long long tok = (((long long)n * (p.T / p.pt) + t / p.pt) * Hp + oy / p.ph) * Wp + ox / p.pw;
    const int col = ((c * p.pt + t % p.pt) * p.ph + oy % p.ph) * p.pw + ox % p.pw;
    p.out[tok * p.ldo + col] = f2bf(v);
  }
}


// LDS-tiled form of the resize: a workgroup owns 8 x 64 output pixels of one plane.  The source window (R rows x CW columns, a few KB) is loaded ONCE with
// coalesced loads and converted to fp32 in LDS; the horizontal pass runs from LDS into a second LDS image h[R][64]; the vertical pass reads h.  A source pixel is
// read from HBM / L2 once per tile instead of ~25-100 times per output pixel, and the arithmetic drops from taps_x * taps_y to taps_x * R / 8 + taps_y per output.
// Same summation order per output as resize_patch_kernel (sum_y w_y (sum_x w_x src)), hence bit-identical results.
constexpr int RT_TH = 8, RT_TW = 64, RT_RMAX = 48, RT_CWMAX = 320;   // bounds of the window the host accepts: (48 x 321 + 48 x 64) floats = 74 KB at most

template <bool FP32>
__global__ __launch_bounds__(256) void resize_patch_tiled_kernel(const PreP p, int tiles_x, int tiles_y, int win_elems, int win_rows) {
  extern __shared__ float rt_smem[];       // [window: R x pitch | h: R x 64 | wx: 64 x taps_x | wy: 8 x taps_y | x0, nx: 64 + 64 | y0, ny: 8 + 8]
  float* win = rt_smem;
  float* hbuf = win + win_elems;
  float* swx = hbuf + win_rows * RT_TW;
  float* swy = swx + RT_TW * p.taps_x;
  int* sx0 = (int*)(swy + RT_TH * p.taps_y);
  int* snx = sx0 + RT_TW;
  int* sy0 = snx + RT_TW;
  int* sny = sy0 + RT_TH;
  const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6;      // lane -> output column of the tile, wave -> row phase: no divisions in the passes
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; b /= tiles_y;
  const int c = b % p.C; b /= p.C;
  const int t = b % p.T;
  const int n = b / p.T;
  const int oy0 = ty * RT_TH, ox0 = tx * RT_TW;
  const int th = min(RT_TH, p.OH - oy0), tw = min(RT_TW, p.OW - ox0);
  // the tile's tap tables into LDS (the inner loops below never touch global memory for them)
  if (tid < tw) { sx0[tid] = p.x0[ox0 + tid]; snx[tid] = p.nx[ox0 + tid]; }
  if (tid >= 64 && tid < 64 + th) { sy0[tid - 64] = p.y0[oy0 + tid - 64]; sny[tid - 64] = p.ny[oy0 + tid - 64]; }
  for (int i = tid; i < tw * p.taps_x; i += 256) swx[i] = p.wx[(long long)ox0 * p.taps_x + i];
  for (int i = tid; i < th * p.taps_y; i += 256) swy[i] = p.wy[(long long)oy0 * p.taps_y + i];
  // source window of the tile: span starts and span ends are both non-decreasing in the output coordinate (ATen's xmin / xmax), so the first and the
  // last output coordinate of the tile bound it -- four table reads, not a scan
  const int r0 = p.y0[oy0], r1 = p.y0[oy0 + th - 1] + p.ny[oy0 + th - 1];
  const int c0 = p.x0[ox0], c1 = p.x0[ox0 + tw - 1] + p.nx[ox0 + tw - 1];
  const int R = r1 - r0, CW = c1 - c0, pitch = CW | 1;     // odd pitch: the column-strided reads of the horizontal pass spread over the banks
  const int ts = p.frame_idx ? p.frame_idx[t] : t;
  const long long base = (long long)n * p.s_n + (long long)ts * p.s_t + (long long)c * p.s_c + (long long)r0 * p.W + c0;
  for (int r = ly; r < R; r += 4) {                        // coalesced: a wave reads 64 consecutive pixels of one source row
    const long long rowoff = base + (long long)r * p.W;
    for (int cc = lx; cc < CW; cc += 64) win[r * pitch + cc] = FP32 ? ((const float*)p.src)[rowoff + cc] : bf2f(((const bf16_t*)p.src)[rowoff + cc]);
  }
  __syncthreads();
  if (lx < tw) {                                           // horizontal pass: h[r][ox], weights and spans in registers per lane
    const int xs = sx0[lx] - c0, xc = snx[lx];
    const float* wxp = swx + lx * p.taps_x;
    for (int r = ly; r < R; r += 4) {
      const float* srow = win + r * pitch + xs;
      float h = 0.f;
      for (int k = 0; k < xc; ++k) h = fmaf(wxp[k], srow[k], h);
      hbuf[r * RT_TW + lx] = h;
    }
  }
  __syncthreads();
  const int Hp = p.OH / p.ph, Wp = p.OW / p.pw;
  if (lx < tw) {
    for (int oyl = ly; oyl < th; oyl += 4) {               // vertical pass + affine + patch-row store
      const int oy = oy0 + oyl, oxg = ox0 + lx;
      const int ys = sy0[oyl] - r0, yc = sny[oyl];
      const float* wyp = swy + oyl * p.taps_y;
      float acc = 0.f;
      for (int j = 0; j < yc; ++j) acc = fmaf(wyp[j], hbuf[(ys + j) * RT_TW + lx], acc);
      const float v = fmaf(p.a[c], acc, p.b[c]);
      const long long tok = (((long long)n * (p.T / p.pt) + t / p.pt) * Hp + oy / p.ph) * Wp + oxg / p.pw;
      const int col = ((c * p.pt + t % p.pt) * p.ph + oy % p.ph) * p.pw + oxg % p.pw;
      p.out[tok * p.ldo + col] = f2bf(v);
    }
  }
}

// zero the K padding of the patch rows (columns Kvalid .. ldo) -- DINOv2's 3 * 14 * 14 = 588 -> 640
__global__ __launch_bounds__(256) void zero_pad_kernel(bf16_t* out, long long rows, long long ldo, int Kvalid) {
  const int padw = (int)(ldo - Kvalid);
  const long long total = rows * padw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[(i / padw) * ldo + Kvalid + (i % padw)] = 0;
}

// ViT token assembly (transformers Dinov2Embeddings.forward / VideoMAEEmbeddings.forward): out[n, j] = (j < P ? prefix[j] : x[n, j - P]) + pos[j]
// 8 bf16 per thread (D % 8 == 0)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const bf16_t* x, const bf16_t* prefix, const bf16_t* pos, bf16_t* out,
                                                              long long N, int L, int P, int D) {
  const int dv = D / 8;
  const long long total = N * (long long)(L + P) * dv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % dv) * 8;
    const long long r = i / dv;
    const int j = (int)(r % (L + P));
    const long long n = r / (L + P);
    const uint4 a = j < P ? *(const uint4*)(prefix + (long long)j * D + d) : *(const uint4*)(x + (n * L + (j - P)) * D + d);
    uint4 b = make_uint4(0, 0, 0, 0);
    if (pos) b = *(const uint4*)(pos + (long long)j * D + d);
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    unsigned ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = bf2f((bf16_t)(aw[k] & 0xffff)) + bf2f((bf16_t)(bw[k] & 0xffff));
      const float hi = bf2f((bf16_t)(aw[k] >> 16)) + bf2f((bf16_t)(bw[k] >> 16));
      ow[k] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
    }
    *(uint4*)(out + r * D + d) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

}  // namespace

extern "C" int mrag_assemble_tokens_bf16(void* stream, const void* x, const void* prefix, const void* pos, void* out,
                                         int64_t N, int32_t L, int32_t P, int32_t D) {
  if (!x || !out || N <= 0 || L <= 0 || P < 0 || D <= 0 || D % 8 || (P > 0 && !prefix)) return MRAG_EINVAL;
  const long long total = N * (long long)(L + P) * (D / 8);
  const long long blocks = (total + 255) / 256;
  assemble_tokens_kernel<<<(unsigned)(blocks < 256 * 32 ? blocks : 256 * 32), 256, 0, (hipStream_t)stream>>>(
      (const bf16_t*)x, (const bf16_t*)prefix, (const bf16_t*)pos, (bf16_t*)out, N, L, P, D);
  return (int)hipGetLastError();
}

extern "C" int mrag_resize_patchify_bf16(void* stream, const mrag_resize_patch_args* a) {
  if (!a || !a->src || !a->out || !a->wy || !a->wx || !a->y0 || !a->x0 || !a->ny || !a->nx) return MRAG_EINVAL;
  if (a->N <= 0 || a->T <= 0 || a->C <= 0 || a->C > 4 || a->H <= 0 || a->W <= 0 || a->OH <= 0 || a->OW <= 0) return MRAG_EINVAL;
  if (a->pt <= 0 || a->ph <= 0 || a->pw <= 0 || a->T % a->pt || a->OH % a->ph || a->OW % a->pw) return MRAG_EINVAL;
  if (a->taps_y <= 0 || a->taps_x <= 0) return MRAG_EINVAL;
  const int Kvalid = a->C * a->pt * a->ph * a->pw;
  if (a->ldo < Kvalid) return MRAG_EINVAL;
  PreP p;
  p.src = a->src; p.frame_idx = a->frame_idx;
  p.wy = a->wy; p.y0 = a->y0; p.ny = a->ny; p.wx = a->wx; p.x0 = a->x0; p.nx = a->nx;
  p.out = (bf16_t*)a->out;
  p.s_n = a->s_n; p.s_t = a->s_t; p.s_c = a->s_c; p.ldo = a->ldo;
  p.N = a->N; p.T = a->T; p.C = a->C; p.H = a->H; p.W = a->W; p.OH = a->OH; p.OW = a->OW;
  p.taps_y = a->taps_y; p.taps_x = a->taps_x; p.pt = a->pt; p.ph = a->ph; p.pw = a->pw; p.src_fp32 = a->src_fp32; p.Kvalid = Kvalid;
  for (int c = 0; c < 4; ++c) { p.a[c] = c < a->C ? a->scale[c] : 0.f; p.b[c] = c < a->C ? a->shift[c] : 0.f; }
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)a->N * a->T * a->C * a->OH * a->OW;
  const long long rows = (long long)a->N * (a->T / a->pt) * (a->OH / a->ph) * (a->OW / a->pw);
  if (a->ldo > Kvalid) {
    const long long zt = rows * (a->ldo - Kvalid);
    zero_pad_kernel<<<(unsigned)((zt + 255) / 256 < 65536 ? (zt + 255) / 256 : 65536), 256, 0, s>>>(p.out, rows, p.ldo, Kvalid);
  }
  // the tiled kernel when every tile's source window fits its LDS image: the window of 8 (64) output coordinates spans at most 8 (64) * scale + taps source
  // coordinates; extreme down-scales fall back to the per-pixel kernel
  const long long win_r = ((long long)RT_TH * a->H + a->OH - 1) / a->OH + a->taps_y + 2, win_c = ((long long)RT_TW * a->W + a->OW - 1) / a->OW + a->taps_x + 2;
  const int tiles_x = (a->OW + RT_TW - 1) / RT_TW, tiles_y = (a->OH + RT_TH - 1) / RT_TH;
  const long long tiles = (long long)a->N * a->T * a->C * tiles_x * tiles_y;
  if (!a->no_tiling && win_r <= RT_RMAX && win_c <= RT_CWMAX && tiles < 0x7fffffffLL) {
    const int win_elems = (int)(win_r * (win_c | 1));
    const size_t lds = (size_t)(win_elems + win_r * RT_TW + RT_TW * a->taps_x + RT_TH * a->taps_y + 2 * RT_TW + 2 * RT_TH) * sizeof(float);
    if (a->src_fp32) resize_patch_tiled_kernel<true><<<(unsigned)tiles, 256, lds, s>>>(p, tiles_x, tiles_y, win_elems, (int)win_r);
    else resize_patch_tiled_kernel<false><<<(unsigned)tiles, 256, lds, s>>>(p, tiles_x, tiles_y, win_elems, (int)win_r);
    return (int)hipGetLastError();
  }
  const long long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 256 * 64 ? blocks : 256 * 64);
  if (a->src_fp32) resize_patch_kernel<true><<<grid, 256, 0, s>>>(p);
  else resize_patch_kernel<false><<<grid, 256, 0, s>>>(p);
  return (int)hipGetLastError();
}
