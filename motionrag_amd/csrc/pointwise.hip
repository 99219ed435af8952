// pointwise.hip -- HBM-bound elementwise kernels of the denoising loop (gfx950).
// All of them stream 16-byte bf16 vectors per lane (grid-stride over <= 2048 workgroups).
#include "common.h"
#include "../../include/mrag_hip.h"

namespace {

constexpr int kMaxBlocks = 2048;

__device__ __forceinline__ void unpack8(const u32x4 r, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(r[i] << 16);
    f[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return r;
}

inline unsigned grid_for(long long work_items) {
  long long b = (work_items + 255) / 256;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0) == lvdm timestep_embedding:
// out[b, i] = cos(t_b * f_i), out[b, half + i] = sin(t_b * f_i), f_i = exp(-ln(1e4) * i / half)
__global__ void timestep_embedding_kernel(const float* t, bf16_t* out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  const float f = expf(-9.210340371976184f * (float)j / (float)half);
  const float arg = t[b] * f;
  out[(long long)b * dim + j] = f2bf(cosf(arg));
  out[(long long)b * dim + half + j] = f2bf(sinf(arg));
}

__global__ void silu_kernel(const bf16_t* x, bf16_t* y, long long n8, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float v[8]; unpack8(*(const u32x4*)(x + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    *(u32x4*)(y + i * 8) = pack8(v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const long long i = n8 * 8 + threadIdx.x;
    y[i] = f2bf(silu_f(bf2f(x[i])));
  }
}

__global__ void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long long n8, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    unpack8(*(const u32x4*)(a + i * 8), u);
    unpack8(*(const u32x4*)(b + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += v[e];
    *(u32x4*)(y + i * 8) = pack8(u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const long long i = n8 * 8 + threadIdx.x;
    y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
  }
}

// y[r, :] = x[r, :] + table[r % period, :]
__global__ void add_rows_kernel(const bf16_t* x, const bf16_t* table, bf16_t* y, long long rows, long long D8, long long period) {
  const long long total = rows * D8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / D8, c = i - r * D8;
    float u[8], v[8];
    unpack8(*(const u32x4*)(x + i * 8), u);
    unpack8(*(const u32x4*)(table + ((r % period) * D8 + c) * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += v[e];
    *(u32x4*)(y + i * 8) = pack8(u);
  }
}

// dst[((b*F+f)*Hp+ph)*Wp+pw, c*4+dy*2+dx] = src[b % Bl, f, c, 2ph+dy, 2pw+dx]; src = cat(src0, src1) on c.
// one thread writes the 4 values (dy,dx) of one (row, c): 8 contiguous bytes.
__global__ void patchify_kernel(const bf16_t* s0, const bf16_t* s1, bf16_t* dst, int B, int Bl, int F, int C0, int C1, int H, int W) {
  const int Hp = H / 2, Wp = W / 2, C = C0 + C1;
  const long long total = (long long)B * F * Hp * Wp * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int pw = (int)(r % Wp); long long r2 = r / Wp;
    const int ph = (int)(r2 % Hp); r2 /= Hp;
    const int f = (int)(r2 % F);
    const int b = (int)(r2 / F);
    const int bl = b % Bl;
    const bf16_t* src; int cc, CC;
    if (c < C0) { src = s0; cc = c; CC = C0; } else { src = s1; cc = c - C0; CC = C1; }
    const bf16_t* sp = src + (((long long)(bl * F + f) * CC + cc) * H + 2 * ph) * W + 2 * pw;
    const unsigned top = *(const unsigned*)sp;        // (dy=0, dx=0..1)
    const unsigned bot = *(const unsigned*)(sp + W);  // (dy=1, dx=0..1)
    u32x2 o; o[0] = top; o[1] = bot;
    *(u32x2*)(dst + r * (C * 4) + c * 4) = o;
  }
}

// dst[b, f, c, 2ph+dy, 2pw+dx] = src[b, (f*Hp+ph)*Wp+pw, c*4+dy*2+dx]
__global__ void unpatchify_kernel(const bf16_t* src, bf16_t* dst, int B, int F, int C, int H, int W) {
  const int Hp = H / 2, Wp = W / 2;
  const long long total = (long long)B * F * C * Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % Wp); long long r = i / Wp;
    const int ph = (int)(r % Hp); r /= Hp;
    const int c = (int)(r % C); r /= C;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    const long long srow = ((long long)(b * F + f) * Hp + ph) * Wp + pw;
    const u32x2 v = *(const u32x2*)(src + srow * (C * 4) + c * 4);
    bf16_t* dp = dst + (((long long)(b * F + f) * C + c) * H + 2 * ph) * W + 2 * pw;
    *(unsigned*)dp = v[0];
    *(unsigned*)(dp + W) = v[1];
  }
}

// v = v_u + g (v_c - v_u); x0 = sa x - sb v; x <- a x + b x0      (CogVideoX DDIM, v-prediction)
__global__ void cfg_ddim_kernel(const bf16_t* vp, bf16_t* x, long long n8, long long n, float g, float sa, float sb, float a, float b) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float vu[8], vc[8], xx[8];
    unpack8(*(const u32x4*)(vp + i * 8), vu);
    unpack8(*(const u32x4*)(vp + n + i * 8), vc);
    unpack8(*(const u32x4*)(x + i * 8), xx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = vu[e] + g * (vc[e] - vu[e]);
      const float x0 = sa * xx[e] - sb * v;
      xx[e] = a * xx[e] + b * x0;
    }
    *(u32x4*)(x + i * 8) = pack8(xx);
  }
}

// v = v_u + g (v_c - v_u); x0 = sa x - sb v; d = second ? m3 x0 - m4 x0_old : x0; x <- m1 x - m2 d + mn noise; x0_old <- x0
// (CogVideoXDPMScheduler.step: the SDE form of DPM-Solver++(2M) the reference's shipped CogVideoX config samples with)
__global__ void cfg_dpm_kernel(const bf16_t* vp, bf16_t* x, bf16_t* x0_old, const bf16_t* noise, long long n8, long long n, float g, float sa, float sb, float m1,
                               float m2, float m3, float m4, float mn, int second) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float vu[8], vc[8], xx[8], xo[8], nz[8];
    unpack8(*(const u32x4*)(vp + i * 8), vu);
    unpack8(*(const u32x4*)(vp + n + i * 8), vc);
    unpack8(*(const u32x4*)(x + i * 8), xx);
    unpack8(*(const u32x4*)(noise + i * 8), nz);
    if (second) unpack8(*(const u32x4*)(x0_old + i * 8), xo);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = vu[e] + g * (vc[e] - vu[e]);
      const float x0 = bf_round(sa * xx[e] - sb * v);            // the reference carries pred_original_sample between steps in the latents' dtype
      const float d = second ? m3 * x0 - m4 * xo[e] : x0;
      xx[e] = m1 * xx[e] - m2 * d + mn * nz[e];
      xo[e] = x0;
    }
    *(u32x4*)(x + i * 8) = pack8(xx);
    *(u32x4*)(x0_old + i * 8) = pack8(xo);
  }
}

// y[r, :] = x[r, :] + table[(r / div) % period, :]  (per-frame / per-sample vectors broadcast over the pixels of a frame)
__global__ void add_bcast_kernel(const bf16_t* x, const bf16_t* table, bf16_t* y, long long rows, long long D8, long long div, long long period, long long tstride8) {
  const long long total = rows * D8;
  // (row, column vector) advance incrementally: one 64-bit division per thread instead of three per 16-byte vector
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long rstep = stride / D8, cstep = stride - rstep * D8;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long r = i / D8, c = i - r * D8;
  long long bidx = (r / div) % period, brem = r % div;       // table row of r and r's offset inside its group of `div` rows
  for (; i < total; i += stride) {
    float u[8], v[8];
    unpack8(*(const u32x4*)(x + i * 8), u);
    unpack8(*(const u32x4*)(table + (bidx * tstride8 + c) * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += v[e];
    *(u32x4*)(y + i * 8) = pack8(u);
    long long dr = rstep;
    c += cstep;
    if (c >= D8) { c -= D8; ++dr; }
    brem += dr;
    if (brem >= div) {                                       // crossing into another group: divide only then (always, for div == 1)
      const long long q = brem / div;
      brem -= q * div;
      bidx = (bidx + q) % period;
    }
  }
}

// out = a x + b y in fp32, one rounding (AlphaBlender)
__global__ void axpby_kernel(const bf16_t* x, const bf16_t* y, bf16_t* out, long long n8, float a, float b) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    unpack8(*(const u32x4*)(x + i * 8), u);
    unpack8(*(const u32x4*)(y + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = a * u[e] + b * v[e];
    *(u32x4*)(out + i * 8) = pack8(u);
  }
}

// latents <- cx * latents + cv * (v_u + g[f] (v_c - v_u)), f = (i / frame_elems) % F
__global__ void cfg_euler_kernel(const bf16_t* vp, bf16_t* x, long long n8, long long n, const float* g, int F, long long frame_elems, float cx, float cv) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float vu[8], vc[8], xx[8];
    unpack8(*(const u32x4*)(vp + i * 8), vu);
    unpack8(*(const u32x4*)(vp + n + i * 8), vc);
    unpack8(*(const u32x4*)(x + i * 8), xx);
    const float gf = g[((i * 8) / frame_elems) % F];
#pragma unroll
    for (int e = 0; e < 8; ++e) xx[e] = cx * xx[e] + cv * (vu[e] + gf * (vc[e] - vu[e]));
    *(u32x4*)(x + i * 8) = pack8(xx);
  }
}

// condition_fusion 'mean' / 'weight' (src/projects/condition/utils.py:21-27): out[b, n] = (sum_k w[b, k] x[b, k, n]) / div, fp32 weights and
// accumulation in k order, ONE rounding to bf16 (the weights are never rounded to bf16: 1/9 in bf16 would bias the mean by 0.2 %).
__global__ void weighted_sum_kernel(const bf16_t* x, const float* w, bf16_t* out, int K, long long n8, float div) {
  const int b = blockIdx.y;
  const bf16_t* xb = x + (long long)b * K * n8 * 8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      float v[8]; unpack8(*(const u32x4*)(xb + ((long long)k * n8 + i) * 8), v);
      const float wk = w ? w[(long long)b * K + k] : 1.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(wk, v[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] / div;
    *(u32x4*)(out + ((long long)b * n8 + i) * 8) = pack8(acc);
  }
}

// Row softmax of a materialised score matrix: y[r, :] = softmax(scale * x[r, :]).  The single-head, head_dim-512 attention of the KL-VAE mid block
// (AttnBlock.forward, lvdm/modules/networks/ae_modules.py:54-79: bmm -> * c^-0.5 -> softmax(dim=2) -> bmm) runs as two GEMMs around this pass; at 9 216 keys a
// row is 18 KB, read three times through L2 (max, sum, write) -- HBM sees one read and one write.  One 256-thread workgroup per row, fp32 statistics.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16_t* x, bf16_t* y, long long cols, long long ldx, long long ldy, float scale) {
  __shared__ float red[4];
  const bf16_t* xr = x + (long long)blockIdx.x * ldx;
  bf16_t* yr = y + (long long)blockIdx.x * ldy;
  const int tid = threadIdx.x;
  const long long c8 = cols / 8;
  float m = -INFINITY;
  for (long long i = tid; i < c8; i += 256) {
    float v[8]; unpack8(*(const u32x4*)(xr + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[e]);
  }
  for (long long i = c8 * 8 + tid; i < cols; i += 256) m = fmaxf(m, __uint_as_float(((unsigned)xr[i]) << 16));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  __syncthreads();
  const float k = scale * 1.4426950408889634f, mk = m * 1.4426950408889634f;
  float sum = 0.f;
  for (long long i = tid; i < c8; i += 256) {
    float v[8]; unpack8(*(const u32x4*)(xr + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += exp2f(fmaf(v[e], k, -mk));
  }
  for (long long i = c8 * 8 + tid; i < cols; i += 256) sum += exp2f(fmaf(__uint_as_float(((unsigned)xr[i]) << 16), k, -mk));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (long long i = tid; i < c8; i += 256) {
    float v[8]; unpack8(*(const u32x4*)(xr + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = exp2f(fmaf(v[e], k, -mk)) * inv;
    *(u32x4*)(yr + i * 8) = pack8(v);
  }
  for (long long i = c8 * 8 + tid; i < cols; i += 256) {
    const float e = exp2f(fmaf(__uint_as_float(((unsigned)xr[i]) << 16), k, -mk)) * inv;
    float one[8] = {e, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    yr[i] = (bf16_t)(pack8(one)[0] & 0xffffu);
  }
}

// denormalize (src/utils/pipeline.py:178-184): uint8(clip((x + 1) / 2, 0, 1) * 255) with the reference's rounding points -- on a bf16 tensor every torch op rounds
// to bf16 (x + 1, then / 2 is exact, then * 255), and `.to(torch.uint8)` truncates.  Byte output: bit-exact against the torch ops on the host.
template <bool FP32>
__global__ void denormalize_u8_kernel(const void* x, unsigned char* y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v;
    if (FP32) {
      v = ((const float*)x)[i];
      v = (v + 1.0f) / 2.0f;
      v = fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f;
    } else {
      auto rb = [](float f) { float o[8] = {f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; return __uint_as_float((pack8(o)[0] & 0xffffu) << 16); };   // round to bf16
      v = __uint_as_float(((unsigned)((const bf16_t*)x)[i]) << 16);
      v = rb(rb(v + 1.0f) / 2.0f);
      v = rb(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f);
    }
    y[i] = v != v ? 0 : (unsigned char)v;          // NaN -> 0 (the host's cast of NaN is unspecified; clip keeps everything else in [0, 255])
  }
}

}  // namespace

extern "C" int mrag_timestep_embedding_bf16(void* stream, const float* t, void* out, int32_t B, int32_t dim) {
  if (!t || !out || B <= 0 || dim <= 0 || (dim & 1)) return MRAG_EINVAL;
  const int total = B * (dim / 2);
  MRAG_LAUNCH(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (bf16_t*)out, B, dim);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_silu_bf16(void* stream, const void* x, void* y, int64_t n) {
  if (!x || !y || n <= 0 || (((uintptr_t)x | (uintptr_t)y) & 15)) return MRAG_EINVAL;
  MRAG_LAUNCH(silu_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long long)(n / 8), (long long)n);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_add_bf16(void* stream, const void* a, const void* b, void* y, int64_t n) {
  if (!a || !b || !y || n <= 0 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15)) return MRAG_EINVAL;
  MRAG_LAUNCH(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, (long long)(n / 8), (long long)n);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_add_rows_bf16(void* stream, const void* x, const void* table, void* y, int64_t rows, int64_t D, int64_t period) {
  if (!x || !table || !y || rows <= 0 || D <= 0 || period <= 0 || D % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)x | (uintptr_t)table | (uintptr_t)y) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(add_rows_kernel, dim3(grid_for(rows * D / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)table, (bf16_t*)y, (long long)rows, (long long)(D / 8), (long long)period);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_patchify_bf16(void* stream, const void* src0, const void* src1, void* dst, int32_t B, int32_t Bl, int32_t F,
                                  int32_t C0, int32_t C1, int32_t H, int32_t W) {
  if (!src0 || !dst || B <= 0 || Bl <= 0 || F <= 0 || C0 <= 0 || C1 < 0 || H <= 0 || W <= 0) return MRAG_EINVAL;
  if ((H & 1) || (W & 1) || (C1 > 0 && !src1) || B % Bl != 0) return MRAG_EINVAL;
  if (((uintptr_t)src0 | (uintptr_t)src1) & 3) return MRAG_EINVAL;
  if ((uintptr_t)dst & 7) return MRAG_EINVAL;
  const long long total = (long long)B * F * (H / 2) * (W / 2) * (C0 + C1);
  MRAG_LAUNCH(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src0, (const bf16_t*)src1,
                     (bf16_t*)dst, B, Bl, F, C0, C1, H, W);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_unpatchify_bf16(void* stream, const void* src, void* dst, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W) {
  if (!src || !dst || B <= 0 || F <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return MRAG_EINVAL;
  if (((uintptr_t)src & 7) || ((uintptr_t)dst & 3)) return MRAG_EINVAL;
  const long long total = (long long)B * F * C * (H / 2) * (W / 2);
  MRAG_LAUNCH(unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, B, F, C, H, W);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_cfg_ddim_step_bf16(void* stream, const void* v_pred, void* latents, int64_t n, float guidance, float sqrt_alpha_t,
                                       float sqrt_beta_t, float a_t, float b_t) {
  if (!v_pred || !latents || n <= 0 || n % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)v_pred | (uintptr_t)latents) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(cfg_ddim_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v_pred, (bf16_t*)latents,
                     (long long)(n / 8), (long long)n, guidance, sqrt_alpha_t, sqrt_beta_t, a_t, b_t);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_cfg_dpm_step_bf16(void* stream, const void* v_pred, void* latents, void* x0_prev, const void* noise, int64_t n, float guidance,
                                      float sqrt_alpha_t, float sqrt_beta_t, float m1, float m2, float m3, float m4, float m_noise, int32_t second_order) {
  if (!v_pred || !latents || !x0_prev || !noise || n <= 0 || n % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)v_pred | (uintptr_t)latents | (uintptr_t)x0_prev | (uintptr_t)noise) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(cfg_dpm_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v_pred, (bf16_t*)latents, (bf16_t*)x0_prev,
              (const bf16_t*)noise, (long long)(n / 8), (long long)n, guidance, sqrt_alpha_t, sqrt_beta_t, m1, m2, m3, m4, m_noise, second_order ? 1 : 0);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_add_bcast_bf16(void* stream, const void* x, const void* table, void* y, int64_t rows, int64_t D, int64_t div, int64_t period, int64_t table_stride) {
  if (!x || !table || !y || rows <= 0 || D <= 0 || div <= 0 || period <= 0 || D % 8 != 0) return MRAG_EINVAL;
  if (table_stride == 0) table_stride = D;
  if (table_stride < D || table_stride % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)x | (uintptr_t)table | (uintptr_t)y) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(add_bcast_kernel, dim3(grid_for(rows * D / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)table, (bf16_t*)y,
              (long long)rows, (long long)(D / 8), (long long)div, (long long)period, (long long)(table_stride / 8));
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_axpby_bf16(void* stream, const void* x, const void* y, void* out, int64_t n, float a, float b) {
  if (!x || !y || !out || n <= 0 || n % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(axpby_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)out, (long long)(n / 8), a, b);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_cfg_euler_step_bf16(void* stream, const void* v_pred, void* latents, int64_t n, const float* guidance, int32_t F,
                                        int64_t frame_elems, float c_x, float c_v) {
  if (!v_pred || !latents || !guidance || n <= 0 || n % 8 != 0 || F <= 0 || frame_elems <= 0 || frame_elems % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)v_pred | (uintptr_t)latents) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(cfg_euler_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v_pred, (bf16_t*)latents, (long long)(n / 8),
              (long long)n, guidance, (int)F, (long long)frame_elems, c_x, c_v);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_weighted_sum_bf16(void* stream, const void* x, const float* w, void* out, int32_t B, int32_t K, int64_t n, float div) {
  if (!x || !out || B <= 0 || K <= 0 || n <= 0 || n % 8 != 0 || div == 0.f) return MRAG_EINVAL;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return MRAG_EINVAL;
  const dim3 grid(grid_for(n / 8), (unsigned)B);
  MRAG_LAUNCH(weighted_sum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, (bf16_t*)out, K, n / 8, div);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_softmax_rows_bf16(void* stream, const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float scale) {
  if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols || !(scale > 0.f)) return MRAG_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) || (ldx % 8) || (ldy % 8)) return MRAG_EINVAL;
  if (rows > 0x7fffffffLL) return MRAG_EINVAL;
  MRAG_LAUNCH(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long long)cols, (long long)ldx, (long long)ldy, scale);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

namespace {
// Tile seams of the tiled VAE decode / encode: tile [T, th, tw, C] is blended IN PLACE with the bottom `ev` rows of the tile above (`up`
// [T, uh, tw, C]) and then with the right `eh` columns of the tile to its left (`left` [T, th, lw, C]) -- both already blended themselves,
// which is what the in-place loops of diffusers' blend_v / blend_h produce when the tiles are visited row by row.
__global__ __launch_bounds__(256) void blend_tile_kernel(bf16_t* tile, const bf16_t* up, const bf16_t* left, int T, int th, int tw, int C, int uh, int lw,
                                                         int ev, int eh) {
  const long long total = (long long)T * th * tw * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % tw); r /= tw;
    const int y = (int)(r % th);
    const int t = (int)(r / th);
    const bool bv = up && y < ev, bh = left && x < eh;
    if (!bv && !bh) continue;
    float v = bf2f(tile[i]);
    if (bv) {
      const float w = (float)y / (float)ev;
      v = bf2f(f2bf(bf2f(up[(((long long)t * uh + (uh - ev + y)) * tw + x) * C + c]) * (1.f - w) + v * w));
    }
    if (bh) {
      const float w = (float)x / (float)eh;
      v = bf2f(left[(((long long)t * th + y) * lw + (lw - eh + x)) * C + c]) * (1.f - w) + v * w;
    }
    tile[i] = f2bf(v);
  }
}
}  // namespace

extern "C" int mrag_blend_tile_bf16(void* stream, void* tile, const void* up, const void* left, int32_t T, int32_t th, int32_t tw, int32_t C, int32_t up_h,
                                    int32_t left_w, int32_t extent_v, int32_t extent_h) {
  if (!tile || T <= 0 || th <= 0 || tw <= 0 || C <= 0) return MRAG_EINVAL;
  if ((up && (up_h <= 0 || extent_v <= 0)) || (left && (left_w <= 0 || extent_h <= 0))) return MRAG_EINVAL;
  if (!up && !left) return MRAG_OK;
  int ev = extent_v, eh = extent_h;                       // min(a.shape, b.shape, extent) of blend_v / blend_h
  if (up) { if (ev > up_h) ev = up_h; if (ev > th) ev = th; }
  if (left) { if (eh > left_w) eh = left_w; if (eh > tw) eh = tw; }
  const long long total = (long long)T * th * tw * C, blocks = (total + 255) / 256;
  MRAG_LAUNCH(blend_tile_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)tile, (const bf16_t*)up,
              (const bf16_t*)left, T, th, tw, C, up_h, left_w, ev, eh);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_denormalize_u8(void* stream, const void* x, void* y, int64_t n, int32_t src_fp32) {
  if (!x || !y || n <= 0) return MRAG_EINVAL;
  const long long blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks < 256 * 64 ? blocks : 256 * 64));
  if (src_fp32) MRAG_LAUNCH(denormalize_u8_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)y, (long long)n);
  else MRAG_LAUNCH(denormalize_u8_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)y, (long long)n);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}
