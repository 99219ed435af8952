// unet_ops.hip -- HBM-bound kernels of the spatio-temporal UNet denoisers (DynamiCrafter / SVD) for gfx950.
// Activations are channels-last rows [(n, y, x), C] bf16 (n = b*t frames), so every linear / attention / conv of the UNet
// is a row-major GEMM or a row gather in front of one.
//
//   mrag_groupnorm_bf16   nn.GroupNorm(32, C) [+ per-(n, c) embedding pre-add] [+ SiLU]
//                         (lvdm/basics.py:81-88 GroupNorm32; openaimodel3d.py:152-181 ResBlock in/out layers, :258-268
//                          TemporalConvBlock; attention.py:286,357 transformer norms)
//   mrag_im2col3x3_bf16   row gather for nn.Conv2d 3x3 (stride 1 / 2, pad 1; optional nearest x2 upsample in front:
//                         openaimodel3d.py:52-107 Downsample / Upsample, :152-181 ResBlock convs) -> implicit GEMM
//   mrag_unfold_t3_bf16   row gather for nn.Conv3d (3,1,1), pad (1,0,0) (openaimodel3d.py:256-268)
//   mrag_geglu_bf16       x * gelu(gate) (attention.py:448-455)
//   mrag_ddim_v_step_f32  CFG + v-prediction DDIM update with dynamic rescale and eta noise (samplers/ddim.py:236-296)
#include "common.h"
#include "../../include/mrag_hip.h"

namespace {

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

// ---------------------------------------------------------------------------------------------- GroupNorm
// pass 1: per-(n, chunk, channel) partial sum / sum of squares in fp32.  A workgroup owns one (n, chunk of pixels); a thread
// owns one 8-channel vector column and strides over the chunk's pixels (coalesced 16-byte loads along C).
struct GnP {
  const bf16_t* x; bf16_t* y; const bf16_t* gamma; const bf16_t* beta; const bf16_t* emb; float* part;
  long long N, HW, C, emb_stride;
  int G, chunks, silu;
  float eps;
  // spatially conditioned form (CogVideoXSpatialNorm3D): y = gn(x) * mod[.., :C] + mod[.., C:], mod at the latent resolution
  const bf16_t* mod;
  int mT, mH, mW, mTz, mshift, msplit;
  long long y_stride_n;
  unsigned* tickets;   // FOLD: one arrival counter per sample (zero between calls)
  float* ab;           // FOLD: per-(n, c) scale / shift
};

constexpr int GN_TICKET_BYTES = 16384, GN_FOLD_MAX_CHUNKS = 128;   // the last-arriver fold: up to 4096 samples, up to 128 chunk partials per channel

// FOLD (opt-in: mrag_groupnorm_args.fold; measured 5 % SLOWER on the UNet steps, see mrag_groupnorm_bf16): the workgroup that arrives LAST for a sample (ticket)
// also does pass 2 for that sample -- one launch less per GroupNorm (105 / 145 launches of 10-13 us per SVD / DynamiCrafter CFG step).  Possible when the partial
// lists are short (<= 128 chunks: the per-frame norms of the spatial blocks, N = 28 / 32 samples); the (t, h, w) norms of the temporal blocks (N = 2, 1024 chunks:
// 2.6-10 MB of partials per sample) keep the parallel fold kernel below.
template <bool FOLD>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnP p) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int c_off = blockIdx.z * 2048;                                  // channel segment of at most 2048 channels
  const int C8 = (int)(((p.C - c_off) < 2048 ? (p.C - c_off) : 2048) / 8);
  const long long px_per_chunk = (p.HW + p.chunks - 1) / p.chunks;
  const long long px0 = chunk * px_per_chunk;
  long long px1 = px0 + px_per_chunk;
  if (px1 > p.HW) px1 = p.HW;
  // threads are laid out [rows_per_pass][C8]; rows_per_pass = 256 / C8 (>= 1; C8 <= 256 enforced by the host)
  const int rpp = 256 / C8;
  const int col = threadIdx.x % C8, rsub = threadIdx.x / C8;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  if (rsub < rpp) {
    const bf16_t* base = p.x + ((long long)n * p.HW) * p.C + c_off + col * 8;
    long long px = px0 + rsub;
#ifndef MRAG_GN_OLD
    // four pixel rows in flight per thread (the chunk is one contiguous run of the sample: px advances by the rows a pass covers), nontemporal: x is streamed once
    // here; accumulated in the same order as the one-at-a-time loop, so the partial sums keep their bits (round 6: the HBM-bound GroupNorm passes sat at 4.8 TB/s
    // against the 6.0 TB/s of the stream copy, whose sweep -- profiles/r6_copy_probe_sweep.txt -- says: contiguous runs, several loads in flight, nontemporal)
    for (; px + 3 * rpp < px1; px += 4 * rpp) {
      u32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = __builtin_nontemporal_load((const u32x4*)(base + (px + u * rpp) * p.C));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8(raw[u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
      }
    }
#endif
    for (; px < px1; px += rpp) {
      float v[8];
      unpack8(*(const u32x4*)(base + px * p.C), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
  }
  // reduce the rpp row-groups through LDS, then one thread per channel vector writes the partial
  __shared__ float red[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][8 + e] = q[e]; }
  __syncthreads();
  if (threadIdx.x < C8) {
    float ts[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ts[e] = 0.f;
    for (int r = 0; r < rpp; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) ts[e] += red[r * C8 + threadIdx.x][e];
    float* out = p.part + (((long long)n * p.chunks + chunk) * p.C + c_off + threadIdx.x * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { out[2 * e] = ts[e]; out[2 * e + 1] = ts[8 + e]; }
  }
  if constexpr (FOLD) {
    // ---- arrival: every partial of this workgroup has left, then the ticket (agent-scope release / acquire, the pattern of topk_scan_kernel's fused merge)
    float* fl = &red[0][0];                                 // 4 352 floats, free again behind the barrier below: [0, C) sums, [2048, 2048 + C) squares, [4096, ..) groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned total = gridDim.x * gridDim.z;
      const unsigned old = __hip_atomic_fetch_add(p.tickets + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = old == total - 1;
      if (last) {
        __hip_atomic_store(p.tickets + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      fl[4351] = last ? 1.f : 0.f;
    }
    __syncthreads();
    if (fl[4351] == 0.f) return;
    __syncthreads();
    // ---- pass 2 for sample n, fixed orders (bit-reproducible): per channel over the chunks, then the analytic embedding terms, per group over its channels
    const int C = (int)p.C, cpg = C / p.G;
    for (int c = threadIdx.x; c < C; c += 256) {
      float cs = 0.f, cq = 0.f;
      const float2* pr = (const float2*)p.part + (long long)n * p.chunks * C + c;
      for (int k = 0; k < p.chunks; ++k) { const float2 v = pr[(long long)k * C]; cs += v.x; cq += v.y; }
      if (p.emb) {
        const float e = bf2f(p.emb[(long long)n * p.emb_stride + c]);
        cq += 2.f * e * cs + (float)p.HW * e * e;           // sum (x+e)^2 = sum x^2 + 2 e sum x + P e^2
        cs += (float)p.HW * e;
      }
      fl[c] = cs; fl[2048 + c] = cq;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < p.G; g += 256) {
      float gs = 0.f, gq = 0.f;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) { gs += fl[c]; gq += fl[2048 + c]; }
      const float cnt = (float)p.HW * (float)cpg;
      const float mean = gs / cnt;
      const float var = fmaxf(gq / cnt - mean * mean, 0.f);
      fl[4096 + 2 * g] = mean; fl[4097 + 2 * g] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int g = c / cpg;
      const float mean = fl[4096 + 2 * g], rstd = fl[4097 + 2 * g];
      const float ga = p.gamma ? bf2f(p.gamma[c]) : 1.f, be = p.beta ? bf2f(p.beta[c]) : 0.f;
      const float e = p.emb ? bf2f(p.emb[(long long)n * p.emb_stride + c]) : 0.f;
      p.ab[((long long)n * C + c) * 2] = ga * rstd;
      p.ab[((long long)n * C + c) * 2 + 1] = be + (e - mean) * ga * rstd;
    }
  }
}

// pass 2 (tiny): one workgroup per (group, n) folds the chunk partials of its channels in a FIXED order (bit-reproducible),
// adds the analytic contribution of the per-(n, c) embedding, and writes per-channel scale / shift a_c, b_c to the workspace.
__global__ __launch_bounds__(256) void gn_fold_kernel(const GnP p, float* ab) {
  __shared__ float red_s[256], red_q[256];
  __shared__ float ch_s[256], ch_q[256];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int cpg = (int)(p.C / p.G);          // <= 256 (host check)
  const int ksl = 256 / cpg;                 // k-slices: thread t sums channel (t % cpg) over chunks k = t / cpg, + ksl, ...
  const int cl = t % cpg, ks = t / cpg;
  float s = 0.f, q = 0.f;
  if (ks < ksl) {
    const int c = g * cpg + cl;
    for (int k = ks; k < p.chunks; k += ksl) {
      const float* pr = p.part + (((long long)n * p.chunks + k) * p.C + c) * 2;
      s += pr[0]; q += pr[1];
    }
  }
  red_s[t] = s; red_q[t] = q;
  __syncthreads();
  if (t < cpg) {                             // fixed order over the slices, then the analytic embedding terms
    s = 0.f; q = 0.f;
    for (int k = 0; k < ksl; ++k) { s += red_s[k * cpg + t]; q += red_q[k * cpg + t]; }
    if (p.emb) {
      const float e = bf2f(p.emb[(long long)n * p.emb_stride + g * cpg + t]);
      q += 2.f * e * s + (float)p.HW * e * e;   // sum (x+e)^2 = sum x^2 + 2 e sum x + P e^2
      s += (float)p.HW * e;
    }
    ch_s[t] = s; ch_q[t] = q;
  }
  __syncthreads();
  if (t == 0) {
    s = 0.f; q = 0.f;
    for (int c = 0; c < cpg; ++c) { s += ch_s[c]; q += ch_q[c]; }
    red_s[0] = s; red_q[0] = q;
  }
  __syncthreads();
  const float cnt = (float)p.HW * (float)cpg;
  const float mean = red_s[0] / cnt;
  const float var = fmaxf(red_q[0] / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + p.eps);
  if (t < cpg) {
    const int c = g * cpg + t;
    const float ga = p.gamma ? bf2f(p.gamma[c]) : 1.f, be = p.beta ? bf2f(p.beta[c]) : 0.f;
    const float e = p.emb ? bf2f(p.emb[(long long)n * p.emb_stride + c]) : 0.f;
    ab[((long long)n * p.C + c) * 2] = ga * rstd;
    ab[((long long)n * p.C + c) * 2 + 1] = be + (e - mean) * ga * rstd;
  }
}

// pass 3: stream y = act(x * a_c + b_c) with the per-channel pairs parked in LDS; enough workgroups to fill the chip even
// when N is 1 or 2 (GroupNorm over (t, h, w) of the temporal blocks)
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnP p, const float* ab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* abl = (float2*)smem;          // [C]
  const int n = blockIdx.y;
  for (int c = threadIdx.x; c < p.C; c += 256) abl[c] = ((const float2*)ab)[(long long)n * p.C + c];
  __syncthreads();
  const long long C8 = p.C / 8;
  const long long vecs = p.HW * C8;
  const bf16_t* xb = p.x + (long long)n * p.HW * p.C;
  bf16_t* yb = p.y + (long long)n * p.HW * p.C;
  auto one = [&](const u32x4 raw, const unsigned cvv) {
    const int c0 = (int)cvv * 8;
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 w = abl[c0 + e];
      const float o = v[e] * w.x + w.y;
      v[e] = p.silu ? silu_f(o) : o;
    }
    return pack8(v);
  };
#ifndef MRAG_GN_OLD
  // a workgroup walks CONTIGUOUS runs of 4 x 256 vectors (16 KiB), four nontemporal loads in flight per thread (x is not read again before the residual add at the
  // end of the block; y stays a plain store: the convolution behind it reads it nine times).  Channel vector of a thread's first element of a run by one modulo,
  // then conditional subtracts per unrolled vector and per grid stride of runs (a 64-bit modulo per vector is ~100 instructions against 8 FMAs).
  constexpr int U = 4;
  const long long run = 256LL * U, nrun = vecs / run;
  const unsigned c8u = (unsigned)C8, ustep = 256u % c8u, rstep = (unsigned)(((long long)gridDim.x * run) % C8);
  unsigned cv0 = (unsigned)(((long long)blockIdx.x * run + threadIdx.x) % C8);
  for (long long r = blockIdx.x; r < nrun; r += gridDim.x, cv0 = cv0 + rstep >= c8u ? cv0 + rstep - c8u : cv0 + rstep) {
    const long long i = r * run + threadIdx.x;
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) raw[u] = __builtin_nontemporal_load((const u32x4*)(xb + (i + u * 256) * 8));
    unsigned cv = cv0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      *(u32x4*)(yb + (i + u * 256) * 8) = one(raw[u], cv);
      cv = cv + ustep >= c8u ? cv + ustep - c8u : cv + ustep;
    }
  }
  const long long tail0 = nrun * run;
#else
  const long long tail0 = 0;
#endif
  // (the vectors behind the last whole run -- or, in the MRAG_GN_OLD developer build, all of them: one vector per thread and grid stride)
  const unsigned stride = gridDim.x * 256u, cstep = stride % (unsigned)C8;
  unsigned cv = (unsigned)((tail0 + blockIdx.x * 256u + threadIdx.x) % C8);
  for (long long i = tail0 + (long long)blockIdx.x * 256 + threadIdx.x; i < vecs; i += stride, cv = cv + cstep >= (unsigned)C8 ? cv + cstep - (unsigned)C8 : cv + cstep)
    *(u32x4*)(yb + i * 8) = one(*(const u32x4*)(xb + i * 8), cv);
}

// pass 3, spatially conditioned: x [N, (t, y, x), C]; mod [N, Tz, H >> shift, W >> shift, 2C] holds conv_y(zq) | conv_b(zq) at the latent
// resolution (a 1x1x1 convolution commutes with the nearest-neighbour upsampling in front of it), so the upsampled maps never exist.
// Frame map = F.interpolate(nearest) of diffusers' CogVideoXSpatialNorm3D: odd T > 1 -> frame 0 from latent frame 0 and the rest
// resampled separately; otherwise floor(t * Tz / T).  A thread owns one (pixel, 8-channel vector); 256 / (C / 8) pixels per pass.
__global__ __launch_bounds__(256) void gn_apply_mod_kernel(const GnP p, const float* ab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* abl = (float2*)smem;          // [C]
  const int n = blockIdx.y;
  for (int c = threadIdx.x; c < p.C; c += 256) abl[c] = ((const float2*)ab)[(long long)n * p.C + c];
  __syncthreads();
  const int C8 = (int)(p.C / 8), ppb = 256 / C8;
  const int col = threadIdx.x % C8, rsub = threadIdx.x / C8;
  const int c0 = col * 8;
  const unsigned HWf = (unsigned)(p.mH * p.mW);
  const int hz = p.mH >> p.mshift, wz = p.mW >> p.mshift;
  const bf16_t* xb = p.x + (long long)n * p.HW * p.C;
  bf16_t* yb = p.y + (long long)n * (p.y_stride_n ? p.y_stride_n : p.HW * p.C);
  const bf16_t* mb = p.mod + (long long)n * p.mTz * hz * wz * 2 * p.C;
  for (unsigned px = blockIdx.x * ppb + rsub; px < (unsigned)p.HW; px += gridDim.x * ppb) {
    const unsigned t = px / HWf, rem = px - t * HWf;
    const unsigned y = rem / (unsigned)p.mW, x = rem - y * (unsigned)p.mW;
    unsigned tz;
    if (p.msplit) tz = t == 0 ? 0u : 1u + ((t - 1) * (unsigned)(p.mTz - 1)) / (unsigned)(p.mT - 1);
    else tz = (t * (unsigned)p.mTz) / (unsigned)p.mT;
    const bf16_t* m = mb + (((long long)tz * hz + (y >> p.mshift)) * wz + (x >> p.mshift)) * 2 * p.C + c0;
    float v[8], my[8], mbv[8];
    unpack8(*(const u32x4*)(xb + (long long)px * p.C + c0), v);
    unpack8(*(const u32x4*)m, my);
    unpack8(*(const u32x4*)(m + p.C), mbv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 w = abl[c0 + e];
      const float o = (v[e] * w.x + w.y) * my[e] + mbv[e];
      v[e] = p.silu ? silu_f(o) : o;
    }
    *(u32x4*)(yb + (long long)px * p.C + c0) = pack8(v);
  }
}

// ---------------------------------------------------------------------------------------------- conv row gathers
// dst[(n, yo, xo), (ky, kx, c)] = src[n, yo*stride + ky - 1, xo*stride + kx - 1, c]   (zero outside; src read at (y/2, x/2)
// of the stored tensor when `up` (nearest x2 upsample fused in front)); columns [9C, Kpad) are zero.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* src, bf16_t* dst, int N, int H, int W, int C, int stride, int up, int Kpad) {
  const int Hi = up ? 2 * H : H, Wi = up ? 2 * W : W;   // logical input extent
  const int Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
  const int K8 = Kpad / 8, C8 = C / 8;
  const long long total = (long long)N * Ho * Wo * K8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kv = (int)(i % K8);
    const long long row = i / K8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (kv < 9 * C8) {
      const int tap = kv / C8, cv = kv - tap * C8;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int xo = (int)(row % Wo);
      const long long r2 = row / Wo;
      const int yo = (int)(r2 % Ho), n = (int)(r2 / Ho);
      const int yi = yo * stride + ky - 1, xi = xo * stride + kx - 1;
      if (yi >= 0 && yi < Hi && xi >= 0 && xi < Wi) {
        const int ys = up ? yi >> 1 : yi, xs = up ? xi >> 1 : xi;
        v = *(const u32x4*)(src + (((long long)n * H + ys) * W + xs) * C + cv * 8);
      }
    }
    *(u32x4*)(dst + row * Kpad + kv * 8) = v;
  }
}

// dst[(b, t, hw), (kt, c)] = src[b, t + kt - 1, hw, c]  (zero for t + kt - 1 outside [0, T))
__global__ __launch_bounds__(256) void unfold_t3_kernel(const bf16_t* src, bf16_t* dst, int B, int T, long long HW, int C) {
  const int C8 = C / 8;
  const long long total = (long long)B * T * HW * 3 * C8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kv = (int)(i % (3 * C8));
    const long long row = i / (3 * C8);
    const int kt = kv / C8, cv = kv - kt * C8;
    const long long hw = row % HW;
    const long long bt = row / HW;
    const int t = (int)(bt % T) + kt - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t >= 0 && t < T) v = *(const u32x4*)(src + (((bt / T) * T + t) * HW + hw) * C + cv * 8);
    *(u32x4*)(dst + row * (3LL * C) + kv * 8) = v;
  }
}

// y[r, j] = x[r, j] * gelu_erf(x[r, inner + j])
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* x, bf16_t* y, long long rows, long long inner8) {
  const long long total = rows * inner8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / inner8, j = i - r * inner8;
    float a[8], g[8];
    unpack8(*(const u32x4*)(x + (r * 2 * inner8 + j) * 8), a);
    unpack8(*(const u32x4*)(x + (r * 2 * inner8 + inner8 + j) * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= gelu_erf_f(g[e]);
    *(u32x4*)(y + i * 8) = pack8(a);
  }
}

// samplers/ddim.py:236-296 (v-parameterisation):  v = v_u + s (v_c - v_u)   [cond FIRST in the batch, :219-237]
//   eps = sa v + sb x ; x0 = (sa x - sb v) * rescale ; x <- sqrt(a_prev) x0 + dir eps + sigma noise
__global__ __launch_bounds__(256) void ddim_v_kernel(const bf16_t* v_pred, float* x, const float* noise, long long n, float s, float sa, float sb,
                                                     float rescale, float sqrt_aprev, float dir, float sigma) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float vc = bf2f(v_pred[i]), vu = bf2f(v_pred[n + i]);
    const float v = vu + s * (vc - vu);
    const float xx = x[i];
    const float eps = sa * v + sb * xx;
    const float x0 = (sa * xx - sb * v) * rescale;
    x[i] = sqrt_aprev * x0 + dir * eps + (noise ? sigma * noise[i] : 0.f);
  }
}

inline unsigned grid_for(long long items) {
  long long b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int64_t mrag_groupnorm_workspace_bytes(int64_t N, int64_t C, int32_t chunks) {
  if (N <= 0 || C <= 0 || chunks <= 0) return 0;
  return GN_TICKET_BYTES + (N * chunks * C * 2 + N * C * 2) * (int64_t)sizeof(float);   // arrival counters, chunk partials, per-(n, c) scale / shift
}

extern "C" int mrag_groupnorm_bf16(void* stream, const mrag_groupnorm_args* a) {
  if (!a || !a->x || !a->y || !a->workspace) return MRAG_EINVAL;
  if (a->N <= 0 || a->HW <= 0 || a->C <= 0 || a->G <= 0 || a->chunks <= 0) return MRAG_EINVAL;
  if (a->C % 8 != 0 || a->C % a->G != 0) return MRAG_EINVAL;
  if (a->C > 8192 || a->N > 65535 || a->chunks > 1024) return MRAG_ENOTSUP;
  if (((uintptr_t)a->x | (uintptr_t)a->y) & 15) return MRAG_EINVAL;
  if (a->mod) {
    if (a->emb || a->mod_T <= 0 || a->mod_H <= 0 || a->mod_W <= 0 || a->mod_Tz <= 0 || a->mod_shift < 0 || a->mod_shift > 4) return MRAG_EINVAL;
    if ((int64_t)a->mod_T * a->mod_H * a->mod_W != a->HW || a->HW >= (1LL << 31)) return MRAG_EINVAL;
    if ((a->mod_H & ((1 << a->mod_shift) - 1)) || (a->mod_W & ((1 << a->mod_shift) - 1))) return MRAG_EINVAL;
    if (a->mod_split && (a->mod_T < 3 || !(a->mod_T & 1) || a->mod_Tz < 2)) return MRAG_EINVAL;
    if (a->C > 2048 || 256 % (a->C / 8) != 0 || ((uintptr_t)a->mod & 15) || a->y_stride_n % 8 != 0) return MRAG_EINVAL;
  } else if (a->y_stride_n != 0) return MRAG_EINVAL;
  GnP p{};
  p.x = (const bf16_t*)a->x; p.y = (bf16_t*)a->y; p.gamma = (const bf16_t*)a->gamma; p.beta = (const bf16_t*)a->beta;
  if (((uintptr_t)a->workspace) & 15) return MRAG_EINVAL;
  p.emb = (const bf16_t*)a->emb; p.tickets = (unsigned*)a->workspace; p.part = (float*)((char*)a->workspace + GN_TICKET_BYTES);
  p.N = a->N; p.HW = a->HW; p.C = a->C; p.emb_stride = a->emb_stride; p.G = a->G; p.chunks = a->chunks; p.silu = a->silu; p.eps = a->eps;
  hipStream_t s = (hipStream_t)stream;
  if (a->C / a->G > 256) return MRAG_ENOTSUP;
  float* ab = p.part + a->N * a->chunks * a->C * 2;
  p.ab = ab;
  // MEASURED SLOWER and therefore off: same box, interleaved, SVD / DynamiCrafter CFG step 110.4 -> 116.2 ms / 141.6 -> 149.0 ms with the fold by the last
  // arriver (profiles/r6_unet_gn_fold_ab.txt).  Every statistics workgroup (2 048 per call) pays an agent-scope release in front of its ticket -- an L2
  // write-back -- which costs far more than the 10-13 us fold launch it removes.  `fold` stays reachable through the args' `fold` flag (tests keep the path
  // alive; a cheaper arrival protocol would make it pay).
  const bool fold = a->fold == 1 && a->chunks <= GN_FOLD_MAX_CHUNKS && a->C <= 2048 && a->G <= 127 && a->N <= GN_TICKET_BYTES / 4;
  const dim3 sgrid(a->chunks, (unsigned)a->N, (unsigned)((a->C + 2047) / 2048));
  if (fold) {
    MRAG_LAUNCH(gn_stats_kernel<true>, sgrid, dim3(256), 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_GN_STATS_FOLD);
  } else {
    MRAG_LAUNCH(gn_stats_kernel<false>, sgrid, dim3(256), 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_GN_STATS);
    MRAG_LAUNCH(gn_fold_kernel, dim3((unsigned)a->G, (unsigned)a->N), dim3(256), 0, s, p, ab);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_GN_FOLD);
  }
  const size_t lds = (size_t)a->C * 2 * sizeof(float);
  if (a->mod) {
    p.mod = (const bf16_t*)a->mod; p.mT = a->mod_T; p.mH = a->mod_H; p.mW = a->mod_W; p.mTz = a->mod_Tz; p.mshift = a->mod_shift; p.msplit = a->mod_split;
    p.y_stride_n = a->y_stride_n;
    const int ppb = 256 / (int)(a->C / 8);
    long long bxm = (a->HW + 4LL * ppb - 1) / (4LL * ppb);      // >= 4 pixels per thread
    const long long capm = 8192 / a->N > 16 ? 8192 / a->N : 16;
    if (bxm > capm) bxm = capm;
    MRAG_LAUNCH(gn_apply_mod_kernel, dim3((unsigned)bxm, (unsigned)a->N), dim3(256), lds, s, p, (const float*)ab);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_GN_APPLY_MOD);
    return MRAG_OK;
  }
  long long bx = (a->HW * (a->C / 8) + 1023) / 1024;          // >= 4 vectors per thread
  const long long cap = 4096 / a->N > 16 ? 4096 / a->N : 16;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  MRAG_LAUNCH(gn_apply_kernel, dim3((unsigned)bx, (unsigned)a->N), dim3(256), lds, s, p, (const float*)ab);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_GN_APPLY);
  return MRAG_OK;
}

extern "C" int mrag_im2col3x3_bf16(void* stream, const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride,
                                   int32_t upsample, int32_t Kpad) {
  if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MRAG_EINVAL;
  if (C % 8 != 0 || Kpad % 8 != 0 || Kpad < 9 * C || (stride != 1 && stride != 2)) return MRAG_EINVAL;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return MRAG_EINVAL;
  const int Hi = upsample ? 2 * H : H, Wi = upsample ? 2 * W : W;
  const long long total = (long long)N * ((Hi - 1) / stride + 1) * ((Wi - 1) / stride + 1) * (Kpad / 8);
  MRAG_LAUNCH(im2col3x3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, N, H, W, C, stride,
              upsample ? 1 : 0, Kpad);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_unfold_t3_bf16(void* stream, const void* src, void* dst, int32_t B, int32_t T, int64_t HW, int32_t C) {
  if (!src || !dst || B <= 0 || T <= 0 || HW <= 0 || C <= 0 || C % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return MRAG_EINVAL;
  const long long total = (long long)B * T * HW * 3 * (C / 8);
  MRAG_LAUNCH(unfold_t3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, B, T, (long long)HW, C);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_geglu_bf16(void* stream, const void* x, void* y, int64_t rows, int64_t inner) {
  if (!x || !y || rows <= 0 || inner <= 0 || inner % 8 != 0) return MRAG_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return MRAG_EINVAL;
  MRAG_LAUNCH(geglu_kernel, dim3(grid_for(rows * inner / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long long)rows,
              (long long)(inner / 8));
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int mrag_ddim_v_step_f32(void* stream, const void* v_pred, float* x, const float* noise, int64_t n, float guidance, float sqrt_alpha_t,
                                    float sqrt_one_minus_alpha_t, float rescale, float sqrt_alpha_prev, float dir_coef, float sigma) {
  if (!v_pred || !x || n <= 0) return MRAG_EINVAL;
  MRAG_LAUNCH(ddim_v_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v_pred, x, noise, (long long)n, guidance,
              sqrt_alpha_t, sqrt_one_minus_alpha_t, rescale, sqrt_alpha_prev, dir_coef, sigma);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}
