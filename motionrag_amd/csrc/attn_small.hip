// attn_small.hip -- softmax(scale * Q K^T) V for SHORT sequences at head dims the flash kernels were not built for (they are specialised for 64).
// First user: the CLIP-ViT-H image encoder of the SVD path (transformers CLIPVisionModelWithProjection behind diffusers' StableVideoDiffusionPipeline._encode_image,
// src/projects/svd/pipelines/pipeline.py:113-119): 257 tokens, 16 heads of 80, once per clip.  Not a throughput kernel: plain fp32 FMAs, no MFMA.
//   one workgroup per (batch, head): K and V of the head are staged in LDS as bf16 (Skv * D * 4 bytes <= 128 KB); a thread owns query rows tid, tid + 256, ...
//   with the row's Q and its output accumulator in registers, walks the keys in blocks of 16 (scores -> block max -> one rescale of the accumulator -> exp2 -> P V).
//   Every lane reads the same K / V element at the same time: LDS broadcasts, no bank conflicts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrag_hip.h"
#include "common.h"

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

struct SmallP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
  long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss;
  int B, H, Sq, Skv;
  float qscale;   // scale * log2 e
};

template <int D>
__global__ __launch_bounds__(256) void attn_small_kernel(const SmallP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* ks = (bf16_t*)smem;
  bf16_t* vs = ks + (long long)p.Skv * D;
  const int tid = threadIdx.x;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const bf16_t* kg = p.K + (long long)b * p.k_sb + (long long)h * p.k_sh;
  const bf16_t* vg = p.V + (long long)b * p.v_sb + (long long)h * p.v_sh;
  constexpr int DV = D / 8;
  for (int i = tid; i < p.Skv * DV; i += 256) {
    const int j = i / DV, c = i - j * DV;
    *(u32x4*)(ks + j * D + c * 8) = *(const u32x4*)(kg + (long long)j * p.k_ss + c * 8);
    *(u32x4*)(vs + j * D + c * 8) = *(const u32x4*)(vg + (long long)j * p.v_ss + c * 8);
  }
  __syncthreads();
  for (int row = tid; row < p.Sq; row += 256) {
    float q[D], acc[D];
    const bf16_t* qg = p.Q + (long long)b * p.q_sb + (long long)row * p.q_ss + (long long)h * p.q_sh;
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      float f[8]; unpack8(*(const u32x4*)(qg + c * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { q[c * 8 + e] = f[e] * p.qscale; acc[c * 8 + e] = 0.f; }
    }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < p.Skv; j0 += 16) {
      float s[16];
      float bm = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        float d = -INFINITY;
        if (j < p.Skv) {
          d = 0.f;
          const bf16_t* kr = ks + j * D;
#pragma unroll
          for (int c = 0; c < DV; ++c) {
            float f[8]; unpack8(*(const u32x4*)(kr + c * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(q[c * 8 + e], f[e], d);
          }
        }
        s[jj] = d;
        bm = fmaxf(bm, d);
      }
      const float m_new = fmaxf(m, bm);
      const float alpha = exp2f(m - m_new);          // m = -inf on the first block: exp2(-inf) = 0, acc and l are 0 anyway
      l *= alpha;
#pragma unroll
      for (int e = 0; e < D; ++e) acc[e] *= alpha;
      m = m_new;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        if (j >= p.Skv) break;
        const float pj = exp2f(s[jj] - m);
        l += pj;
        const bf16_t* vr = vs + j * D;
#pragma unroll
        for (int c = 0; c < DV; ++c) {
          float f[8]; unpack8(*(const u32x4*)(vr + c * 8), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[c * 8 + e] = fmaf(pj, f[e], acc[c * 8 + e]);
        }
      }
    }
    const float inv = 1.0f / l;
    bf16_t* og = p.O + (long long)b * p.o_sb + (long long)row * p.o_ss + (long long)h * D;
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = acc[c * 8 + e] * inv;
      *(u32x4*)(og + c * 8) = pack8(f);
    }
  }
}

template <int D>
int launch_small(hipStream_t s, const SmallP& p) {
  const size_t lds = (size_t)p.Skv * D * 2 * 2;
  if (lds > 128 * 1024) return MRAG_ENOTSUP;
  auto kfn = attn_small_kernel<D>;
  const hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH(kfn, dim3((unsigned)(p.B * p.H)), dim3(256), lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_ATTN_SMALL);
  return MRAG_OK;
}

}  // namespace

extern "C" int mrag_attn_small_bf16(void* stream, const mrag_attn_args* a, int32_t head_dim) {
  if (!a || !a->Q || !a->K || !a->V || !a->O || a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Skv <= 0) return MRAG_EINVAL;
  if (a->mask || a->bias || a->resid || a->kv_batch_div != 1 || a->q_prescaled) return MRAG_ENOTSUP;
  if (((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) & 15) return MRAG_EINVAL;
  if ((a->q_sb | a->q_ss | a->q_sh | a->k_sb | a->k_ss | a->k_sh | a->v_sb | a->v_ss | a->v_sh | a->o_sb | a->o_ss) % 8 != 0) return MRAG_EINVAL;
  SmallP p;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V; p.O = (bf16_t*)a->O;
  p.q_sb = a->q_sb; p.q_ss = a->q_ss; p.q_sh = a->q_sh; p.k_sb = a->k_sb; p.k_ss = a->k_ss; p.k_sh = a->k_sh;
  p.v_sb = a->v_sb; p.v_ss = a->v_ss; p.v_sh = a->v_sh; p.o_sb = a->o_sb; p.o_ss = a->o_ss;
  p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Skv = a->Skv;
  p.qscale = a->scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 32: return launch_small<32>(s, p);
    case 80: return launch_small<80>(s, p);
    case 96: return launch_small<96>(s, p);
    case 128: return launch_small<128>(s, p);
    default: return MRAG_ENOTSUP;
  }
}
