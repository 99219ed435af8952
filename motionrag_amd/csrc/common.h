// common.h -- device helpers shared by the gfx950 kernels of libmrag_hip.so.
// Wavefront = 64 lanes everywhere (CDNA4); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MRAG_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf_round(float f) { return bf2f(f2bf(f)); }   // value after a bf16 store + reload
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
// one v_cvt_pk_bf16_f32: lo -> bits [15:0], hi -> bits [31:16]
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  const bf16v2 b = __builtin_convertvector(f, bf16v2);
  return __builtin_bit_cast(unsigned, b);
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  == x * sigmoid(2 u)
  // x * sigmoid(2u) with u = x (c1 + c2 x^2); exp2 with the constants folded and v_rcp_f32 (1 ulp) instead of the IEEE division sequence
  // (~10 instructions) that `x / (1 + e)` compiles to: the result is rounded to bf16 anyway
  const float t = __builtin_fmaf(x * x, 0.0356774081363001f, 0.7978845608028654f);
  const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * (t * x));      // exp(-2u) = exp2(-2 log2(e) u)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// the same function on two values with packed fp32 arithmetic (v_pk_mul / v_pk_fma / v_pk_add: five vector instructions + two v_exp + two v_rcp for TWO values,
// 4.5 issue slots per value against 7): the identical operations per value, so the identical bits.  For epilogues that one wave per SIMD issues alone.
__device__ __forceinline__ f32x2 gelu_tanh_f2(const f32x2 x) {
  const f32x2 c2 = {0.0356774081363001f, 0.0356774081363001f}, c1 = {0.7978845608028654f, 0.7978845608028654f};
  const f32x2 t = __builtin_elementwise_fma(x * x, c2, c1);
  const f32x2 z = (t * x) * -2.8853900817779268f;
  f32x2 e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  e = e + 1.0f;
  const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  return x * r;
}
// exact-erf GELU.  Phi's upper tail by Abramowitz & Stegun 7.1.26, Q(a) = 0.5 erfc(a / sqrt 2) = 0.5 t P(t) exp(-a^2 / 2), t = 1 / (1 + p a / sqrt 2)
// (|error| <= 0.75e-7, far below a bf16 ulp), used symmetrically:  gelu(x) = x Phi(x) = max(x, 0) - |x| Q(|x|)  -- no copysign, no 1 + erf, no
// 0.5 x: one v_rcp + one v_exp + 6 FMAs + 5 multiplies / max (13 vector instructions; the 1 - erf form took 16; the branchy libm erff ~40).  In a
// short-K GEMM (the UNets' K = 320 GEGLU projections) this epilogue is as long as the K loop.  |x| rides as a source modifier.
#ifndef MRAG_GELU_PACKED
#define MRAG_GELU_PACKED 0     // 1: the same operations on value PAIRS with v_pk_*_f32 (A/B builds; measured no faster: packed fp32 issues at half rate here)
#endif
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.7071067811865476f, 1.0f));   // v_rcp_f32 (1 ulp), not the IEEE reciprocal sequence
  float pl = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);                                // 0.5 P(t): the 1/2 of Q folded into the coefficients
  pl = __builtin_fmaf(pl, t, 0.5f * 1.421413741f);
  pl = __builtin_fmaf(pl, t, 0.5f * -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f((ax * -0.7213475204444817f) * ax);                               // exp(-x^2 / 2)
  const float q = ((pl * t) * e) * ax;                                                                      // |x| Q(|x|)
  return fmaxf(x, 0.0f) - q;
}
__device__ __forceinline__ f32x2 gelu_erf_f2(const f32x2 x) {
#if MRAG_GELU_PACKED
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 den = __builtin_elementwise_fma(ax, f32x2{0.3275911f * 0.7071067811865476f, 0.3275911f * 0.7071067811865476f}, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 pl = __builtin_elementwise_fma(t, f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  const f32x2 z = (ax * -0.7213475204444817f) * ax;
  const f32x2 e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  const f32x2 q = ((pl * t) * e) * ax;
  return f32x2{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)} - q;
#else
  return f32x2{gelu_erf_f(x[0]), gelu_erf_f(x[1])};
#endif
}
// GEGLU on four (value, gate) pairs of one lane: v <- bf16(v) * act(bf16(g)) (the reference rounds both halves of proj(x) to bf16 before the product:
// nn.Linear output dtype); TANH selects the tanh gate (T5's gated-gelu), else the exact-erf one (lvdm attention.py:448-455 / diffusers GEGLU)
template <bool TANH>
__device__ __forceinline__ void geglu4(float (&v)[4], const float (&g)[4]) {
  const unsigned gw0 = pack_bf2(g[0], g[1]), gw1 = pack_bf2(g[2], g[3]), vw0 = pack_bf2(v[0], v[1]), vw1 = pack_bf2(v[2], v[3]);
  const f32x2 g01 = {__uint_as_float(gw0 << 16), __uint_as_float(gw0 & 0xffff0000u)}, g23 = {__uint_as_float(gw1 << 16), __uint_as_float(gw1 & 0xffff0000u)};
  const f32x2 v01 = {__uint_as_float(vw0 << 16), __uint_as_float(vw0 & 0xffff0000u)}, v23 = {__uint_as_float(vw1 << 16), __uint_as_float(vw1 & 0xffff0000u)};
  const f32x2 a01 = TANH ? gelu_tanh_f2(g01) : gelu_erf_f2(g01), a23 = TANH ? gelu_tanh_f2(g23) : gelu_erf_f2(g23);
  const f32x2 o01 = v01 * a01, o23 = v23 * a23;
  v[0] = o01[0]; v[1] = o01[1]; v[2] = o23[0]; v[3] = o23[1];
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// async global -> LDS copy of 16 bytes per lane; LDS destination is
// wave-uniform base + lane*16 (cdna_hip_programming.md section 5 caveat).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

// same copy with the SCALAR-base form: source = sbase (wave-uniform, SGPR pair) + voff (32-bit per-lane byte offset).  Keeps
// loop-invariant lane offsets in ONE VGPR each instead of 64-bit VGPR pointers that hipcc recomputes / spills per tile (its spill
// reloads come with s_waitcnt vmcnt(0), which would drain the DMA ring).  The statement is invisible to hipcc's vmcnt bookkeeping:
// the caller retires it with its own counted s_waitcnt.  M0 (LDS destination base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// bijective XCD-aware remap of a 1-D grid: blocks with equal (id % 8) share an
// XCD's L2, so give each XCD one contiguous chunk of the logical tile space.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (id >> 3);
}

// diagnostic launch counters (include/mrag_hip.h: enum mrag_kernel_id, mrag_dispatch_counts); the table lives in api.hip
extern unsigned long long mrag_dispatch_table[];
#define MRAG_COUNT(id) ((void)__atomic_fetch_add(&mrag_dispatch_table[(id)], 1ull, __ATOMIC_RELAXED))

#define MRAG_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

// hipGetLastError() is per-thread and sticky across unrelated runtime calls (torch leaves benign errors behind):
// clear it right before our launch so that MRAG_LAUNCH_CHECK reports only this launch's status.
#define MRAG_LAUNCH(...)            \
  do {                              \
    (void)hipGetLastError();        \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)
