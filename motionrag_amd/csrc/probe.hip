// probe.hip -- measurement probes of libmrag_hip.so (NOT on the product path): what this part sustains, so that bench.py can print the
// ceilings beside the roofline fractions it reports (SURVEY.md section 8d asks for measured ceilings next to the nominal peaks).
//
//   mrag_probe_mfma_bf16: a register-resident loop of v_mfma_f32_16x16x32_bf16 on caller-provided (random) operand bits -- no LDS, no memory traffic
//   inside the loop, 8 waves per CU (2 per SIMD), 32 independent accumulator tiles per wave so the matrix pipe never waits for a result.  This is
//   the upper bound of ANY bf16 kernel on the box in its current power state: the nominal 2.5 PFLOP/s assumes the 2.4 GHz peak clock, which the
//   part does not hold under a dense matrix load on random data (profiles/r2_mfma_shape_power.txt).
//   mrag_probe_mfma_f32: the same for v_mfma_f32_32x32x2_f32 (nominal 157 TFLOP/s), the pipe of the retrieval fan-out kernel.
//   mrag_probe_stream_copy: a grid-stride copy with 16-byte loads and stores, four of each in flight per lane -- the 1 : 1 read / write stream every
//   "HBM-bound" kernel of the library is priced against: 5.9-6.0 TB/s on this pool (the guide's float4 copy: 6.3 of the 8 TB/s; `dst.copy_(src)` on uint8, which
//   bench.py quoted until round 5: 4.7-5.1 -- not a ceiling, the library's own LayerNorm and top-k kernels move more).
#include "common.h"
#include "../../include/mrag_hip.h"

namespace {

__global__ __launch_bounds__(512) void probe_mfma_kernel(const bf16x8* __restrict__ operands, long long n_frag, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[4];
  const long long t = (long long)blockIdx.x * 512 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = operands[(t * 12 + i) % n_frag];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = operands[(t * 12 + 8 + j) % n_frag];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[t] = s;
}

// the fp32 matrix pipe the retrieval fan-out kernel runs on: v_mfma_f32_32x32x2_f32, four waves per CU (one per SIMD, the fan-out kernel's shape), 8 independent
// 32x32 accumulators per wave -- the kernel's own register image without its LDS / DMA traffic
__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(const float* __restrict__ operands, long long n, float* __restrict__ out, int iters) {
  float a[4], b[8][4];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = operands[(t * 36 + i) % n];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) b[j][i] = operands[(t * 36 + 4 + 4 * j + i) % n];
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j][i], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[t] = s;
}

// 16 bytes per lane and access, UNROLL independent loads before the first store.  CONTIG: a workgroup's UNROLL accesses of one iteration are ONE contiguous run of
// UNROLL x 4 KiB (grid-stride over such runs); otherwise they are UNROLL 4-KiB runs a whole grid stride apart (the first form: 4.4 TB/s, slower than torch's copy --
// too many concurrent DRAM pages).  NT: nontemporal loads / stores (streamed once, as the library's LayerNorm kernels do).
template <int UNROLL, bool CONTIG, bool NT>
__global__ __launch_bounds__(256) void probe_stream_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n16) {
  auto ld = [&](long long i) { return NT ? __builtin_nontemporal_load(src + i) : src[i]; };
  auto st = [&](long long i, u32x4 v) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; };
  if constexpr (CONTIG) {
    const long long run = 256LL * UNROLL, nrun = n16 / run;
    for (long long r = blockIdx.x; r < nrun; r += gridDim.x) {
      const long long i = r * run + threadIdx.x;
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = ld(i + u * 256);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) st(i + u * 256, v[u]);
    }
    for (long long i = nrun * run + (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) st(i, ld(i));
  } else {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = ld(i + u * stride);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) st(i + u * stride, v[u]);
    }
    for (; i < n16; i += stride) st(i, ld(i));
  }
}

}  // namespace

// variant: bits 0-3 the kernel form (0 = the shipped one), bits 4-11 workgroups per CU (0 = the shipped count) -- developer sweep (tools/microbench.py copy_probe)
extern "C" int mrag_probe_stream_copy(void* stream, const void* src, void* dst, int64_t bytes, int32_t variant) {
  if (!src || !dst || bytes < 16 || (bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15) || variant < 0) return MRAG_EINVAL;
  const long long n16 = bytes / 16;
  const int form = variant & 15, per_cu = (variant >> 4) & 0xff ? (variant >> 4) & 0xff : 16;   // (swept: profiles/r6_copy_probe_sweep.txt -- 16 workgroups per CU of the contiguous nontemporal form: 5.9-6.0 TB/s)
  long long wgs = (n16 + 4 * 256 - 1) / (4 * 256);
  if (wgs > 256LL * per_cu) wgs = 256LL * per_cu;
  const dim3 grid((unsigned)wgs), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (form) {
    case 0: MRAG_LAUNCH((probe_stream_copy_kernel<4, true, true>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    case 1: MRAG_LAUNCH((probe_stream_copy_kernel<4, true, false>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    case 2: MRAG_LAUNCH((probe_stream_copy_kernel<8, true, true>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    case 3: MRAG_LAUNCH((probe_stream_copy_kernel<4, false, false>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    case 4: MRAG_LAUNCH((probe_stream_copy_kernel<2, true, true>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    case 5: MRAG_LAUNCH((probe_stream_copy_kernel<1, true, true>), grid, block, 0, s, (const u32x4*)src, (u32x4*)dst, n16); break;
    default: return MRAG_EINVAL;
  }
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int64_t mrag_probe_mfma_f32_flops(int32_t iters) {
  // 256 workgroups x 4 waves x iters x 32 MFMAs x (2 * 32 * 32 * 2) FLOP
  return iters <= 0 ? 0 : 256LL * 4 * (int64_t)iters * 32 * 4096;
}

extern "C" int mrag_probe_mfma_f32(void* stream, const void* operands, int64_t operand_bytes, float* out, int32_t iters) {
  if (!operands || !out || operand_bytes < 4 || iters <= 0 || ((uintptr_t)operands & 3)) return MRAG_EINVAL;
  MRAG_LAUNCH(probe_mfma_f32_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const float*)operands, (long long)(operand_bytes / 4), out, iters);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}

extern "C" int64_t mrag_probe_mfma_flops(int32_t iters) {
  // 256 workgroups x 8 waves x iters x 32 MFMAs x (2 * 16 * 16 * 32) FLOP
  return iters <= 0 ? 0 : 256LL * 8 * (int64_t)iters * 32 * 16384;
}

extern "C" int mrag_probe_mfma_bf16(void* stream, const void* operands, int64_t operand_bytes, float* out, int32_t iters) {
  if (!operands || !out || operand_bytes < 16 || iters <= 0 || ((uintptr_t)operands & 15)) return MRAG_EINVAL;
  MRAG_LAUNCH(probe_mfma_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, (const bf16x8*)operands, (long long)(operand_bytes / 16), out, iters);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}
