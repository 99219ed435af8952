// developer experiment: sustained MFMA throughput under the power cap, 16x16x32 vs 32x32x16 bf16 (no memory traffic)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += 512) ((int*)lds)[i] = 0x3c003c00 + i;
  __syncthreads();
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (short)(0x3c00 + threadIdx.x + i + e);
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (short)(0x3c00 + threadIdx.x * 3 + i + e);
  if (MODE == 0 || MODE == 2) {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = lds + (wave & 1) * 16384 + (lane & 15) * 128 + ((lane >> 4) ^ (lane & 7)) * 16;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 2) {      // the GEMM's LDS traffic: 12 conflict-free ds_read_b128 per 32 MFMAs
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(base + i * 2048 + (it & 1) * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(base + 32768 + j * 2048 + (it & 1) * 64);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)     // same flops per iteration as MODE 0: 8 tiles x 2 k-steps x 32768
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j + 2 * kk], a[i + 4 * kk], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int r = 0; r < 4; ++r) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 4.0 * 256 * 8 * (double)iters * 32 * 16384.0;
      printf("%s: %.1f ms  %.0f TFLOP/s\n", mode == 0 ? "16x16x32" : mode == 1 ? "32x32x16" : "16x16x32 + LDS reads", ms, flop / (ms * 1e-3) / 1e12);
    }
  return 0;
}
