// developer experiment: issue cost per wave-instruction of v_exp_f32 / v_add_f32 / v_max3 / v_cvt_pk at 1 and 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&a[i & ~1]));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
  const char* names[] = {"v_exp_f32", "v_add_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_fma_f32", "v_pk_fma_f32"};
  const int iters = 2000;
  for (int op = 0; op < 6; ++op)
    for (int wps = 1; wps <= 4; wps *= 2) {
      const int threads = 64 * 4 * wps;   // wps waves per SIMD on each CU (one block per CU)
      for (int rep = 0; rep < 2; ++rep) {
        switch (op) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        }
        hipDeviceSynchronize();
      }
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-20s %d wave(s)/SIMD: %.2f cycles per wave-instruction (SIMD throughput %.2f cycles/instr)\n", names[op], wps, (double)c / (iters * 16.0), (double)c / (iters * 16.0) / wps);
    }
  return 0;
}
