// developer check: v_mfma_scale_f32_32x32x64_f8f6f4 with random e4m3 operands, a non-trivial B scale and a non-zero C, against a host reference
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb, float c0) {
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  i32x8 a = *(const i32x8*)(A + r * 64 + hh * 32), b = *(const i32x8*)(B + r * 64 + hh * 32);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = c0;
  f32x16 d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * hh) * 32 + r] = d[i];
}
static float dec(unsigned char x) {
  int s = x >> 7, e = (x >> 3) & 15, m = x & 7;
  float v = e ? ldexpf(1.0f + m / 8.0f, e - 7) : ldexpf(m / 8.0f, -6);
  return s ? -v : v;
}
int main() {
  unsigned char hA[32 * 64], hB[32 * 64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { do hA[i] = rand() & 0xff; while ((hA[i] & 0x7f) == 0x7f); do hB[i] = rand() & 0xff; while ((hB[i] & 0x7f) == 0x7f); }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  const int sbs[] = {0x7f7f7f7f, 0x71717171, 0x7f7f7f71, 0x717f7f7f};
  for (int t = 0; t < 4; ++t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0x7f7f7f7f, sbs[t], -2.5f);
    float hD[32 * 32];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int e = 127; e >= 113; e -= 14) {
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double acc = 0;
        for (int kk = 0; kk < 64; ++kk) acc += (double)dec(hA[i * 64 + kk]) * dec(hB[j * 64 + kk]);
        const double ref = acc * ldexp(1.0, e - 127) - 2.5;
        maxerr = fmax(maxerr, fabs(ref - hD[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
      }
      printf("scale_b=%08x  vs reference with 2^%d: max |err| %g (max |ref| %g)\n", sbs[t], e - 127, maxerr, maxref);
    }
  }
  return 0;
}
