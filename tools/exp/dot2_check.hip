// what does v_dot2c_f32_bf16 compute on gfx950, alone and in a dependent chain fed by v_cvt_pk_bf16_f32?
//   hipcc --offload-arch=gfx950 -O2 dot2_check.hip -o dot2_check && ./dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16v2));
}
__global__ void chain(const float* x, float* out, int nops) {
  const int i = threadIdx.x;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = x[i * 16 + j];
  float acc = 0.f, ref = 0.f;
  unsigned w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16v2, w[j]), __builtin_bit_cast(bf16v2, 0x3f803f80u), acc, false);
    if (nops) asm volatile("s_nop 4");
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) ref += __uint_as_float(w[j] << 16) + __uint_as_float(w[j] & 0xffff0000u);
  out[i * 2] = acc; out[i * 2 + 1] = ref;
}
int main() {
  const int n = 64;
  float hx[n * 16];
  for (int i = 0; i < n * 16; ++i) hx[i] = (float)((i * 7919) % 1000) / 1000.f;
  float *dx, *dout;
  (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dout, n * 8);
  (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  for (int nops = 0; nops < 2; ++nops) {
    chain<<<1, n>>>(dx, dout, nops);
    float out[n * 2]; (void)hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0; float worst = 0;
    for (int i = 0; i < n; ++i) { float e = out[2 * i] - out[2 * i + 1]; if (e < 0) e = -e; if (e > 1e-5f) ++bad; if (e > worst) worst = e; }
    printf("nops=%d: %d / %d lanes differ, worst abs diff %g (sum ~8); lane0 %g vs %g\n", nops, bad, n, worst, out[0], out[1]);
  }
  return 0;
}
