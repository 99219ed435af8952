// developer experiment: LDS read throughput per CU of ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16 at 16 waves per CU
//   hipcc --offload-arch=gfx950 -O2 lds_rate.hip -o lds_rate && ./lds_rate
// (s_memtime counts at a fixed 100 MHz; bytes per shader clock are derived from the wall time and rocm-smi's sclk is not read: the ratio
//  between the three instructions is what matters)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  // conflict-free addresses: b128 -> lane * 16 within a 1 KB row; b64 -> lane * 8
  const unsigned a128 = (unsigned)(size_t)smem + lane * 16, a64 = (unsigned)(size_t)smem + lane * 8;
  // tr_b16: the V^T access pattern of attn_flash.hip (16-lane groups, 4 rows x 4 columns of 8 bytes)
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, hh = lane >> 5;
  const unsigned atr = (unsigned)(size_t)smem + (4 * hh + q4) * 128 + (g16 & 1) * 32 + p4 * 8 + (OP == 3 ? 0 : (q4 >> 1) * 64);   // OP 3: without the half swap of odd key pairs
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {
      u32x4 v[8];
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                   "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a128) : "memory");
      for (int i = 0; i < 8; ++i) acc ^= v[i][0];
    } else {
      u32x2 v[8];
      if (OP == 1)
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                     "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a64) : "memory");
      else if (OP >= 2)
        asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\tds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
                     "ds_read_b64_tr_b16 %4, %8 offset:4096\n\tds_read_b64_tr_b16 %5, %8 offset:5120\n\tds_read_b64_tr_b16 %6, %8 offset:6144\n\tds_read_b64_tr_b16 %7, %8 offset:7168\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(atr) : "memory");
      for (int i = 0; i < 8; ++i) acc ^= v[i][0];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  unsigned* out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  const char* names[] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16 (V^T pattern)", "ds_read_b64_tr_b16 (no half swap)"};
  const int bytes[] = {1024, 512, 512, 512};
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int op = 0; op < 4; ++op)
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 65536, 0, out, iters);
      if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 65536, 0, out, iters);
      if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 65536, 0, out, iters);
      if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 65536, 0, out, iters);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double total = 256.0 * 16 * iters * 8 * bytes[op];   // CUs x waves x iters x 8 instr x bytes
      printf("%-36s: %.3f ms  %.1f TB/s aggregate  = %.1f B/ns per CU  (%.1f B/clk at 2.0 GHz)\n", names[op], ms, total / ms / 1e9, total / 256 / (ms * 1e6), total / 256 / (ms * 1e6) / 2.0);
    }
  return 0;
}
