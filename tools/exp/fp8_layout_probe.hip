// fp8_layout_probe.hip -- developer probe: which (lane, byte) of the A / B operand registers holds which (row, k) / (k, col) element of the
// gfx950 fp8 MFMAs (non-scaled 16x16x32 / 32x32x16 and block-scaled 32x32x64 / 16x16x128)?  The guide gives the bf16 maps only and says to check
// other dtypes with exact data.  Method: A = one-hot at (lane L, byte j), B = bit p of the B position's index as 0.0 / 1.0; the result D[row][col]
// over all p spells, for every column, the index of the B position that multiplies A's one-hot element, i.e. the B position with the same k.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/fp8_layout_probe.hip -o tools/exp/fp8_layout_probe && tools/exp/fp8_layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr unsigned char ONE = 0x38;   // 1.0 in OCP e4m3

// KIND 0: 16x16x32 fp8 (8 B / lane), 1: 32x32x16 fp8 (8 B), 2: scaled 32x32x64 (32 B), 3: scaled 16x16x128 (32 B)
template <int KIND>
__global__ void probe(int* out_row, int* out_bpos) {
  constexpr int NB = KIND < 2 ? 8 : 32;            // operand bytes per lane
  constexpr int NC = (KIND == 0 || KIND == 3) ? 16 : 32;   // rows = cols of D
  constexpr int NR = (KIND == 0 || KIND == 3) ? 4 : 16;    // D registers per lane
  constexpr int BITS = KIND < 2 ? 9 : 11;          // 64 * NB positions
  const int lane = threadIdx.x;
  const int apos = blockIdx.x;                     // A one-hot position: lane aL, byte aj
  const int aL = apos / NB, aj = apos % NB;
  unsigned char abytes[32] = {0}, bbytes[32];
  if (lane == aL) abytes[aj] = ONE;
  float dsum[NR][BITS + 1];
  for (int p = 0; p <= BITS; ++p) {
    for (int j = 0; j < NB; ++j) {
      const int idx = lane * NB + j;
      bbytes[j] = (p == BITS) ? ONE : (((idx >> p) & 1) ? ONE : 0);
    }
    if constexpr (KIND == 0) {
      f32x4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*(long*)abytes, *(long*)bbytes, c, 0, 0, 0);
      for (int r = 0; r < NR; ++r) dsum[r][p] = c[r];
    } else if constexpr (KIND == 1) {
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(*(long*)abytes, *(long*)bbytes, c, 0, 0, 0);
      for (int r = 0; r < NR; ++r) dsum[r][p] = c[r];
    } else if constexpr (KIND == 2) {
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(*(i32x8*)abytes, *(i32x8*)bbytes, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      for (int r = 0; r < NR; ++r) dsum[r][p] = c[r];
    } else {
      f32x4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(*(i32x8*)abytes, *(i32x8*)bbytes, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      for (int r = 0; r < NR; ++r) dsum[r][p] = c[r];
    }
  }
  // D layout (dtype-independent per the guide): 16x16: col = lane & 15, row = 4 (lane >> 4) + r; 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  for (int r = 0; r < NR; ++r) {
    if (dsum[r][BITS] != 0.f) {                    // all-ones B: this (row, col) sees A's one-hot
      const int row = NC == 16 ? 4 * (lane >> 4) + r : (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int col = NC == 16 ? (lane & 15) : (lane & 31);
      int idx = 0;
      for (int p = 0; p < BITS; ++p) idx |= (dsum[r][p] != 0.f ? 1 : 0) << p;
      out_row[apos] = row;
      out_bpos[apos * NC + col] = idx;
    }
  }
}

template <int KIND>
void run(const char* name) {
  constexpr int NB = KIND < 2 ? 8 : 32;
  constexpr int NC = (KIND == 0 || KIND == 3) ? 16 : 32;
  const int npos = 64 * NB;
  int *d_row, *d_b;
  hipMalloc(&d_row, npos * sizeof(int));
  hipMalloc(&d_b, npos * NC * sizeof(int));
  hipMemset(d_row, 0xff, npos * sizeof(int));
  hipMemset(d_b, 0xff, npos * NC * sizeof(int));
  hipLaunchKernelGGL(probe<KIND>, dim3(npos), dim3(64), 0, 0, d_row, d_b);
  hipDeviceSynchronize();
  std::vector<int> row(npos), bp(npos * NC);
  hipMemcpy(row.data(), d_row, npos * sizeof(int), hipMemcpyDeviceToHost);
  hipMemcpy(bp.data(), d_b, npos * NC * sizeof(int), hipMemcpyDeviceToHost);
  printf("=== %s: A operand %d bytes/lane, D %dx%d\n", name, NB, NC, NC);
  // hypothesis H: A[row = L % NC][k = NB' * (L / NC) + j]-style maps.  Print the table compactly and test closed forms.
  // (1) row of A position
  int bad_row = 0;
  for (int a = 0; a < npos; ++a) if (row[a] != (a / NB) % NC) ++bad_row;
  printf("row(L, j) == L %% %d for all positions: %s (%d mismatches)\n", NC, bad_row ? "NO" : "yes", bad_row);
  // (2) B position matched to A position (L, j) at column c: is it lane' = c + NC * g', byte j' with (g', j') == (L / NC, j)?
  int same = 0, tot = 0;
  for (int a = 0; a < npos; ++a)
    for (int c = 0; c < NC; ++c) {
      const int b = bp[a * NC + c];
      if (b < 0) continue;
      ++tot;
      const int Lb = b / NB, jb = b % NB;
      if (Lb % NC == c && Lb / NC == (a / NB) / NC && jb == a % NB) ++same;
    }
  printf("B position for A(L, j) at column c is (lane c + %d * (L / %d), byte j): %d of %d\n", NC, NC, same, tot);
  // (3) the raw map for a few lanes so that a k numbering can be read off: A(L, j) -> B(lane', byte') at column 0
  const int lanes[] = {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63};
  for (int L : lanes) {
    printf("A lane %2d row %2d | B(lane,byte) at col 0 for j=0..%d:", L, row[L * NB], NB - 1);
    for (int j = 0; j < NB; ++j) {
      const int b = bp[(L * NB + j) * NC + 0];
      printf(" (%d,%d)", b < 0 ? -1 : b / NB, b < 0 ? -1 : b % NB);
    }
    printf("\n");
  }
  hipFree(d_row); hipFree(d_b);
}

// scale operand check for the block-scaled forms: scale byte selection and which lanes' scales apply.  A = all ones (K ones per row), B = all ones:
// D = K everywhere with scale 127 (2^0).  Then lane L's A-scale byte 0 = 128 (2^1): which rows double?
__global__ void scale_probe(float* out) {   // out[64 lanes][64 rows... ] small: report per changed lane the set of rows that changed
  const int lane = threadIdx.x;
  const int L = blockIdx.x;                  // the lane whose A scale is 2^1
  unsigned char ones[32];
  for (int j = 0; j < 32; ++j) ones[j] = ONE;
  const int sa = (lane == L) ? 0x7f7f7f80 : 0x7f7f7f7f;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(*(i32x8*)ones, *(i32x8*)ones, c, 0, 0, 0, sa, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    if (col == 0) out[L * 32 + row] = c[r];
  }
}

int main() {
  run<0>("v_mfma_f32_16x16x32_fp8_fp8");
  run<1>("v_mfma_f32_32x32x16_fp8_fp8");
  run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8, scales 2^0)");
  run<3>("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8, scales 2^0)");
  float* d;
  hipMalloc(&d, 64 * 32 * sizeof(float));
  hipLaunchKernelGGL(scale_probe, dim3(64), dim3(64), 0, 0, d);
  hipDeviceSynchronize();
  std::vector<float> h(64 * 32);
  hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  printf("=== scaled 32x32x64: A-scale byte 0 of lane L set to 2^1 (others 2^0), all-ones operands: D[row][0] per L (64 = unscaled)\n");
  for (int L : {0, 1, 31, 32, 33, 63}) {
    printf("L=%2d:", L);
    for (int r = 0; r < 32; ++r) printf(" %g", h[L * 32 + r]);
    printf("\n");
  }
  return 0;
}
