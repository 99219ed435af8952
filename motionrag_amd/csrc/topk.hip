// topk.hip -- retrieval: flat-scan distance + top-k over the reference-motion database (gfx950).
//
// Stands behind `table.search(vec).limit(k)[.where('video != ...')]` (lancedb 0.14.0, flat scan,
// src/data/rag.py:54; caller src/data/datamodule.py:231-236).  HBM-bound: the [N, D] fp32 database is
// streamed once per tile of 16 queries.
//
//   * lane = database row: every distance is ONE sequential fp32 fmaf chain over d = 0..D-1, so the
//     result is bit-identical to oracle/topk_oracle.c (and independent of grid shape);
//   * a 256-row x 32-float chunk of the database is staged in LDS with coalesced 16-byte loads
//     (row stride 36 floats -> conflict-free ds_read_b128 with lane = row), the 16 query vectors
//     sit in LDS and are read as wave-wide broadcasts;
//   * selection: each wavefront keeps, per query, a sorted top-64 spread over its 64 lanes.  A new
//     64-row batch is bitonic-sorted with wave shuffles and merged (elementwise min against the
//     reversed batch, then one bitonic merge); batches that cannot enter the current top-k are
//     skipped with one ballot;
//   * order is (distance asc, row asc) -> deterministic ties; excluded rows (`video != self`)
//     and padding carry distance +inf / row INT_MAX and come out as row -1.
#include "common.h"
#include "../../include/mrag_hip.h"
#include <limits.h>

namespace {

constexpr int QT = 16;         // queries per workgroup
constexpr int ROWS = 256;      // rows per workgroup iteration (4 waves x 64 lanes)
constexpr int DCH = 32;        // floats of D staged per chunk
constexpr int LDT = 36;        // LDS row stride in floats (16-byte aligned, bank-conflict-free)
constexpr int MAX_SLICES = 1024;

struct Cand { float d; int r; };

__device__ __forceinline__ bool cand_less(const Cand a, const Cand b) { return a.d < b.d || (a.d == b.d && a.r < b.r); }

__device__ __forceinline__ Cand cand_shfl_xor(const Cand c, int m) {
  Cand o; o.d = __shfl_xor(c.d, m); o.r = __shfl_xor(c.r, m); return o;
}
__device__ __forceinline__ Cand cand_shfl(const Cand c, int src) {
  Cand o; o.d = __shfl(c.d, src); o.r = __shfl(c.r, src); return o;
}

// ascending bitonic sort of one Cand per lane across the 64-lane wavefront
__device__ __forceinline__ Cand wave_sort(Cand c, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const Cand o = cand_shfl_xor(c, j);
      const bool up = (lane & k) == 0 || k == 64;
      const bool lower = (lane & j) == 0;
      const bool take_min = (lower == up);
      const bool o_less = cand_less(o, c);
      if (take_min == o_less) c = o;
    }
  }
  return c;
}

// c is bitonic across the wave -> ascending
__device__ __forceinline__ Cand wave_bitonic_merge(Cand c, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const Cand o = cand_shfl_xor(c, j);
    const bool lower = (lane & j) == 0;
    const bool o_less = cand_less(o, c);
    if (lower == o_less) c = o;
  }
  return c;
}

// run: sorted ascending top-64; batch: sorted ascending 64 new candidates -> new top-64 of the union
__device__ __forceinline__ Cand wave_merge_top(Cand run, Cand batch, int lane) {
  const Cand rev = cand_shfl(batch, 63 - lane);
  const Cand m = cand_less(rev, run) ? rev : run;  // 64 smallest of the 128, bitonic
  return wave_bitonic_merge(m, lane);
}

struct TopkP {
  const float* db; const int* group; const float* q; const int* excl;
  Cand* ws; int* out_rows; float* out_dist;
  long long n_rows; int dim, nq, k, metric, slices, rows_per_slice;
};

template <int METRIC>
__global__ __launch_bounds__(256) void topk_scan_kernel(const TopkP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = (float*)smem;                      // [QT][dim]
  float* tile = qs + QT * p.dim;                 // [ROWS][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.y * QT;
  const int slice = blockIdx.x;

  // stage the query tile (zero-fill past nq)
  for (int i = tid; i < QT * p.dim; i += 256) {
    const int qi = i / p.dim;
    qs[i] = (q0 + qi < p.nq) ? p.q[(long long)(q0 + qi) * p.dim + (i - qi * p.dim)] : 0.f;
  }
  int excl[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) excl[qi] = (p.excl && q0 + qi < p.nq) ? p.excl[q0 + qi] : INT_MIN;

  Cand run[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) { run[qi].d = INFINITY; run[qi].r = INT_MAX; }

  const long long row_begin = (long long)slice * p.rows_per_slice;
  long long row_end = row_begin + p.rows_per_slice;
  if (row_end > p.n_rows) row_end = p.n_rows;

  for (long long r0 = row_begin; r0 < row_end; r0 += ROWS) {
    float acc[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) acc[qi] = 0.f;
    for (int d0 = 0; d0 < p.dim; d0 += DCH) {
      __syncthreads();  // previous chunk consumed (also orders the query staging on the first pass)
      // coalesced stage: thread -> (row = tid/8 + 32 i, 16-byte column tid%8)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = (tid >> 3) + 32 * i;
        long long grow = r0 + rr;
        if (grow >= p.n_rows) grow = p.n_rows - 1;
        const f32x4 v = *(const f32x4*)(p.db + grow * p.dim + d0 + (tid & 7) * 4);
        *(f32x4*)(tile + rr * LDT + (tid & 7) * 4) = v;
      }
      __syncthreads();
      const float* trow = tile + tid * LDT;
#pragma unroll
      for (int c = 0; c < DCH; c += 4) {
        const f32x4 x = *(const f32x4*)(trow + c);
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
          const f32x4 qv = *(const f32x4*)(qs + qi * p.dim + d0 + c);  // wave-wide broadcast
          if constexpr (METRIC == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float df = qv[e] - x[e]; acc[qi] = __builtin_fmaf(df, df, acc[qi]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[qi] = __builtin_fmaf(qv[e], x[e], acc[qi]);
          }
        }
      }
    }
    const long long myrow = r0 + tid;
    const bool valid = myrow < row_end;
    const int grp = (valid && p.group) ? p.group[myrow] : INT_MIN + 1;
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
      Cand c;
      const float dist = METRIC == 0 ? acc[qi] : 1.0f - acc[qi];
      const bool ok = valid && !(p.group && grp == excl[qi]) && (q0 + qi < p.nq);
      c.d = ok ? dist : INFINITY;
      c.r = ok ? (int)myrow : INT_MAX;
      // skip the sort when nothing in this 64-row batch can enter the current top-k
      const Cand kth = cand_shfl(run[qi], p.k - 1);
      if (!__any(cand_less(c, kth))) continue;
      c = wave_sort(c, lane);
      run[qi] = wave_merge_top(run[qi], c, lane);
    }
  }
  // partial result of this wave: [query][part][64]
  const int part = slice * 4 + wave, nparts = p.slices * 4;
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    if (q0 + qi < p.nq) p.ws[((long long)(q0 + qi) * nparts + part) * 64 + lane] = run[qi];
  }
}

// one workgroup (4 waves) per query: merge the per-wave partial lists
__global__ __launch_bounds__(256) void topk_merge_kernel(const TopkP p) {
  __shared__ Cand sh[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  const int nparts = p.slices * 4;
  Cand run; run.d = INFINITY; run.r = INT_MAX;
  for (int part = wave; part < nparts; part += 4) {
    const Cand c = p.ws[((long long)q * nparts + part) * 64 + lane];  // already sorted ascending
    run = wave_merge_top(run, c, lane);
  }
  sh[wave][lane] = run;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) run = wave_merge_top(run, sh[w][lane], lane);
    if (lane < p.k) {
      const bool ok = run.r != INT_MAX;
      p.out_rows[(long long)q * p.k + lane] = ok ? run.r : -1;
      p.out_dist[(long long)q * p.k + lane] = run.d;
    }
  }
}

void plan(long long n_rows, int nq, int* slices, int* rows_per_slice) {
  const int ntq = (nq + QT - 1) / QT;
  long long tiles = (n_rows + ROWS - 1) / ROWS;
  long long s = MAX_SLICES / ntq;
  if (s < 1) s = 1;
  if (s > tiles) s = tiles;
  long long tps = (tiles + s - 1) / s;  // tiles per slice
  s = (tiles + tps - 1) / tps;
  *slices = (int)s;
  *rows_per_slice = (int)(tps * ROWS);
}

}  // namespace

extern "C" int64_t mrag_topk_workspace_bytes(int64_t n_rows, int32_t n_queries) {
  if (n_rows <= 0 || n_queries <= 0) return 0;
  int slices, rps;
  plan(n_rows, n_queries, &slices, &rps);
  return (int64_t)n_queries * slices * 4 * 64 * (int64_t)sizeof(Cand);
}

extern "C" int mrag_topk_f32(void* stream, const float* db, const int32_t* group, int64_t n_rows, int32_t dim, const float* queries,
                             const int32_t* exclude, int32_t n_queries, int32_t k, int32_t metric, int32_t* out_rows, float* out_dist,
                             void* workspace, int64_t workspace_bytes) {
  if (!db || !queries || !out_rows || !out_dist || !workspace) return MRAG_EINVAL;
  if (n_rows <= 0 || n_rows > INT_MAX - 1 || n_queries <= 0 || dim <= 0) return MRAG_EINVAL;
  if (k <= 0 || k > 64) return MRAG_ENOTSUP;
  if (dim % DCH != 0 || dim > 1024) return MRAG_ENOTSUP;
  if (metric != 0 && metric != 1) return MRAG_EINVAL;
  if (exclude && !group) return MRAG_EINVAL;
  if (((uintptr_t)db | (uintptr_t)queries) & 15) return MRAG_EINVAL;
  if (workspace_bytes < mrag_topk_workspace_bytes(n_rows, n_queries)) return MRAG_EINVAL;
  TopkP p{};
  p.db = db; p.group = exclude ? group : nullptr; p.q = queries; p.excl = exclude;
  p.ws = (Cand*)workspace; p.out_rows = out_rows; p.out_dist = out_dist;
  p.n_rows = n_rows; p.dim = dim; p.nq = n_queries; p.k = k; p.metric = metric;
  plan(n_rows, n_queries, &p.slices, &p.rows_per_slice);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(QT * dim + ROWS * LDT) * sizeof(float);
  auto kfn = metric == 0 ? topk_scan_kernel<0> : topk_scan_kernel<1>;
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  MRAG_LAUNCH(kfn, dim3(p.slices, (n_queries + QT - 1) / QT), dim3(256), lds, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_LAUNCH(topk_merge_kernel, dim3(n_queries), dim3(256), 0, s, p);
  MRAG_LAUNCH_CHECK();
  return MRAG_OK;
}
