// topk.hip -- retrieval: flat-scan distance + top-k over the reference-motion database (gfx950).
//
// Stands behind `table.search(vec).limit(k)[.where('video != ...')]` (lancedb 0.14.0, flat scan,
// src/data/rag.py:54; caller src/data/datamodule.py:231-236).  HBM-bound: the [N, D] fp32 database is
// streamed once per tile of 16 queries.
//
//   * 16 lanes share a database row: lane s owns one of 16 interleaved fp32 fmaf chains (16-byte loads, the 16 lanes read
//     256 contiguous bytes), a wavefront streams 4 rows per load instruction from HBM straight into registers with a
//     one-step register prefetch; the 16 partial sums fold through a fixed butterfly -> bit-identical to
//     oracle/topk_oracle.c mode 0 and independent of grid shape;
//   * the query vectors (1, 4 or 16 per workgroup pass) sit in LDS and are read 16-lane-contiguous;
//   * selection: each wavefront keeps, per query, a sorted top-64 spread over its 64 lanes.  A new
//     64-row batch is bitonic-sorted with wave shuffles and merged (elementwise min against the
//     reversed batch, then one bitonic merge); batches that cannot enter the current top-k are
//     skipped with one ballot;
//   * order is (distance asc, row asc) -> deterministic ties; excluded rows (`video != self`)
//     and padding carry distance +inf / row INT_MAX and come out as row -1;
//   * filter order (lancedb's `where(filter, prefilter=...)`): prefilter -> excluded rows never enter the selection (k results whenever k
//     rows pass); postfilter (lancedb 0.14.0's default) -> the k nearest rows are selected WITHOUT the filter, the excluded ones are then
//     dropped from that list and the survivors move up (possibly fewer than k results; the tail is row -1 / +inf).
#include "common.h"
#include "../../include/mrag_hip.h"
#include <limits.h>

namespace {

constexpr int ROWS = 256;      // rows per workgroup iteration (4 waves x 64 lanes)

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
  const int* post_group;  // postfilter: group ids consulted AFTER selection (then `group` above is null and the scan excludes nothing)
  unsigned* tickets;    // FUSED: one arrival counter per query tile (zero between calls)
  int wpb;              // waves per workgroup: always 4 (small databases spread by giving each WAVE 16 rows instead of 64: small_db())
};

// Distance of one (query, row) pair = 16 interleaved fp32 fmaf chains + a fixed 4-level pairwise tree (the definition
// oracle/topk_oracle.c mode 0 restates):
//   chain l (0..15) runs over k = 64 j + 4 l + c, j = 0.., c = 0..3, in that order;  d = tree(p[0..15]) with
//   p[l] += p[l ^ 8], then ^4, ^2, ^1 (float addition is commutative, so every lane of the butterfly holds the same bits).
// Mapping: 16 lanes share a row (lane s owns chain s: one 16-byte load per 64-float block -> the 16 lanes read 256
// contiguous bytes), a wavefront streams 4 rows per load instruction straight from HBM into registers (no LDS staging of the
// database), 16 such row-quads make the 64-row batch whose candidates sit one per lane for the bitonic selection.
// The queries (QT per workgroup pass) live in LDS and are read as 16-lane-contiguous ds_read_b128.
__device__ __forceinline__ void merge_query(const TopkP& p, int q, Cand* sh);

template <int METRIC, int QT, int JC, bool FUSED = false, int NQD = 16>
__global__ __launch_bounds__(256, QT == 16 ? 3 : 4) void topk_scan_kernel(const TopkP p) {   // <= 168 / 128 VGPRs: 3-4 waves per SIMD stream
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = (float*)smem;                      // [QT][dimp], dimp = dim rounded up to 64, zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, s = lane & 15;
  const int q0 = blockIdx.y * QT;
  const int slice = blockIdx.x;
  const int nj = (p.dim + 63) / 64, dimp = nj * 64;

  // QT == 1: qs[k].  Query tiles: query-minor image qs[(block j, lane s)][c][qi] (quad pitch QS = 4 QT + 4 floats, conflict-free for the 16
  // lanes of a row) so one ds_read_b128 returns the SAME feature of 4 queries -> packed fp32 math (v_pk_add_f32 / v_pk_fma_f32) on query pairs
  constexpr int QS = 4 * QT + 4;
  if constexpr (QT == 1) {
    for (int i = tid; i < dimp; i += blockDim.x) qs[i] = (q0 < p.nq && i < p.dim) ? p.q[(long long)q0 * p.dim + i] : 0.f;
  } else {
    for (int i = tid; i < QT * dimp; i += blockDim.x) {
      const int qi = i / dimp, k = i - qi * dimp;
      qs[(k >> 2) * QS + (k & 3) * QT + qi] = (q0 + qi < p.nq && k < p.dim) ? p.q[(long long)(q0 + qi) * p.dim + k] : 0.f;
    }
  }
  __syncthreads();
  int excl[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) excl[qi] = (p.excl && q0 + qi < p.nq) ? p.excl[q0 + qi] : INT_MIN;
  Cand run[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) { run[qi].d = INFINITY; run[qi].r = INT_MAX; }

  const long long row_begin = (long long)slice * p.rows_per_slice;
  long long row_end = row_begin + p.rows_per_slice;
  if (row_end > p.n_rows) row_end = p.n_rows;
  const int nchunk = nj / JC;                    // JC divides nj (host picks JC)
  const bool tail = (p.dim & 63) != 0;           // last 64-block is partial: lanes past the row end contribute exact zeros

  // a wave scans NQD row-quads (4 NQD rows) per pass: 64 rows, or 16 for small databases (4x the waves -> 4x the bytes in flight: a 10 k-row
  // scan is latency-bound, 157 waves with 12 KB in flight each reached 0.7 TB/s)
  for (long long r0 = row_begin + wave * (4 * NQD); r0 < row_end; r0 += (blockDim.x >> 6) * (4 * NQD)) {
    // (quad t, chunk ch) stream
    auto load = [&](int t, int ch, f32x4* dst) {
      long long row = r0 + 4 * t + g;
      if (row >= p.n_rows) row = p.n_rows - 1;
      const float* base = p.db + row * p.dim + 4 * s;
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int j = ch * JC + jj;
        if (tail && 64 * j + 4 * s >= p.dim) dst[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        else dst[jj] = __builtin_nontemporal_load((const f32x4*)(base + 64 * j));
      }
    };
    float dist[QT], acc[1] = {0.f};
    f32x2 acc2[QT / 2 > 0 ? QT / 2 : 1];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) dist[qi] = 0.f;
#pragma unroll
    for (int i = 0; i < (QT / 2 > 0 ? QT / 2 : 1); ++i) acc2[i] = f32x2{0.f, 0.f};
    int lt = 0, lch = 0;                          // next (quad, chunk) to request
    auto advance = [&]() { if (++lch == nchunk) { lch = 0; ++lt; } };
    int t = 0, ch = 0;                            // (quad, chunk) being consumed
    auto consume = [&](const f32x4* xb) {
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const f32x4 x = xb[jj];
        if constexpr (QT == 1) {
          const f32x4 qv = *(const f32x4*)(qs + (ch * JC + jj) * 64 + 4 * s);
          if constexpr (METRIC == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float df = qv[e] - x[e]; acc[0] = __builtin_fmaf(df, df, acc[0]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0] = __builtin_fmaf(qv[e], x[e], acc[0]);
          }
        } else {
          // chain order per query is unchanged (feature c = 0..3 in sequence); two queries share one packed instruction
          const float* qp = qs + ((ch * JC + jj) * 16 + s) * QS;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x2 xc = {x[c], x[c]};
#pragma unroll
            for (int qb = 0; qb < QT / 4; ++qb) {
              const f32x4 qv = *(const f32x4*)(qp + c * QT + qb * 4);
              const f32x2 qlo = {qv[0], qv[1]}, qhi = {qv[2], qv[3]};
              if constexpr (METRIC == 0) {
                const f32x2 dlo = qlo - xc, dhi = qhi - xc;
                acc2[2 * qb] = __builtin_elementwise_fma(dlo, dlo, acc2[2 * qb]);
                acc2[2 * qb + 1] = __builtin_elementwise_fma(dhi, dhi, acc2[2 * qb + 1]);
              } else {
                acc2[2 * qb] = __builtin_elementwise_fma(qlo, xc, acc2[2 * qb]);
                acc2[2 * qb + 1] = __builtin_elementwise_fma(qhi, xc, acc2[2 * qb + 1]);
              }
            }
          }
        }
      }
      if (++ch == nchunk) {                       // row-quad t finished: fixed tree over the 16 chains, lane s keeps quad s
        ch = 0;
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
          float v;
          if constexpr (QT == 1) { v = acc[0]; acc[0] = 0.f; }
          else { v = acc2[qi >> 1][qi & 1]; }
          v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
          if (t == s) dist[qi] = v;
        }
        if constexpr (QT > 1) {
#pragma unroll
          for (int i = 0; i < QT / 2; ++i) acc2[i] = f32x2{0.f, 0.f};
        }
        ++t;
      }
    };
    const int nit = NQD * nchunk;
    if constexpr (QT == 1) {
      // single query: latency-bound -> 4-deep register ring, statically indexed (step loop unrolled by 4; nit % 4 == 0)
      constexpr int PF = 4;
      f32x4 ring[PF][JC];
#pragma unroll
      for (int u = 0; u < PF - 1; ++u) {
        if (lt < NQD) { load(lt, lch, ring[u]); advance(); }
      }
      for (int it = 0; it < nit; it += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (lt < NQD) { load(lt, lch, ring[(u + PF - 1) % PF]); advance(); }
          if (it + u < nit) consume(ring[u]);         // nit = NQD * nchunk need not be a multiple of the ring depth
        }
      }
    } else {
      // query tiles: the 4 / 16 chains need the registers and the issue slots -> two steps ahead, rotated by moves
      f32x4 cur[JC], n1[JC], n2[JC];
      load(lt, lch, cur); advance();
      if (lt < NQD) { load(lt, lch, n1); advance(); }
      for (int it = 0; it < nit; ++it) {
        if (lt < NQD) { load(lt, lch, n2); advance(); }
        consume(cur);
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) { cur[jj] = n1[jj]; n1[jj] = n2[jj]; }
      }
    }
    const long long myrow = r0 + 4 * s + g;      // the row whose distance this lane captured
    const bool valid = myrow < row_end && s < NQD;
    const int grp = (valid && p.group) ? p.group[myrow] : INT_MIN + 1;
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
      Cand c;
      const float d = METRIC == 0 ? dist[qi] : 1.0f - dist[qi];
      const bool ok = valid && !(p.group && grp == excl[qi]) && (q0 + qi < p.nq);
      c.d = ok ? d : INFINITY;
      c.r = ok ? (int)myrow : INT_MAX;
      // skip the sort when nothing in this 64-row batch can enter the current top-k
      const Cand kth = cand_shfl(run[qi], p.k - 1);
      if (!__any(cand_less(c, kth))) continue;
      c = wave_sort(c, lane);
      run[qi] = wave_merge_top(run[qi], c, lane);
    }
  }
  // partial result of this wave: [query][part][64]
  const int part = slice * p.wpb + wave, nparts = p.slices * p.wpb;
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    if (q0 + qi < p.nq) p.ws[((long long)(q0 + qi) * nparts + part) * 64 + lane] = run[qi];
  }
  if constexpr (FUSED) {
    // ONE launch for the latency-bound single-query search: the workgroup that arrives LAST at this query tile's counter merges the lists.
    // Placement-independent hand-off (cdna guide, Guideline 16, counter form): plain stores -> every wave drains -> barrier -> one lane:
    // agent-scope release, asm wait, relaxed agent fetch_add; the last arriver: agent-scope acquire, wait, barrier, plain loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = (unsigned*)smem;                              // the query image is dead: LDS scratch for the flag and the merge
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(p.tickets + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned last = t == (unsigned)p.slices - 1;
      if (last) {
        __hip_atomic_store(p.tickets + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *flag = last;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();
    if (!last) return;
    for (int qi = 0; qi < QT; ++qi)
      if (q0 + qi < p.nq) merge_query(p, q0 + qi, (Cand*)smem);
  }
}

// one workgroup (4 waves) per query: merge the per-wave partial lists (each sorted ascending, 64 entries).
// Phase A bounds the answer: the k-th smallest of the lists' MINIMA is an upper bound of the final k-th distance, so only lists
// whose minimum does not exceed it can contribute (about k of thousands).  Phase B merges just those.  The result is the
// unique top-k under the total order (distance, row), whatever the merge order.
// `sh` = 4 x 64 candidates + 1 of LDS scratch.
__device__ __forceinline__ void merge_query(const TopkP& p, int q, Cand* sh) {
  Cand* thr_s = sh + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int nparts = p.slices * p.wpb;
  const Cand* lists = p.ws + (long long)q * nparts * 64;
  Cand inf; inf.d = INFINITY; inf.r = INT_MAX;
  Cand best = inf;
  for (int base = wave * 64; base < nparts; base += nw * 64) {
    const int part = base + lane;
    Cand m = part < nparts ? lists[(long long)part * 64] : inf;
    m = wave_sort(m, lane);
    best = wave_merge_top(best, m, lane);
  }
  sh[wave * 64 + lane] = best;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < nw; ++w) best = wave_merge_top(best, sh[w * 64 + lane], lane);
    if (lane == p.k - 1) *thr_s = best;           // k-th smallest minimum
  }
  __syncthreads();
  const Cand thr = *thr_s;
  Cand run = inf;
  for (int base = wave * 64; base < nparts; base += nw * 64) {
    const int part = base + lane;
    const Cand m = part < nparts ? lists[(long long)part * 64] : inf;
    unsigned long long todo = __ballot(part < nparts && !cand_less(thr, m));   // min <= thr
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const Cand c = lists[(long long)(base + src) * 64 + lane];
      run = wave_merge_top(run, c, lane);
    }
  }
  __syncthreads();
  sh[wave * 64 + lane] = run;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < nw; ++w) run = wave_merge_top(run, sh[w * 64 + lane], lane);
    if (p.post_group) {
      // lancedb's postfilter: the k nearest are final; rows of the excluded group leave the list, the rest keep their order and move up
      const bool ok = lane < p.k && run.r != INT_MAX;
      const bool keep = ok && p.post_group[ok ? run.r : 0] != p.excl[q];
      const unsigned long long m = __ballot(keep);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      const int kept = __popcll(m);
      if (keep) {
        p.out_rows[(long long)q * p.k + pos] = run.r;
        p.out_dist[(long long)q * p.k + pos] = run.d;
      }
      if (lane >= kept && lane < p.k) {
        p.out_rows[(long long)q * p.k + lane] = -1;
        p.out_dist[(long long)q * p.k + lane] = INFINITY;
      }
    } else if (lane < p.k) {
      const bool ok = run.r != INT_MAX;
      p.out_rows[(long long)q * p.k + lane] = ok ? run.r : -1;
      p.out_dist[(long long)q * p.k + lane] = run.d;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const TopkP p) {
  __shared__ Cand sh[257];
  merge_query(p, blockIdx.x, sh);
}

inline int pick_qt(int nq) { return nq >= 9 ? 16 : nq >= 2 ? 4 : 1; }   // queries per workgroup pass

inline bool small_db(long long n_rows, int nq) { return nq <= 4 && n_rows < 256LL * ROWS; }   // < 65 536 rows, <= 4 queries: 16 rows per wave

void plan(long long n_rows, int nq, int* slices, int* rows_per_slice) {
  const int QT = pick_qt(nq);
  const int ntq = (nq + QT - 1) / QT;
  // workgroups along the database.  A single query (QT = 1) streams best with 512: two workgroups per CU, ~2 000 rows each at 10^6 rows, and only
  // 2 048 per-wave lists for the last arriver to merge -- 523 us = 5.88 TB/s at 10^6 rows against 627 us with 2 048 slices, 6.60 against 6.23 TB/s
  // at 4 x 10^6 (256: 4.1 TB/s, 384: 5.2, 768: 5.8; round 4, tools/topk_slices_probe.py).  Query tiles keep 2 048 (1 024 costs them 15 %).
#ifdef MRAG_TOPK_MAX_SLICES
  const int MAX_SLICES = MRAG_TOPK_MAX_SLICES;
#else
  const int MAX_SLICES = QT == 1 ? 512 : 2048;
#endif
  const int ROWS = small_db(n_rows, nq) ? 64 : 256;     // rows per workgroup pass (4 waves x 16 or 64 rows; 8 rows per wave measured slower:
                                                        // 34.0 vs 27.5 us at 10 k rows -- the 1 256-list merge of the last arriver then dominates)
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

constexpr int64_t kTicketBytes = 64;     // arrival counters of the fused single-launch form (<= 4 queries = 1 query tile ... 4 tiles of QT = 1)

extern "C" int64_t mrag_topk_workspace_bytes(int64_t n_rows, int32_t n_queries) {
  if (n_rows <= 0 || n_queries <= 0) return 0;
  int slices, rps;
  plan(n_rows, n_queries, &slices, &rps);
  return kTicketBytes + (int64_t)n_queries * slices * 4 * 64 * (int64_t)sizeof(Cand);
}

extern "C" int mrag_topk_f32(void* stream, const float* db, const int32_t* group, int64_t n_rows, int32_t dim, const float* queries,
                             const int32_t* exclude, int32_t n_queries, int32_t k, int32_t metric, int32_t* out_rows, float* out_dist,
                             void* workspace, int64_t workspace_bytes, int32_t postfilter) {
  if (!db || !queries || !out_rows || !out_dist || !workspace) return MRAG_EINVAL;
  if (n_rows <= 0 || n_rows > INT_MAX - 1 || n_queries <= 0 || dim <= 0) return MRAG_EINVAL;
  if (k <= 0 || k > 64) return MRAG_ENOTSUP;
  if (dim % 4 != 0 || dim > 1024) return MRAG_ENOTSUP;
  if (metric != 0 && metric != 1) return MRAG_EINVAL;
  if (exclude && !group) return MRAG_EINVAL;
  if (((uintptr_t)db | (uintptr_t)queries) & 15) return MRAG_EINVAL;
  if (workspace_bytes < mrag_topk_workspace_bytes(n_rows, n_queries)) return MRAG_EINVAL;
  TopkP p{};
  if (postfilter != 0 && postfilter != 1) return MRAG_EINVAL;
  const bool post = postfilter && exclude;
  p.db = db; p.group = (exclude && !post) ? group : nullptr; p.post_group = post ? group : nullptr; p.q = queries; p.excl = exclude;
  p.tickets = (unsigned*)workspace; p.ws = (Cand*)((char*)workspace + kTicketBytes); p.out_rows = out_rows; p.out_dist = out_dist;
  p.n_rows = n_rows; p.dim = dim; p.nq = n_queries; p.k = k; p.metric = metric;
  plan(n_rows, n_queries, &p.slices, &p.rows_per_slice);
  p.wpb = 4;
  const bool small = small_db(n_rows, n_queries);
  hipStream_t s = (hipStream_t)stream;
  const int QT = pick_qt(n_queries), nj = (dim + 63) / 64;
  // blocks of 64 floats per register-ring step: 4 for the single query, 2 for query tiles (their chains need the registers)
  const int JCsel = QT == 1 ? (nj % 4 == 0 ? 4 : 1) : (nj % 2 == 0 ? 2 : 1);
  size_t lds = (QT == 1 ? (size_t)nj * 64 : (size_t)nj * 16 * (4 * QT + 4)) * sizeof(float);
  const dim3 grid(p.slices, (n_queries + QT - 1) / QT), block(256);
  // <= 4 queries (the interactive search of rag.py:63-80): ONE launch, the last workgroup to arrive merges (needs the first 64 workspace
  // bytes ZERO on entry -- see the header; the kernel leaves them zero)
  const bool fused = n_queries <= 4 && ((uintptr_t)workspace & 15) == 0;
  if (small && !fused) return MRAG_EINVAL;                                          // the small-database plan exists only in the fused form
  if (fused && lds < 257 * sizeof(Cand)) lds = 257 * sizeof(Cand);
#define MRAG_TOPK_FUSED(M, Q, J)                                                                                  \
  if (fused && metric == M && QT == Q && JCsel == J) {                                                            \
    auto kfn = small ? topk_scan_kernel<M, Q, J, true, 4> : topk_scan_kernel<M, Q, J, true, 16>;                  \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    if (e != hipSuccess) return (int)e;                                                                           \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                                     \
    MRAG_LAUNCH_CHECK();                                                                                          \
    MRAG_COUNT(MRAG_K_TOPK_SCAN_FUSED_MERGE);                                                                     \
    return MRAG_OK;                                                                                               \
  }
  MRAG_TOPK_FUSED(0, 1, 1) MRAG_TOPK_FUSED(0, 1, 4) MRAG_TOPK_FUSED(0, 4, 1) MRAG_TOPK_FUSED(0, 4, 2)
  MRAG_TOPK_FUSED(1, 1, 1) MRAG_TOPK_FUSED(1, 1, 4) MRAG_TOPK_FUSED(1, 4, 1) MRAG_TOPK_FUSED(1, 4, 2)
#undef MRAG_TOPK_FUSED
#define MRAG_TOPK_CASE(M, Q, J)                                                                                   \
  if (metric == M && QT == Q && JCsel == J) {                                                                     \
    auto kfn = topk_scan_kernel<M, Q, J>;                                                                         \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    if (e != hipSuccess) return (int)e;                                                                           \
    MRAG_LAUNCH(kfn, grid, block, lds, s, p);                                                                     \
  }
  MRAG_TOPK_CASE(0, 1, 1) MRAG_TOPK_CASE(0, 1, 4) MRAG_TOPK_CASE(0, 4, 1) MRAG_TOPK_CASE(0, 4, 2) MRAG_TOPK_CASE(0, 16, 1) MRAG_TOPK_CASE(0, 16, 2)
  MRAG_TOPK_CASE(1, 1, 1) MRAG_TOPK_CASE(1, 1, 4) MRAG_TOPK_CASE(1, 4, 1) MRAG_TOPK_CASE(1, 4, 2) MRAG_TOPK_CASE(1, 16, 1) MRAG_TOPK_CASE(1, 16, 2)
#undef MRAG_TOPK_CASE
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_TOPK_SCAN);
  MRAG_LAUNCH(topk_merge_kernel, dim3(n_queries), dim3(256), 0, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_TOPK_MERGE);
  return MRAG_OK;
}
