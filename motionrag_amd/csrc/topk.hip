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
#include <type_traits>

namespace {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

constexpr int ROWS = 256;      // rows per workgroup iteration (4 waves x 64 lanes)

struct Cand { float d; int r; };

__device__ __forceinline__ bool cand_less(const Cand a, const Cand b) { return a.d < b.d || (a.d == b.d && a.r < b.r); }

// rank of this lane's candidate among the candidates of lanes 0 .. n - 1 (n wave-uniform): the number of them that come before it in the strict order
// (distance, row).  Lane j's candidate reaches every lane through v_readlane (an SGPR broadcast) as ONE 64-bit key -- the order-preserving image of the
// distance (float_key below; -0 counted as +0, as cand_less does) over the row -- so a step is two readlanes, one 64-bit compare and an add: n short steps
// without memory.  (A 64-lane bitonic network is 21 dependent shuffle stages; counting over an LDS array exposed one LDS round trip per element.)
__device__ __forceinline__ int cand_rank(const Cand c, const int n) {
  const unsigned u = __float_as_uint(c.d + 0.0f);
  const unsigned long long kc = ((unsigned long long)((u & 0x80000000u) ? ~u : (u | 0x80000000u)) << 32) | (unsigned)c.r;
  const int klo = (int)(unsigned)kc, khi = (int)(unsigned)(kc >> 32);
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    const unsigned long long ko = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(khi, j) << 32) | (unsigned)__builtin_amdgcn_readlane(klo, j);
    rank += ko < kc ? 1 : 0;
  }
  return rank;
}

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
  int rescore;          // merge (LIST = 16, metric l2 of the fan-out form): the 16 nearest candidates are re-scored with the direct sum of (q - x)^2 and re-ranked
};

// Distance of one (query, row) pair = 16 interleaved fp32 fmaf chains + a fixed 4-level pairwise tree (the definition
// oracle/topk_oracle.c mode 0 restates):
//   chain l (0..15) runs over k = 64 j + 4 l + c, j = 0.., c = 0..3, in that order;  d = tree(p[0..15]) with
//   p[l] += p[l ^ 8], then ^4, ^2, ^1 (float addition is commutative, so every lane of the butterfly holds the same bits).
// Mapping: 16 lanes share a row (lane s owns chain s: one 16-byte load per 64-float block -> the 16 lanes read 256
// contiguous bytes), a wavefront streams 4 rows per load instruction straight from HBM into registers (no LDS staging of the
// database), 16 such row-quads make the 64-row batch whose candidates sit one per lane for the bitonic selection.
// The queries (QT per workgroup pass) live in LDS and are read as 16-lane-contiguous ds_read_b128.
template <int LIST = 64>
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
  if constexpr (FUSED) {
    // the four waves' lists meet in LDS first: ONE list per workgroup leaves (p.wpb = 1), so the last arriver bounds and merges a quarter of the lists
    // (10 000 rows: 157 instead of 628; 10^6 rows: 512 instead of 2 048 -- the merge was half of the 33 us of a 10 000-row search)
    __syncthreads();                                               // the query image is dead: LDS scratch
    Cand* pm = (Cand*)smem;                                        // [QT][3][64]
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
      if (wave > 0) pm[(qi * 3 + wave - 1) * 64 + lane] = run[qi];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int qi = 0; qi < QT; ++qi) {
#pragma unroll
        for (int w = 0; w < 3; ++w) run[qi] = wave_merge_top(run[qi], pm[(qi * 3 + w) * 64 + lane], lane);
        if (q0 + qi < p.nq) p.ws[((long long)(q0 + qi) * p.slices + slice) * 64 + lane] = run[qi];
      }
    }
  } else {
    // partial result of this wave: [query][part][64]
    const int part = slice * p.wpb + wave, nparts = p.slices * p.wpb;
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
      if (q0 + qi < p.nq) p.ws[((long long)(q0 + qi) * nparts + part) * 64 + lane] = run[qi];
    }
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
      if (q0 + qi < p.nq) merge_query<64>(p, q0 + qi, (Cand*)smem);
  }
}

// one workgroup (4 waves) per query: merge the per-wave partial lists (each sorted ascending, 64 entries).
// Phase A bounds the answer: the k-th smallest of the lists' MINIMA is an upper bound of the final k-th distance, so only lists
// whose minimum does not exceed it can contribute (about k of thousands).  Phase B merges just those.  The result is the
// unique top-k under the total order (distance, row), whatever the merge order.
// `sh` = 4 x 64 candidates + 1 of LDS scratch.
// LIST = entries per partial list: 64 (the scan kernel's per-wave lists) or 16 (the fan-out kernel's per-workgroup lists; lanes >= 16 read +inf)
// the tail of every merge: `run` = the sorted nearest candidates of query q across the lanes of the group's first wave (lt < 64) -> second scoring
// (p.rescore), filter order, output.  lt = thread index inside the 256-thread group that merges this query (the merge kernels: threadIdx.x; the
// one-launch fan-out form runs two groups per 512-thread workgroup), sh = the group's 257-candidate scratch, store = false: a group without a
// query of its own walks through the barriers only.
__device__ __forceinline__ void finish_query(const TopkP& p, const int q, Cand run, Cand* sh, const int lt, const bool store) {
  const int lane = lt & 63, wave = lt >> 6;
  Cand inf; inf.d = INFINITY; inf.r = INT_MAX;
  if (p.rescore) {
    // The fan-out form scores "l2" through |q|^2 + |x|^2 - 2 q.x: good for SELECTING neighbours, but its absolute error is an ulp of |q|^2 + |x|^2
    // (~1e-4 on unnormalised 768-d embeddings) -- a row's distance to itself came out as 1e-3, near-duplicates as noise or negative, and `_distance`
    // feeds condition_fusion's weights (src/projects/condition/utils.py:7-36).  So the 16 nearest candidates under that score are scored
    // AGAIN here with the scan kernel's definition -- the direct sum of (q - x)^2 on 16 interleaved fmaf chains + the fixed tree, oracle mode 0 -- and
    // re-ranked: the distances a row gets no longer depend on the call shape, and ranks among near neighbours follow the exact form.  16 lanes per
    // candidate, 16 candidates per pass of the group.
    __syncthreads();
    if (wave == 0) sh[lane] = run;
    __syncthreads();
    const int c = lt >> 4, s16 = lt & 15;                   // (256 threads: c < 16)
    const Cand cc = sh[c];
    float acc = 0.f;
    if (cc.r != INT_MAX) {
      const float* x = p.db + (long long)cc.r * p.dim;
      const float* qv = p.q + (long long)q * p.dim;
      // chain s16: features 64 j + 4 s16 + {0, 1, 2, 3} (dim % 4 == 0), j ascending.  The loads of twelve steps (all of a 768-d row) are issued together and the
      // chain then runs over registers: as a plain loop every step waited for its own two loads -- twelve memory round trips per candidate, ~10 us of the merge.
      // (A step past the row loads nothing and adds fmaf(0, 0, acc) = acc: the sum of squares is never -0.)
      constexpr int UB = 12;
      for (int kb = 4 * s16; kb < p.dim; kb += 64 * UB) {
        float4 xv[UB], q4[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int k0 = kb + 64 * u;
          const bool in = k0 < p.dim;
          xv[u] = in ? *(const float4*)(x + k0) : float4{0.f, 0.f, 0.f, 0.f};
          q4[u] = in ? *(const float4*)(qv + k0) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          float df = q4[u].x - xv[u].x; acc = __builtin_fmaf(df, df, acc);
          df = q4[u].y - xv[u].y; acc = __builtin_fmaf(df, df, acc);
          df = q4[u].z - xv[u].z; acc = __builtin_fmaf(df, df, acc);
          df = q4[u].w - xv[u].w; acc = __builtin_fmaf(df, df, acc);
        }
      }
    }
    acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);   // p[l] += p[l ^ 8], ^ 4, ^ 2, ^ 1
    __syncthreads();
    if (s16 == 0 && cc.r != INT_MAX) sh[c].d = acc;
    __syncthreads();
    if (wave == 0) {                                        // re-rank the (up to) 16 by counting: sh[0..15] -> sh[64..79]
      const Cand c = lane < 16 ? sh[lane] : inf;
      const bool have = c.r != INT_MAX;
      const int rank = cand_rank(c, 16);
      if (lane < 16) sh[64 + lane] = inf;
      __builtin_amdgcn_wave_barrier();
      if (have) sh[64 + rank] = c;
      __builtin_amdgcn_wave_barrier();
      run = lane < 16 ? sh[64 + lane] : inf;
    }
  }
  if (wave == 0 && store) {
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

template <int LIST>
__device__ __forceinline__ void merge_query(const TopkP& p, int q, Cand* sh) {
  Cand* thr_s = sh + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int nparts = p.slices * p.wpb;
  const Cand* lists = p.ws + (long long)q * nparts * LIST;
  Cand inf; inf.d = INFINITY; inf.r = INT_MAX;
  Cand best = inf;
  for (int base = wave * 64; base < nparts; base += nw * 64) {
    const int part = base + lane;
    Cand m = part < nparts ? lists[(long long)part * LIST] : inf;
    m = wave_sort(m, lane);
    best = wave_merge_top(best, m, lane);
  }
  sh[wave * 64 + lane] = best;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < nw; ++w) best = wave_merge_top(best, sh[w * 64 + lane], lane);
    if (lane == (p.rescore ? 16 : p.k) - 1) *thr_s = best;   // k-th smallest minimum (the 16th when a second scoring follows: it takes the 16 nearest)
  }
  __syncthreads();
  const Cand thr = *thr_s;
  Cand run = inf;
  for (int base = wave * 64; base < nparts; base += nw * 64) {
    const int part = base + lane;
    const Cand m = part < nparts ? lists[(long long)part * LIST] : inf;
    unsigned long long todo = __ballot(part < nparts && !cand_less(thr, m));   // min <= thr
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const Cand c = lane < LIST ? lists[(long long)(base + src) * LIST + lane] : inf;
      run = wave_merge_top(run, c, lane);
    }
  }
  __syncthreads();
  sh[wave * 64 + lane] = run;
  __syncthreads();
  if (wave == 0)
    for (int w = 1; w < nw; ++w) run = wave_merge_top(run, sh[w * 64 + lane], lane);
  finish_query(p, q, run, sh, tid, true);
}

template <int LIST>
__global__ __launch_bounds__(256) void topk_merge_kernel(const TopkP p) {
  __shared__ Cand sh[257];
  merge_query<LIST>(p, blockIdx.x, sh);
}


// ---------------------------------------------------------------------------------------------- fan-out form: >= 16 queries per call
// The caller that builds the retrieval tables searches in batches (src/data/datamodule.py:231-236 issues one query per annotation; attach_ref_videos
// batches 256): the scan kernel above re-streams the database once per 16 queries and spends its time in fp32 vector FMAs (N = 10 k, Q = 256: 16 passes,
// 246 us).  Here the batch is an fp32 MATRIX product on v_mfma_f32_32x32x2_f32 -- 64 FLOP per cycle and SIMD, the vector unit's peak rate with one VGPR per
// operand and the VALU left free -- and the database is streamed ONCE per 256 queries.  The instruction's result is bit for bit a k-ordered fmaf chain
// (MI355X guide, "FP32-input MFMA"), so the distances are DEFINED: one chain per (query, row) in the feature order 8c, 8c+4, 8c+1, 8c+5, ... (an MFMA takes
// feature k from lanes 0-31 and k' from lanes 32-63; a lane's four MFMAs of a 32-byte block use the four floats of ONE ds_read_b128), squared norms as two
// chains (oracle/topk_oracle.c mode 2 restates it; bit-exact tests).  L2 goes through |q|^2 + |x|^2 - 2 q.x.
//   * workgroup = 4 waves x 32 database rows, every wave against the workgroup's 32 TN queries (TN = 8 / 4 / 2 / 1 tiles of 32: the plan takes the largest
//     TN that still gives the chip >= 256 workgroups, so a 10 k-row table is cut along the QUERIES as well -- grid.y -- instead of leaving CUs idle);
//   * both operands ride the LDS: 32-feature slabs (128-byte rows, 16-byte chunks XOR-swizzled by row & 7 on the DMA's source side) in a ring of NST stages
//     (2 at TN = 8, where a slab is 8 192 MFMA cycles per wave; 3-4 at the narrow tiles, whose 1-2 k cycles per slab are shorter than one DMA round trip):
//     the LDS-DMA of slab i + NST - 1 is issued under the MFMAs of slab i behind a COUNTED vmcnt wait; the stream runs across the row blocks of a workgroup;
//   * accumulator layout: lane (n = lane & 31, h = lane >> 5) holds query n of a 32-query tile against rows (reg & 3) + 8 (reg >> 2) + 4 h: after a row block a
//     lane tests its 16 rows against ITS query's current k-th distance (a register).  Survivors are rare once the lists are warm; they go, one per lane and
//     round, through per-query slots to the query's OWNER thread, which inserts them into the workgroup's sorted top-16 list in LDS and republishes the
//     threshold -- best candidates first, so the thresholds of a cold list converge in about k rounds;
//   * per-workgroup lists [query][part][16] leave through the workspace and the merge kernel (LIST = 16) finishes, filter order included.
__device__ __attribute__((aligned(128))) float g_topk_zero[32];    // source of feature chunks past `dim` (never written)

#ifdef MRAG_TOPK_DIAG_STATS   // developer timing build: [0] selection rounds, [1] cycles inside the selection, [2] cycles in the whole kernel, [3] row blocks (summed over workgroups)
__device__ unsigned long long g_topk_dbg[4];
extern "C" int mrag_debug_topk_stats(unsigned long long* out_host, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_topk_dbg), sizeof(g_topk_dbg));
  if (e == hipSuccess && reset) { unsigned long long z[4] = {0, 0, 0, 0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_topk_dbg), z, sizeof(z)); }
  return (int)e;
}
#define MRAG_TSTAMP(T) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T)::"memory")
#endif
struct TopkMP {
  const float* db; const int* group; const float* q; const int* excl; const float* qq;
  unsigned* tau_g;        // [nq] the best k-th distance any workgroup has published for the query, as an order-preserving unsigned key (atomicMin)
  Cand* lists;
  long long n_rows; int dim, nq, k, nparts, rows_per_part, nslab;
};

// order-preserving map float -> unsigned (a < b  <=>  key(a) < key(b), -0 < +0): the shared thresholds are lowered with atomicMin
__device__ __forceinline__ unsigned float_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// per call: the shared thresholds start at +inf; |q|^2 of every query in the fan-out kernel's order (chains over features
// 8c + t and 8c + 4 + t, added once) when the metric needs it.  One wave per query: the row comes into LDS with coalesced 16-byte loads, then lane 0 runs the
// `lo` chain and lane 1 the `hi` chain over ds_read_b128 quads -- the chains are sequential by definition (dim / 2 dependent FMAs each), the loads need not be
// (a thread per query, the first form, took 91 us at 256 queries x 768: a serial walk over a 3 KB-strided row)
__global__ __launch_bounds__(64) void topk_qq_kernel(const float* q, float* qq, unsigned* tau_g, int nq, int dim, int want_qq) {
  __shared__ __attribute__((aligned(16))) float row[1024];   // dim <= 1024 (mrag_topk_f32)
  const int i = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) tau_g[i] = 0xff800000u;             // float_key(+inf): no threshold yet
  if (!want_qq) return;
  const float4* src = (const float4*)(q + (long long)i * dim);
  const int nquad = dim >> 2;                        // dim % 4 == 0 (mrag_topk_f32)
  for (int c = lane; c < nquad; c += 64) ((float4*)row)[c] = src[c];
  __syncthreads();
  if (lane < 2) {                                    // lane 0: quads 0, 2, 4, .. (features 8c + t); lane 1: quads 1, 3, 5, .. (8c + 4 + t; none for the last block of an odd quad count)
    float acc = 0.f;
    for (int c = lane; c < nquad; c += 2) {
      const float4 v = ((const float4*)row)[c];
      acc = __builtin_fmaf(v.x, v.x, acc); acc = __builtin_fmaf(v.y, v.y, acc); acc = __builtin_fmaf(v.z, v.z, acc); acc = __builtin_fmaf(v.w, v.w, acc);
    }
    const float other = __shfl_xor(acc, 1);
    if (lane == 0) qq[i] = acc + other;              // lo + hi
  }
}

constexpr int mfma_stages(int TN) { return TN >= 8 ? 2 : TN == 4 ? 3 : TN == 2 ? 4 : 3; }   // TN = 1: three stages keep two workgroups per CU (67 KB each)

// WN = 2: eight waves -- four row groups x two query groups of TN tiles each -- so every SIMD holds TWO waves and one's LDS-read / DMA-issue / barrier
// stalls pass under the other's MFMAs (the 256-query workgroup: TN = 4, WN = 2; as four waves of TN = 8 its matrix pipe idled a quarter of the stream)
template <int METRIC, int TN, int WN>
__global__ __launch_bounds__(256 * WN) void topk_mfma_kernel(const TopkMP p) {
  constexpr int WM = 4, NW = WM * WN, NT = 64 * NW, RB = 32 * WM, QB = 32 * TN * WN, NST = mfma_stages(TN * WN);
  constexpr int STAGE = (RB + QB) * 128, NPIECE = (RB + QB) / 8, PPW = NPIECE / NW, NTAB = (RB / 8) / NW, LSTR = 17, NSLOT = 2 * WM;
  static_assert(NPIECE % NW == 0 && (RB / 8) % NW == 0 && PPW == NTAB + TN, "every wave issues the same number of LDS-DMA pieces per slab (the counted vmcnt wait relies on it)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Cand* lists = (Cand*)(smem + NST * STAGE);        // [QB][LSTR]: sorted ascending, entries >= k stay +inf
  Cand* slots = lists + QB * LSTR;                  // [QB][NSLOT]: this round's candidate of each (wave, half) for the query
  Cand* taus = slots + QB * NSLOT;                  // [QB]: the query's k-th best so far
  float* xxs = (float*)(taus + QB);                 // [4][32]: |x|^2 of the wave's 32 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & (WM - 1), wn = wave / WM;
  const int r32 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * QB, part = blockIdx.x;
  const long long row_begin = (long long)part * p.rows_per_part;
  long long row_end = row_begin + p.rows_per_part;
  if (row_end > p.n_rows) row_end = p.n_rows;
  const int nblk = row_end > row_begin ? (int)((row_end - row_begin + RB - 1) / RB) : 0;
  Cand inf; inf.d = INFINITY; inf.r = INT_MAX;
  for (int i = tid; i < QB * LSTR; i += NT) lists[i] = inf;
  for (int i = tid; i < QB; i += NT) taus[i] = inf;

  // per lane: the query of each of its TN tiles
  // (scalars and scalar arrays only below: a private ARRAY OF STRUCTS is not promoted to registers by hipcc -- it lives in scratch, and every scratch access is a
  // vector-memory operation whose s_waitcnt vmcnt(0) also waits for the whole LDS-DMA ring: measured 50 k cycles per selection round)
  float qqv[TN]; int exclv[TN]; bool qok[TN]; float tau_d[TN]; int tau_r[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int qi = q0 + (wn * TN + j) * 32 + r32;
    qok[j] = qi < p.nq;
    qqv[j] = (METRIC == 0 && qok[j]) ? p.qq[qi] : 0.f;
    exclv[j] = (p.excl && p.group && qok[j]) ? p.excl[qi] : INT_MIN;
    tau_d[j] = INFINITY; tau_r[j] = INT_MAX;
  }

  // ---- the LDS-DMA stream: item `it` = (row block, feature slab), stage it % NST.  Per wave and slab: NTAB pieces of table rows + TN pieces of query rows
  // (1 KiB = 8 rows x 128 bytes each).  The row pointers are kept in registers (queries: fixed; table rows: per row block), so a piece costs one 64-bit add;
  // the pieces of slab it + NST - 1 are issued in four portions BETWEEN the MFMA groups of slab it (an LDS-DMA instruction takes ~100 cycles to issue: a
  // burst of 12 in front of the MFMAs idled the matrix pipe for a fifth of a slab).
  // LDS image: 128-byte rows, the 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7).  ds_read_b128 serves a wave in four 16-lane groups
  // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS) over a 256-byte bank row, i.e. a group's 8 even and 8 odd rows
  // must each hit 8 distinct chunks: (r >> 1) & 7 is distinct over them, r & 7 (the first form) was not -- every fragment read was a 2-way conflict
  // (SQ_LDS_BANK_CONFLICT = 54 % of SQ_LDS_IDX_ACTIVE).  A DMA piece is 8 rows starting at row 8 P, P = wave + NW i: (r >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7.
  const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);   // source chunk of this lane inside its 128-byte slab row
  const float* qptr[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    int qi = q0 + 8 * (wave + NW * i) + (lane >> 3);
    qi = qi < p.nq ? qi : p.nq - 1;
    qptr[i] = p.q + (long long)qi * p.dim;
  }
  const float* aptr[NTAB];
  int d_blk = 0, d_s = 0;
  auto set_rows = [&](const int blk_) {
#pragma unroll
    for (int i = 0; i < NTAB; ++i) {
      long long row = row_begin + (long long)blk_ * RB + 8 * (wave + NW * i) + (lane >> 3);
      row = row < p.n_rows ? row : p.n_rows - 1;     // (past the table -- also past the END of the stream -- the last row is re-read and never used)
      aptr[i] = p.db + row * p.dim;
    }
  };
  set_rows(0);
  auto issue_piece = [&](auto I, const int stage) {  // piece I (< NTAB: table rows, else query rows) of the cursor's slab
    constexpr int i = decltype(I)::value;
    const int kk = d_s * 32 + chunk * 4;
    const float* src = i < NTAB ? aptr[i < NTAB ? i : 0] : qptr[i < NTAB ? 0 : i - NTAB];
    src = kk < p.dim ? src + kk : g_topk_zero + chunk * 4;
    char* dst = smem + stage * STAGE + ((i < NTAB ? 0 : RB / 8) + wave + NW * (i < NTAB ? i : i - NTAB)) * 1024;
#ifndef MRAG_TOPK_DIAG_NODMA  // developer timing build: no operand traffic (the MFMAs run on whatever the LDS holds)
    glds16(src, dst);
#endif
  };
  auto advance = [&]() {
    if (++d_s == p.nslab) { d_s = 0; ++d_blk; set_rows(d_blk); }
  };
  auto issue_phase = [&](auto C, const int stage) {  // the pieces of phase C = 0..3 of a slab: piece i belongs to phase (4 i) / PPW
    constexpr int c = decltype(C)::value;
    static_for<PPW>([&](auto I) __attribute__((always_inline)) {
      if constexpr ((4 * decltype(I)::value) / PPW == c) issue_piece(I, stage);
    });
    if constexpr (c == 3) advance();
  };
  auto issue_all = [&](const int stage) {
    issue_phase(std::integral_constant<int, 0>{}, stage); issue_phase(std::integral_constant<int, 1>{}, stage);
    issue_phase(std::integral_constant<int, 2>{}, stage); issue_phase(std::integral_constant<int, 3>{}, stage);
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float xx = 0.f;
#ifdef MRAG_TOPK_DIAG_STATS
  unsigned long long dbg_t0, dbg_sel = 0, dbg_rounds = 0, dbg_a, dbg_b;
  MRAG_TSTAMP(dbg_t0);
#endif
  const int total = nblk * p.nslab;
  __syncthreads();                                   // lists / thresholds initialised
  // prologue: NST - 1 slabs in flight.  Past the end of the stream `issue` keeps requesting (the cursor clamps to the table's last row and re-reads a slab into a
  // stage nobody reads again), so EVERY iteration issues exactly PPW pieces per wave and one counted wait fits all of them
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue_all(i);
  int s = 0, blk = 0, stg = 0;
  for (int it = 0; it < total; ++it) {
    // INVARIANT of the counted wait: vmcnt retires in order and counts EVERY vector-memory operation of the wave, so between the LDS-DMA pieces of a slab and the wait
    // that guards it no other vector-memory operation may be issued -- the group-id loads, the tau_g atomics and the list stores of the selection all sit BEHIND this
    // wait in program order (they are issued after the pieces they could otherwise be mistaken for).  A later edit that puts a global access in front of it silently lets
    // the MFMAs read a half-landed stage.  -DMRAG_DIAG_VMCNT0 (tools/build_variant.sh) replaces the counted waits by vmcnt(0): the results must not change.
#ifdef MRAG_DIAG_VMCNT0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW * (NST - 2)) : "memory");   // all but the newest NST - 2 slabs have landed: slab `it` is in LDS
#endif
    __syncthreads();                                 // ... for every wave; and every wave is done reading the stage that slab it + NST - 1 overwrites
    const int nstage = stg == 0 ? NST - 1 : stg - 1; // slab it + NST - 1 -> stage (it + NST - 1) % NST
    const char* st = smem + stg * STAGE;
    stg = stg + 1 == NST ? 0 : stg + 1;
    const char* arow = st + (wm * 32 + r32) * 128;
    const char* qrow = st + (RB + wn * TN * 32 + r32) * 128;
    const int sw = (r32 >> 1) & 7;
    static_for<4>([&](auto C) __attribute__((always_inline)) {
      constexpr int c = decltype(C)::value;
      const int off = ((2 * c + h) ^ sw) * 16;
      const f32x4 a4 = *(const f32x4*)(arow + off);
      if constexpr (METRIC == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) xx = __builtin_fmaf(a4[t], a4[t], xx);
      }
      f32x4 b4[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) b4[j] = *(const f32x4*)(qrow + j * 32 * 128 + off);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4[j][0], acc[j], 0, 0, 0);
      issue_phase(C, nstage);                          // (behind the first MFMAs of the group: the DMA's issue time passes under the matrix pipe)
#pragma unroll
      for (int t = 1; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], b4[j][t], acc[j], 0, 0, 0);
    });
    if (++s < p.nslab) continue;
    // ---- end of a row block: distances, then the selection rounds
    s = 0;
#ifdef MRAG_TOPK_DIAG_NOSEL   // developer timing build (tools/topk_variants.sh): the MFMA / DMA stream alone
#if defined(__HIP_DEVICE_COMPILE__)   // (the accumulators stay live: without a consumer hipcc deletes the MFMAs; the host pass cannot parse the "v" constraint)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(acc[j]));
    asm volatile("" :: "v"(xx));
#endif
    ++blk;
    continue;
#endif
    const long long blk_row0 = row_begin + (long long)blk * RB + wm * 32;
    ++blk;
#ifdef MRAG_TOPK_DIAG_STATS
    MRAG_TSTAMP(dbg_a);
#endif
    if constexpr (METRIC == 0) {
      const float xf = xx + __shfl_xor(xx, 32);      // the two half-row chains, added once (either lane: the same two addends)
      if (h == 0 && wn == 0) xxs[wm * 32 + r32] = xf;   // (the query groups hold the same rows: one writes)
      xx = 0.f;
    }
    __syncthreads();
    int gid[16];                                      // the rows' video ids (prefilter): 16 independent loads, one wait
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) gid[reg] = INT_MIN + 1;
    if (p.group) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long long grow = blk_row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        gid[reg] = p.group[grow < row_end ? grow : row_end - 1];
      }
    }
    // the SHARED threshold of each query: the smallest k-th distance any workgroup has published so far.  A row farther than that has k rows in front of it
    // somewhere in the table and cannot be in the answer (equal distances stay: `<=`), so it never becomes a candidate here -- a workgroup sees 1 / parts of
    // the table and its own k-th distance alone admits parts-times more rows (3.5 workgroup-synchronous rounds per row block instead of ~1).  Reading a
    // stale value is harmless (the thresholds only fall).
    float gt[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) gt[j] = qok[j] ? key_float(__hip_atomic_load(p.tau_g + q0 + (wn * TN + j) * 32 + r32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -INFINITY;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const long long grow = blk_row0 + i;
      const bool valid = grow < row_end;
      const float xi = METRIC == 0 ? xxs[wm * 32 + i] : 0.f;
      const int gi = gid[reg];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float dot = acc[j][reg];
        const float d = METRIC == 0 ? __builtin_fmaf(-2.0f, dot, qqv[j] + xi) : 1.0f - dot;
        acc[j][reg] = (valid && d <= gt[j] && gi != exclv[j]) ? d : INFINITY;     // (gt = -inf for a query past the batch; NaN distances drop out too)
      }
    }
    float bd[TN];                                     // the lane's best remaining row of each tile (re-scanned only after it was consumed)
    int br[TN];
    auto rescan = [&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      // branch-free (hipcc turned the compare-and-keep form into a chain of exec-masked branches, ~370 instructions per tile): the minimum by v_min, then the
      // LOWEST register that holds it (the lowest row among equal distances)
      float m = acc[j][0];
#pragma unroll
      for (int reg = 1; reg < 16; ++reg) m = fminf(m, acc[j][reg]);
      int r = 0;
#pragma unroll
      for (int reg = 15; reg >= 1; --reg) r = acc[j][reg] == m ? reg : r;
      r = acc[j][0] == m ? 0 : r;
      bd[j] = m; br[j] = r;
    };
    static_for<TN>([&](auto J) __attribute__((always_inline)) { rescan(J); });
    for (;;) {
      bool any = false;
      int bsel[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int brow = bd[j] < INFINITY ? (int)(blk_row0 + (br[j] & 3) + 8 * (br[j] >> 2) + 4 * h) : INT_MAX;
        const bool pass = bd[j] < tau_d[j] || (bd[j] == tau_d[j] && brow < tau_r[j]);
        Cand c;
        c.d = pass ? bd[j] : INFINITY;
        c.r = pass ? brow : INT_MAX;
        slots[((wn * TN + j) * 32 + r32) * NSLOT + wm * 2 + h] = c;
        bsel[j] = pass ? br[j] : -1;
        any |= pass;
      }
#ifdef MRAG_TOPK_DIAG_STATS
      ++dbg_rounds;
#endif
      if (!__syncthreads_or(any ? 1 : 0)) break;
      if (tid < QB) {                                 // the owner of query tid: at most NSLOT insertions into its sorted list
        // the list and the round's candidates come into REGISTERS in two bursts of independent LDS reads; every insertion is then a fixed chain of
        // compare / select steps (the list keeps its best 16: entries k .. 15 are harmless extras, the threshold is entry k - 1).  A pointer-chasing
        // insertion in LDS cost ~20 k cycles per round -- two dependent LDS accesses per shifted entry -- and the rounds are workgroup-synchronous.
        Cand* Lp = lists + tid * LSTR;
        float Ld[16], cd[NSLOT];
        int Lr[16], cr[NSLOT];
        bool anyc = false;
#pragma unroll
        for (int si = 0; si < NSLOT; ++si) {
          const Cand c = slots[tid * NSLOT + si];
          cd[si] = c.d; cr[si] = c.r;
          anyc |= c.r != INT_MAX;
        }
        if (anyc) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { const Cand l = Lp[e]; Ld[e] = l.d; Lr[e] = l.r; }
          auto less = [](float ad, int ar, float bd2, int br2) { return ad < bd2 || (ad == bd2 && ar < br2); };
          // one insertion per loop trip, best candidate first: the trip count is the LARGEST number of candidates any owner of the wave holds this round
          // (1-2 once the lists are warm), not NSLOT -- a wave pays every trip of its busiest lane with all 64 lanes
          for (;;) {
            float md = INFINITY;
            int mr = INT_MAX, ms = -1;
#pragma unroll
            for (int si = 0; si < NSLOT; ++si)
              if (less(cd[si], cr[si], md, mr)) { md = cd[si]; mr = cr[si]; ms = si; }
            if (mr == INT_MAX) break;
#pragma unroll
            for (int si = 0; si < NSLOT; ++si)
              if (si == ms) { cd[si] = INFINITY; cr[si] = INT_MAX; }
#pragma unroll
            for (int e = 15; e >= 1; --e) {
              const bool before_prev = less(md, mr, Ld[e - 1], Lr[e - 1]), before_this = less(md, mr, Ld[e], Lr[e]);
              Ld[e] = before_prev ? Ld[e - 1] : (before_this ? md : Ld[e]);
              Lr[e] = before_prev ? Lr[e - 1] : (before_this ? mr : Lr[e]);
            }
            const bool b0 = less(md, mr, Ld[0], Lr[0]);
            Ld[0] = b0 ? md : Ld[0];
            Lr[0] = b0 ? mr : Lr[0];
          }
          float thd = Ld[0];
          int thr = Lr[0];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            Cand l; l.d = Ld[e]; l.r = Lr[e];
            Lp[e] = l;
            if (e == p.k - 1) { thd = Ld[e]; thr = Lr[e]; }
          }
          Cand th; th.d = thd; th.r = thr;
          taus[tid] = th;
          if (thr != INT_MAX && q0 + tid < p.nq) atomicMin(p.tau_g + q0 + tid, float_key(thd));   // a full list: publish its k-th distance
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const Cand tq = taus[(wn * TN + j) * 32 + r32];
        tau_d[j] = tq.d; tau_r[j] = tq.r;
      }
      static_for<TN>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if (__any(bsel[j] >= 0)) {                    // (wave-uniform: most tiles of most rounds have nothing to consume)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) acc[j][reg] = reg == bsel[j] ? INFINITY : acc[j][reg];   // consumed
          rescan(J);
        }
      });
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#ifdef MRAG_TOPK_DIAG_STATS
    MRAG_TSTAMP(dbg_b);
    dbg_sel += dbg_b - dbg_a;
#endif
  }
#ifdef MRAG_TOPK_DIAG_STATS
  if (tid == 0) {
    MRAG_TSTAMP(dbg_b);
    atomicAdd(&g_topk_dbg[0], dbg_rounds); atomicAdd(&g_topk_dbg[1], dbg_sel); atomicAdd(&g_topk_dbg[2], dbg_b - dbg_t0); atomicAdd(&g_topk_dbg[3], (unsigned long long)nblk);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last NST - 1 (redundant) slabs must have landed before the workgroup gives its LDS back
  __syncthreads();
  if (tid < QB && q0 + tid < p.nq) {
    Cand* out = p.lists + ((long long)(q0 + tid) * p.nparts + part) * 16;
    const Cand* L = lists + tid * LSTR;
    for (int e = 0; e < 16; ++e) out[e] = e < p.k ? L[e] : inf;
  }
}

// ---------------------------------------------------------------------------------------------- fan-out, ONE launch: tables that fit one round of workgroups
// BASELINE config #1's own table (10 000 rows x 256 queries) is ONE row block per workgroup of the streaming form above: every workgroup paid its ~6
// workgroup-synchronous selection rounds on a cold list (half of its time), and the call was three launches (|q|^2 pre-pass, fan-out, merge: 6.5 + 115 + 14 us
// + the gaps between them).  This form drops the in-kernel selection and the extra launches:
//   * the same LDS-DMA / fp32-MFMA stream over ONE row block per workgroup; |q|^2 is accumulated from the query fragments the MFMAs read anyway (the same
//     two half-block chains, added once: bit-identical to topk_qq_kernel), |x|^2 as above;
//   * the workgroup's 128 x QB first scores go through an LDS tile into a DENSE [query][row] matrix in the workspace (512-byte runs per query; rows past the
//     table, excluded rows and NaNs as +inf) -- 10 MB at 10 000 x 256;
//   * every workgroup then arrives at one counter and waits for the others (the plan launches this form only when the whole grid is resident at once --
//     hipOccupancyMaxActiveBlocksPerMultiprocessor x 256 CUs; the wait is BOUNDED: a workgroup that gives up simply leaves), and the workgroups that have
//     seen everybody arrive -- always including the last arriver -- CLAIM queries from a second counter and finish them: thread minima -> the 16th (k-th)
//     smallest minimum bounds the answer -> the few scores under that bound are compacted per wave, sorted and merged -> finish_query (second scoring,
//     filter order, output).  The result is the top-k under the total order (first score, row), i.e. what the streaming form and oracle mode 2 define.
//   * the counters live in words 8..14 of the workspace's first 64 bytes (two sets used alternately: a call's last arriver zeroes the other set).
// agent-coherent accesses of the one-launch form's hand-over data (first scores, group minima): written THROUGH the XCD's L2 (sc1) and read past it, i.e. what an
// agent-scope atomic store / load compiles to, 16 bytes wide.  Nothing of the hand-over is then dirty in an L2, and the arrival needs no release fence: a
// `buffer_wbl2` per workgroup walks the whole L2 (632 workgroups: 57 us of a 130-us call; 158: 11 us) -- a wait for the stores' acknowledgements is enough.
__device__ __forceinline__ void store_agent_x4(float* ptr, const f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ float load_agent(const float* ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef MRAG_TOPK_DIAG_STATS   // developer timing build of the one-launch form: sums (and maxima) of phase durations in s_memtime ticks -- tools/topk_diag.py names them
__device__ unsigned long long g_topk_dense_dbg[24];
extern "C" int mrag_debug_topk_dense_stats(unsigned long long* out_host, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_topk_dense_dbg), sizeof(g_topk_dense_dbg));
  if (e == hipSuccess && reset) { unsigned long long z[24] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_topk_dense_dbg), z, sizeof(z)); }
  return (int)e;
}
#define MRAG_DSTAMP(T) do { if (threadIdx.x == 0) MRAG_TSTAMP(T); } while (0)
#else
#define MRAG_DSTAMP(T) do { } while (0)
#endif

struct TopkDP {
  const float* db; const int* group; const float* q; const int* excl;
  float* dist;            // [nq][ld] first scores
  float* gmin;            // [nq][ld / 32]: the smallest first score of every 32-row group (a wave's rows of a workgroup's row block)
  unsigned* sync;         // [7]: seq | {arrivals, go, claims} x 2 (see the kernel's hand-off): zero before the first call on a workspace
  long long n_rows; int dim, nq, nslab, ld, total, spin_limit;
  TopkP mp;               // what finish_query needs (k, rescore, outputs, post-filter ids)
};

constexpr int dense_stages(int tiles) { return tiles == 1 ? 2 : 3; }   // 32 queries per workgroup: two 20-KB stages, THREE workgroups per CU (10 000 x 256: 632 workgroups = 2.47 per CU)
#ifndef MRAG_TOPK_DENSE_SLEEP
#define MRAG_TOPK_DENSE_SLEEP 32     // x 64 cycles between two looks at the `go` word (~1 us)
#endif
constexpr int DENSE_BUF = 128;      // per-wave compaction buffer of the finishing phase: < 64 left over + <= 64 new candidates per step
constexpr int DENSE_SCR = ((257 + 4 * DENSE_BUF) * 8 + (2 + 2048 + 2 + 2048) * 4 + 15) / 16 * 16;   // bytes of finishing scratch per 256-thread group: candidates | 2 counters, list of passing groups | the group minima (ld / 32 <= 2 048)

// finish query q from the dense first scores: one 256-thread group (lt = 0..255); scratch `sh` (257 candidates), `bufs` (4 x DENSE_BUF candidates) and
// `ctr` (2 + ld / 32 ints: listed groups, surviving scores, the list).  The 32-row groups' minima bound the answer (the keep-th smallest of the lanes'
// minima: `keep` groups hold a score at or under it, so the keep-th nearest row does too); only the groups whose minimum passes the bound are read at all --
// about `keep` runs of 128 bytes out of the query's 40 KB at 10 000 rows -- and ALL of them at once: the passing groups are listed first, then every thread
// loads its elements of the list (one memory round trip; the first form walked the groups two at a time, a dependent load each: 18 us per query).  The
// scores at or under the bound (about `keep` again) meet in one LDS array and are ordered by counting ranks.
__device__ __forceinline__ void dense_select(const TopkDP& p, const int q, const bool store, const int lt, Cand* sh, Cand* bufs, int* ctr, unsigned long long* dacc) {
  (void)dacc;
  constexpr int CAP = 4 * DENSE_BUF;
  const int lane = lt & 63, wave = lt >> 6;
  const int ngrp = p.ld >> 5;
  const float* D = p.dist + (long long)q * p.ld;
  const float* G = p.gmin + (long long)q * ngrp;
  const int keep = p.mp.rescore ? 16 : p.mp.k;
  int* glist = ctr + 2;
  float* gsh = (float*)(glist + 2048 + 2);              // [ngrp]: the query's group minima, loaded ONCE by the 256 threads (two loads in flight each)
  Cand inf; inf.d = INFINITY; inf.r = INT_MAX;
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
  (void)a0; (void)a1; (void)a2; (void)a3; (void)a4; (void)a5;
  MRAG_DSTAMP(a0);
  if (lt < 2) ctr[lt] = 0;
  // (every wave loading all of the minima for itself -- 8 waves x 128 workgroups x 5 uncached loads on 323 KB, i.e. on a handful of memory channels -- made
  // this first access after the wait 5 us)
  for (int g0 = 0; g0 < ngrp; g0 += 512) {
    float gv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = g0 + u * 256 + lt;
      const float* a = G + (g < ngrp ? g : 0);
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(gv[u]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gv[0]), "+v"(gv[1]) :: "memory");
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = g0 + u * 256 + lt;
      if (g < ngrp) gsh[g] = gv[u];
    }
  }
  __syncthreads();
  // ---- the bound, by every wave for itself: lane minima over ALL groups (g = lane + 64 i), the keep-th smallest of the 64 by rank counting
  Cand m = inf;
  for (int g = lane; g < ngrp; g += 64) {
    const float v = gsh[g];
    const bool b = v < m.d;
    m.d = b ? v : m.d; m.r = b ? g : m.r;
  }
  unsigned long long b1 = 0, b2 = 0; (void)b1; (void)b2;
  MRAG_DSTAMP(b1);
  float thr;
  {
    const int rank = cand_rank(m, 64);
    const unsigned long long bal = __ballot(rank == keep - 1 && m.d < INFINITY);
    thr = bal ? __shfl(m.d, __builtin_ctzll(bal)) : INFINITY;   // (+inf when fewer than `keep` lanes saw a finite score: every finite score passes then)
  }
  MRAG_DSTAMP(b2);
  __syncthreads();                                      // (the counters are zero; the rank scratch is free again)
  MRAG_DSTAMP(a1);
  // ---- the groups with a score at or under the bound, listed (any order: the result is the top of a strict total order)
  for (int g = lane + 64 * wave; g < ngrp; g += 256) {
    const float gm = gsh[g];
    const bool pass = gm < INFINITY && gm <= thr;
    const unsigned long long bal = __ballot(pass);
    if (bal) {
      int pos = 0;
      if (lane == 0) pos = atomicAdd(ctr, __popcll(bal));
      pos = __shfl(pos, 0);
      if (pass) glist[pos + __popcll(bal & ((1ull << lane) - 1ull))] = g;
    }
  }
  __syncthreads();
  MRAG_DSTAMP(a2);
  const int nel = ctr[0] * 32;
  // ---- their scores: four independent loads per thread and step; the ones at or under the bound are appended to `bufs`
  for (int e0 = 0; e0 < nel; e0 += 1024) {
    Cand c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256 + lt;
      const bool in = e < nel;
      c[u].r = in ? glist[e >> 5] * 32 + (e & 31) : INT_MAX;
      const float* a = D + (in ? c[u].r : 0);
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(c[u].d) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(c[0].d), "+v"(c[1].d), "+v"(c[2].d), "+v"(c[3].d) :: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u].d = c[u].r != INT_MAX ? c[u].d : INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool pass = c[u].d < INFINITY && c[u].d <= thr;
      const unsigned long long bal = __ballot(pass);
      if (bal) {
        int pos = 0;
        if (lane == 0) pos = atomicAdd(ctr + 1, __popcll(bal));
        pos = __shfl(pos, 0) + __popcll(bal & ((1ull << lane) - 1ull));
        if (pass && pos < CAP) bufs[pos] = c[u];
      }
    }
  }
  __syncthreads();
  MRAG_DSTAMP(a3);
  const int ns = ctr[1];
  Cand run = inf;
  if (ns <= CAP) {
    if (wave == 0) {
      if (ns <= 64) {                                   // the usual case: one rank count puts them in order
        const Cand c = lane < ns ? bufs[lane] : inf;
        const int rank = cand_rank(c, ns);
        if (lane < ns) sh[rank] = c;
        __builtin_amdgcn_wave_barrier();
        run = lane < ns ? sh[lane] : inf;
      } else {
        for (int b0 = 0; b0 < ns; b0 += 64) {
          const Cand c = b0 + lane < ns ? bufs[b0 + lane] : inf;
          run = wave_merge_top(run, wave_sort(c, lane), lane);
        }
      }
    }
  } else {
    // more scores at or under the bound than the array holds (thousands of equal scores, or a table with fewer than `keep` finite scores per lane): the
    // listed groups are read again, every wave compacts its own share, sorts 64 at a time and merges; the four runs meet in wave 0
    __syncthreads();
    Cand* buf = bufs + wave * DENSE_BUF;
    int cnt = 0;
    for (int e0 = 0; e0 < nel; e0 += 256) {
      const int e = e0 + lt;
      Cand c;
      c.r = e < nel ? glist[e >> 5] * 32 + (e & 31) : INT_MAX;
      c.d = e < nel ? load_agent(D + c.r) : INFINITY;
      const bool pass = c.d < INFINITY && c.d <= thr;
      const unsigned long long bal = __ballot(pass);
      if (pass) buf[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = c;
      cnt += __popcll(bal);
      __builtin_amdgcn_wave_barrier();                  // (one wave, in-order LDS: the reads below see the writes above)
      if (cnt >= 64) {
        cnt -= 64;
        const Cand t = buf[cnt + lane];
        run = wave_merge_top(run, wave_sort(t, lane), lane);
        __builtin_amdgcn_wave_barrier();
      }
    }
    const Cand t = lane < cnt ? buf[lane] : inf;
    run = wave_merge_top(run, wave_sort(t, lane), lane);
    __syncthreads();
    sh[wave * 64 + lane] = run;
    __syncthreads();
    if (wave == 0)
      for (int w = 1; w < 4; ++w) run = wave_merge_top(run, sh[w * 64 + lane], lane);
  }
  MRAG_DSTAMP(a4);
  finish_query(p.mp, q, run, sh, lt, store);
  MRAG_DSTAMP(a5);
#ifdef MRAG_TOPK_DIAG_STATS   // (accumulated in the caller's registers and flushed at the end of the kernel: an atomic issued here would sit in front of the next vmcnt wait)
  if (threadIdx.x == 0) {
    dacc[0] += 1; dacc[1] += a1 - a0; dacc[2] += a2 - a1; dacc[3] += a3 - a2; dacc[4] += a4 - a3; dacc[5] += a5 - a4;
    dacc[6] = dacc[6] > a5 - a0 ? dacc[6] : a5 - a0; dacc[7] += b1 - a0; dacc[8] += b2 - b1; dacc[9] += a1 - b2;
  }
#endif
}

// QBU = queries a workgroup USES of the 32 TN WN its LDS image holds.  96 of 128 (the eight-wave workgroup whose second query group computes ONE of its two tiles):
// three 32-query tiles per workgroup -- BASELINE config #1's 632 tiles then quantise to 3 per busy CU (237 workgroups) instead of 4 (158), with two waves per SIMD.
template <int METRIC, int TN, int WN, int QBU = 32 * TN * WN>
__global__ __launch_bounds__(256 * WN, TN * WN == 1 ? 3 : WN == 1 ? 2 : 1) void topk_dense_kernel(const TopkDP p) {   // (waves per SIMD: 3 / 2 / 2 workgroups per CU)
  constexpr int WM = 4, NW = WM * WN, NT = 64 * NW, RB = 32 * WM, QB = 32 * TN * WN, NST = dense_stages(TN * WN);
  static_assert(QBU % 32 == 0 && QBU <= QB && QBU > QB - 32 * TN, "only the last query group may run short");
  constexpr int STAGE = (RB + QB) * 128, NPIECE = (RB + QB) / 8, PPW = NPIECE / NW, NTAB = (RB / 8) / NW;
  constexpr int LD = RB + 4;                           // floats per query of the LDS score tile (16-byte aligned rows, spread over the banks)
  constexpr int NG = NT / 256;                         // 256-thread groups of the finishing phase
  static_assert(NPIECE % NW == 0 && (RB / 8) % NW == 0 && PPW == NTAB + TN, "every wave issues the same number of LDS-DMA pieces per slab (the counted vmcnt wait relies on it)");
  static_assert(NST * STAGE >= QB * LD * 4, "the score tile overlays the drained operand ring");
  constexpr int SCR = DENSE_SCR;
  static_assert(NST * STAGE >= NG * SCR, "the finishing phase's scratch overlays it as well");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xxs = (float*)(smem + NST * STAGE);           // [4][32]: |x|^2 of the wave's 32 rows
  unsigned* flag = (unsigned*)(xxs + 128);
  float* gml = xxs + 132;                              // [QB][4]: the waves' minima per query (16-byte aligned)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & (WM - 1), wn = wave / WM;
  const int r32 = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * QBU, part = blockIdx.x;
  const bool all_tiles = QBU == QB || (wn * TN + TN) * 32 <= QBU;   // (wave-uniform) this wave's query group computes all of its TN tiles
  const long long row_begin = (long long)part * RB;
  const long long row_end = row_begin + RB < p.n_rows ? row_begin + RB : p.n_rows;
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  (void)s0; (void)s1; (void)s2; (void)s3; (void)s4;
  MRAG_DSTAMP(s0);

  int exclv[TN]; bool qok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int qi = q0 + (wn * TN + j) * 32 + r32;
    qok[j] = qi < p.nq && (wn * TN + j) * 32 < QBU;
    exclv[j] = (p.excl && p.group && qok[j]) ? p.excl[qi] : INT_MIN;
  }
  // ---- the LDS-DMA stream of topk_mfma_kernel over the row block's slabs (same image, same swizzle, same counted wait)
  const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const float* qptr[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    int qi = q0 + 8 * (wave + NW * i) + (lane >> 3);
    qi = qi < p.nq ? qi : p.nq - 1;
    qptr[i] = p.q + (long long)qi * p.dim;
  }
  const float* aptr[NTAB];
#pragma unroll
  for (int i = 0; i < NTAB; ++i) {
    long long row = row_begin + 8 * (wave + NW * i) + (lane >> 3);
    row = row < p.n_rows ? row : p.n_rows - 1;
    aptr[i] = p.db + row * p.dim;
  }
  int d_s = 0;
  auto issue_piece = [&](auto I, const int stage) {
    constexpr int i = decltype(I)::value;
    const int ds = d_s < p.nslab ? d_s : p.nslab - 1;   // (past the end the last slab is re-read into a stage nobody reads again: every iteration issues PPW pieces)
    const int kk = ds * 32 + chunk * 4;
    const float* src = i < NTAB ? aptr[i < NTAB ? i : 0] : qptr[i < NTAB ? 0 : i - NTAB];
    src = kk < p.dim ? src + kk : g_topk_zero + chunk * 4;
    char* dst = smem + stage * STAGE + ((i < NTAB ? 0 : RB / 8) + wave + NW * (i < NTAB ? i : i - NTAB)) * 1024;
    glds16(src, dst);
  };
  auto issue_phase = [&](auto C, const int stage) {
    constexpr int c = decltype(C)::value;
    static_for<PPW>([&](auto I) __attribute__((always_inline)) {
      if constexpr ((4 * decltype(I)::value) / PPW == c) issue_piece(I, stage);
    });
    if constexpr (c == 3) ++d_s;
  };
  auto issue_all = [&](const int stage) {
    issue_phase(std::integral_constant<int, 0>{}, stage); issue_phase(std::integral_constant<int, 1>{}, stage);
    issue_phase(std::integral_constant<int, 2>{}, stage); issue_phase(std::integral_constant<int, 3>{}, stage);
  };
  f32x16 acc[TN];
  float qq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    qq[j] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  }
  float xx = 0.f;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue_all(i);
  int stg = 0;
  for (int it = 0; it < p.nslab; ++it) {
    // (the invariant of topk_mfma_kernel's counted wait holds here as well: no other vector-memory operation between a slab's pieces and this wait)
#ifdef MRAG_DIAG_VMCNT0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW * (NST - 2)) : "memory");
#endif
    __syncthreads();
    const int nstage = stg == 0 ? NST - 1 : stg - 1;
    const char* st = smem + stg * STAGE;
    stg = stg + 1 == NST ? 0 : stg + 1;
    const char* arow = st + (wm * 32 + r32) * 128;
    const char* qrow = st + (RB + wn * TN * 32 + r32) * 128;
    const int sw = (r32 >> 1) & 7;
    auto slab = [&](auto NTL) __attribute__((always_inline)) {           // NTL = tiles this wave computes (TN, or fewer in a short last query group)
      constexpr int ntl = decltype(NTL)::value;
      static_for<4>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        const int off = ((2 * c + h) ^ sw) * 16;
        const f32x4 a4 = *(const f32x4*)(arow + off);
        f32x4 b4[ntl];
#pragma unroll
        for (int j = 0; j < ntl; ++j) b4[j] = *(const f32x4*)(qrow + j * 32 * 128 + off);
        if constexpr (METRIC == 0) {                   // the half-block chains of |x|^2 (this lane's row) and |q|^2 (this lane's query of every tile)
#pragma unroll
          for (int t = 0; t < 4; ++t) xx = __builtin_fmaf(a4[t], a4[t], xx);
#pragma unroll
          for (int j = 0; j < ntl; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) qq[j] = __builtin_fmaf(b4[j][t], b4[j][t], qq[j]);
        }
#pragma unroll
        for (int j = 0; j < ntl; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4[j][0], acc[j], 0, 0, 0);
        issue_phase(C, nstage);
#pragma unroll
        for (int t = 1; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < ntl; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], b4[j][t], acc[j], 0, 0, 0);
      });
    };
    if constexpr (QBU == QB) slab(std::integral_constant<int, TN>{});
    else {
      if (all_tiles) slab(std::integral_constant<int, TN>{});
      else slab(std::integral_constant<int, (QBU / 32) % TN>{});
    }
  }
  (void)all_tiles;
  MRAG_DSTAMP(s1);
  // ---- first scores of the row block
  if constexpr (METRIC == 0) {
    const float xf = xx + __shfl_xor(xx, 32);          // the two half-row chains, added once (either lane: the same two addends)
    if (h == 0 && wn == 0) xxs[wm * 32 + r32] = xf;
#pragma unroll
    for (int j = 0; j < TN; ++j) qq[j] = qq[j] + __shfl_xor(qq[j], 32);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's last (redundant) slabs have landed: the stages are free for the score tile
  __syncthreads();
  const long long blk_row0 = row_begin + wm * 32;
  int gid[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) gid[reg] = INT_MIN + 1;
  if (p.group) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long long grow = blk_row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      gid[reg] = p.group[grow < row_end ? grow : row_end - 1];
    }
  }
  float* tile = (float*)smem;                          // [QB][LD]
  float mn[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) mn[j] = INFINITY;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int qn = (wn * TN + j) * 32 + r32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int reg = 4 * g4 + e, i = e + 8 * g4 + 4 * h;
        const bool valid = blk_row0 + i < row_end;
        const float xi = METRIC == 0 ? xxs[wm * 32 + i] : 0.f;
        const float dot = acc[j][reg];
        const float d = METRIC == 0 ? __builtin_fmaf(-2.0f, dot, qq[j] + xi) : 1.0f - dot;
        o[e] = (valid && qok[j] && d <= INFINITY && gid[reg] != exclv[j]) ? d : INFINITY;   // (NaN scores drop out: the compare is false)
      }
      *(f32x4*)(tile + qn * LD + wm * 32 + 8 * g4 + 4 * h) = o;
      mn[j] = fminf(fminf(mn[j], fminf(o[0], o[1])), fminf(o[2], o[3]));
    }
    mn[j] = fminf(mn[j], __shfl_xor(mn[j], 32));
    if (h == 0) gml[qn * 4 + wm] = mn[j];
  }
  __syncthreads();
  if (tid < QBU && q0 + tid < p.nq) store_agent_x4(p.gmin + ((long long)(q0 + tid) * (p.ld >> 5) + part * 4), *(const f32x4*)(gml + tid * 4));
  for (int idx = tid; idx < QBU * (RB / 4); idx += NT) {
    const int qn = idx / (RB / 4), c4 = idx % (RB / 4);
#if defined(MRAG_TOPK_DENSE_DIAG_NOD) && MRAG_TOPK_DENSE_DIAG_NOD == 1     // developer timing builds (results are NOT valid): no score stores / plain (cached) stores
    (void)qn; (void)c4;
#elif defined(MRAG_TOPK_DENSE_DIAG_NOD) && MRAG_TOPK_DENSE_DIAG_NOD == 2
    if (q0 + qn < p.nq) *(f32x4*)(p.dist + (long long)(q0 + qn) * p.ld + row_begin + c4 * 4) = *(const f32x4*)(tile + qn * LD + c4 * 4);
#else
    if (q0 + qn < p.nq) store_agent_x4(p.dist + (long long)(q0 + qn) * p.ld + row_begin + c4 * 4, *(const f32x4*)(tile + qn * LD + c4 * 4));
#endif
  }
#if defined(MRAG_TOPK_DENSE_DIAG) && MRAG_TOPK_DENSE_DIAG == 1   // developer timing build: the stream + the dense stores alone (results are NOT produced)
  return;
#endif
  if (p.total == 0) return;                            // the two-launch form: topk_dense_finish_kernel follows (kernel boundary = the hand-over)
  // ---- arrive; wait (bounded) until the grid has arrived; finish the queries of this workgroup's arrival ticket.
  // Words (the workspace's zeroed first 64 bytes, words 8..14): seq | set 0 {arrivals, go, claims} | set 1 {..}.  A call uses set (seq & 1); its last arriver
  // zeroes the OTHER set, publishes `go` and bumps seq, so nothing is reset behind anybody's back and no exit counter is needed (a third same-address atomic
  // per workgroup).  The waiters poll `go`, not the arrival counter (632 pollers on the word the late arrivers still have to increment cost 55 us).
  //   go = 1: every workgroup is here -> STATIC shares: arrival ticket t finishes queries t, t + total, .. (no claim traffic);
  //   go = 2: somebody gave up waiting (it added 0x10000 to the arrival word before it left, so the last arriver -- whose own increment returns the word --
  //           cannot miss it) -> the workgroups that are here CLAIM queries from the set's third word; the last arriver is always among them.
  //   A workgroup that gives up and learns from its own 0x10000 increment that everybody HAS arrived meanwhile stays: `go` is already on its way.
  // Hand-off: agent-coherent (write-through) stores of the scores -> every wave waits for their acknowledgements -> barrier -> one lane: relaxed agent
  // fetch_add; waiters: relaxed agent loads of `go`, barrier, agent-coherent loads of the scores (store_agent_x4 / load_agent above: no L2-wide fences).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MRAG_DSTAMP(s2);
  unsigned seq = 0, ret = 0;                           // (lane 0's)
  unsigned *set = nullptr, *other = nullptr;
  if (tid == 0) {
    seq = __hip_atomic_load(p.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    set = p.sync + 1 + 3 * (seq & 1u);
    other = p.sync + 1 + 3 * ((seq & 1u) ^ 1u);
#ifdef MRAG_TOPK_DENSE_FENCES   // developer A/B build: the fences of the textbook hand-off on top of the write-through stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ret = __hip_atomic_fetch_add(set, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = ret & 0xffffu;
    const unsigned ticket = ret & 0xffffu;
    unsigned mode = 0;
    if (ticket + 1u == (unsigned)p.total) {
#ifdef MRAG_TOPK_DENSE_FENCES
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
#endif
      mode = (ret >> 16) ? 2u : 1u;
      __hip_atomic_store(other + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(other + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(other + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(set + 1, mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.sync, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int spin = 0; spin < p.spin_limit && !mode; ++spin) {
        __builtin_amdgcn_s_sleep(MRAG_TOPK_DENSE_SLEEP);
        mode = __hip_atomic_load(set + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!mode) {
        const unsigned r2 = __hip_atomic_fetch_add(set, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((r2 & 0xffffu) == (unsigned)p.total)
          do {
            __builtin_amdgcn_s_sleep(MRAG_TOPK_DENSE_SLEEP);
            mode = __hip_atomic_load(set + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while (!mode);
      }
    }
#ifdef MRAG_TOPK_DENSE_FENCES
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    flag[0] = mode; flag[2] = seq & 1u;
  }
  __syncthreads();
  const unsigned mode = flag[0], ticket = flag[1];
  MRAG_DSTAMP(s3);
  unsigned long long dacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  (void)dacc;
  unsigned* claims = p.sync + 1 + 3 * flag[2] + 2;
  __syncthreads();
#if defined(MRAG_TOPK_DENSE_DIAG) && MRAG_TOPK_DENSE_DIAG == 2   // developer timing build: ... + the grid wait, no finishing phase
  return;
#endif
#ifdef MRAG_TOPK_DIAG_STATS
#define MRAG_DENSE_FLUSH()                                                                                                                              \
  if (tid == 0) {                                                                                                                                        \
    atomicAdd(&g_topk_dense_dbg[0], 1ull); atomicAdd(&g_topk_dense_dbg[1], s1 - s0); atomicAdd(&g_topk_dense_dbg[2], s2 - s1); atomicAdd(&g_topk_dense_dbg[3], s3 - s2); \
    atomicMax(&g_topk_dense_dbg[4], s1 - s0); atomicMax(&g_topk_dense_dbg[5], s2 - s0); atomicMax(&g_topk_dense_dbg[6], s3 - s0);                     \
    if (dacc[0]) {                                                                                                                                       \
      atomicAdd(&g_topk_dense_dbg[8], dacc[0]); atomicAdd(&g_topk_dense_dbg[9], dacc[1]); atomicAdd(&g_topk_dense_dbg[10], dacc[2]); atomicAdd(&g_topk_dense_dbg[11], dacc[3]); \
      atomicAdd(&g_topk_dense_dbg[12], dacc[4]); atomicAdd(&g_topk_dense_dbg[13], dacc[5]); atomicMax(&g_topk_dense_dbg[14], dacc[6]);                 \
      atomicAdd(&g_topk_dense_dbg[15], dacc[7]); atomicAdd(&g_topk_dense_dbg[16], dacc[8]); atomicAdd(&g_topk_dense_dbg[17], dacc[9]);                 \
    }                                                                                                                                                    \
  }
#else
#define MRAG_DENSE_FLUSH()
#endif
  if (mode == 0) { MRAG_DENSE_FLUSH(); return; }
  Cand* sh = (Cand*)(smem + (tid >> 8) * SCR);
  Cand* bufs = sh + 257;
  int* glist = (int*)(bufs + 4 * DENSE_BUF);   // (2 counters + the list)
  if (mode == 1) {
    for (long long idx = ticket; idx * NG < p.nq; idx += p.total) {
      const int q = (int)idx * NG + (tid >> 8);
      dense_select(p, q < p.nq ? q : p.nq - 1, q < p.nq, tid & 255, sh, bufs, glist, dacc);
    }
    MRAG_DENSE_FLUSH();
    return;
  }
  for (;;) {
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(claims, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned claim = flag[0];
    __syncthreads();
    if ((long long)claim * NG >= p.nq) break;
    const int q = (int)claim * NG + (tid >> 8);
    dense_select(p, q < p.nq ? q : p.nq - 1, q < p.nq, tid & 255, sh, bufs, glist, dacc);
  }
  MRAG_DENSE_FLUSH();
}
#undef MRAG_DENSE_FLUSH

// the finishing phase as a launch of its own (a workgroup per query): the dense form of tables whose grid is not resident at once
__global__ __launch_bounds__(256) void topk_dense_finish_kernel(const TopkDP p) {
  __shared__ __attribute__((aligned(16))) char scr[DENSE_SCR];
  Cand* sh = (Cand*)scr;
  Cand* bufs = sh + 257;
  unsigned long long dacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  dense_select(p, blockIdx.x, true, threadIdx.x, sh, bufs, (int*)(bufs + 4 * DENSE_BUF), dacc);
}

// compute units of the current device (the grid wait of the one-launch form is taken only when the runtime's occupancy x this count holds the whole grid)
inline int dense_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
    cus = n;
  }
  return cus;
}

// the fan-out plan: queries per workgroup (32 TN), parts (workgroups along the table), rows per part
struct MfmaPlan { int TN, WN, QB, RB, gy, parts, rows_per_part; size_t lds, bytes; };
inline MfmaPlan plan_mfma(long long n_rows, int nq) {
  MfmaPlan pl;
  const int qtiles = (nq + 31) / 32;
  pl.RB = 128;
  const long long blocks = (n_rows + pl.RB - 1) / pl.RB;
  // Query tile (32 TN queries per workgroup).  A table that fits ONE round of workgroups with one row block each takes the SMALLEST tile that still fits the
  // round (512 workgroups at TN = 1 -- two per CU --, 256 above): every workgroup pays its pipeline fill and the ~6 selection rounds of a cold list once, so
  // more, smaller workgroups finish sooner (4 000 rows x 256 queries: 94 / 111 / 137 / 191 us at TN = 1 / 2 / 4 / 8; 10 000 rows: 166 / 169 / 140 / 190;
  // 20 000 rows: 221 / 225 / 231 / 190; profiles/r5_topk_query_tile_by_table_size.txt).  A larger table streams: the widest tile the batch fills (one pass
  // over the table per 256 queries).
  int widest = 1;
  for (int t = 8; t >= 1; t >>= 1)
    if (t <= qtiles) { widest = t; break; }
  pl.TN = widest;
  for (int t = 1; t <= widest; t <<= 1)
    if (blocks * ((qtiles + t - 1) / t) <= (t == 1 ? 512 : 256)) { pl.TN = t; break; }
#ifdef MRAG_TOPK_FORCE_TN   // developer knob (tools/topk_variants.sh): the query tile of every plan
  pl.TN = 1;
  for (int t = MRAG_TOPK_FORCE_TN; t >= 1; t >>= 1)
    if (t <= qtiles) { pl.TN = t; break; }
#endif
  pl.QB = 32 * pl.TN;
  pl.lds = (size_t)mfma_stages(pl.TN) * (pl.RB + pl.QB) * 128 + (size_t)pl.QB * (17 + 8 + 1) * sizeof(Cand) + 4 * 32 * sizeof(float);
  pl.gy = (nq + pl.QB - 1) / pl.QB;
  int per_cu = (int)((160 * 1024) / pl.lds);
  per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;                  // (160-208 VGPRs at TN <= 2: two workgroups per CU)
  long long parts = (256LL * per_cu + pl.gy - 1) / pl.gy;
  parts = parts < 1 ? 1 : parts > blocks ? blocks : parts;
  const long long bpp = (blocks + parts - 1) / parts;
  pl.parts = (int)((blocks + bpp - 1) / bpp);
  // waves: the 256-query workgroup is eight waves (two query groups of four tiles); so is the 128-query workgroup of a ONE-row-block plan (two groups of two tiles):
  // its 64 MFMAs per slab and wave were the long pole of a 10 000-row search (42 us on 158 workgroups), and two waves per SIMD halve them
  pl.WN = (pl.TN == 8 || (pl.TN == 4 && bpp == 1)) ? 2 : 1;
  pl.rows_per_part = (int)(bpp * pl.RB);
  pl.bytes = 2 * (((size_t)nq * sizeof(float) + 255) / 256 * 256) + (size_t)nq * pl.parts * 16 * sizeof(Cand);   // |q|^2, shared thresholds, per-workgroup lists
  return pl;
}

inline bool mfma_applies(int nq, int k, int dim) { return nq >= 16 && k <= 16 && dim % 4 == 0; }

// the plan of the dense forms: `ok` = the first scores fit the workspace (<= 65 536 rows, <= 64 MB); `resident` = a tile exists whose whole grid is on the chip at
// once -> ONE launch with the grid wait (checked against the runtime's occupancy at launch); otherwise TWO launches: the same kernel without the wait, then
// topk_dense_finish_kernel (a workgroup per query) -- still no pre-pass and no in-kernel selection rounds (20 000 x 256: 200 -> ~100 us).
struct DensePlan { bool ok, resident; int TN, WN, QB, gy, parts, ld; size_t lds, bytes; };   // QB = queries a workgroup covers (96: the eight-wave workgroup of three tiles)
inline size_t dense_lds(int tn, int wn) {
  return (size_t)dense_stages(tn * wn) * (128 + 32 * tn * wn) * 128 + 132 * sizeof(float) + (size_t)32 * tn * wn * 4 * sizeof(float);   // ring | |x|^2, flag | group minima
}
inline DensePlan plan_dense(long long n_rows, int nq, int dim = 768) {
  DensePlan pl{};
  const int qtiles = (nq + 31) / 32;
  const long long blocks = (n_rows + 127) / 128;
  if (blocks > 512 || (long long)nq * blocks * 128 > (16LL << 20)) return pl;          // (<= 64 MB of first scores)
  pl.ok = true; pl.parts = (int)blocks; pl.ld = (int)blocks * 128;
  pl.bytes = (size_t)nq * pl.ld * sizeof(float) + (size_t)nq * (pl.ld / 32) * sizeof(float);     // first scores | group minima (the same for every tile)
  // Tile = the cheapest of (TN, WN) = 128 queries on eight waves, 64 on four / eight, 32 on four under a two-term model measured at 10 000 x 256 x 768
  // (profiles/r6_topk_one_launch.txt): the busiest CU's MFMA time -- workgroups per CU x 32-query tiles per workgroup x 0.49 us per 32-feature slab -- plus the
  // arrivals at the grid wait, which are same-address atomics and serialise at ~0.045 us each (632 workgroups of 32 queries: 35 + 28 us; 158 of 128: 47 + 7).
  // Small tables take the small tiles (4 000 x 256: 256 workgroups of one tile), BASELINE config #1's takes 128 queries per workgroup.  A grid that is not
  // resident at once (two launches, no wait) is priced by its work per CU plus one workgroup's duration (the tail): it takes the small tiles.  The three-tile
  // workgroup (96 of the 128 queries an eight-wave workgroup holds) exists for config #1's size: 632 tiles = 237 x 3 -> 3 per busy CU (75.7 us) instead of 158 x 4 (80.8).
#ifdef MRAG_TOPK_DENSE_TILE       // developer knob: 11, 21, 22 = TN WN of every plan; 23 = the three-tile workgroup
  const int cand[1][3] = {{MRAG_TOPK_DENSE_TILE == 23 ? 2 : MRAG_TOPK_DENSE_TILE / 10, MRAG_TOPK_DENSE_TILE == 23 ? 2 : MRAG_TOPK_DENSE_TILE % 10, MRAG_TOPK_DENSE_TILE == 23 ? 3 : 0}};
#else
  const int cand[4][3] = {{2, 2, 0}, {2, 2, 3}, {2, 1, 0}, {1, 1, 0}};                           // TN, WN, tiles used (0 = all).  (64 queries on EIGHT waves -- TN 1, WN 2 -- measured 3 % behind four waves and was never the rule's choice: not instantiated)
#endif
  double best = 0;
  bool have = false;
  for (int pass = 0; pass < 2 && !have; ++pass)                                        // pass 0: resident grids; pass 1: any
    for (const auto& c : cand) {
      const int held = c[0] * c[1], tiles = c[2] ? c[2] : held, qb = 32 * tiles, gy = (nq + qb - 1) / qb;
      const int cap = held == 1 ? 3 : held == 2 ? 2 : 1;                                // (the kernels' launch bounds: workgroups per CU)
      int per_cu = (int)((160 * 1024) / dense_lds(c[0], c[1]));
      per_cu = per_cu > cap ? cap : per_cu;
      if (qb > 32 * qtiles && qb > 32) continue;                                       // (a tile wider than the batch)
      const long long wgs = blocks * gy;
      const bool res = wgs <= 256LL * per_cu;
      if (pass == 0 && !res) continue;
      const double slab = ((dim + 31) / 32) * 0.49;
      const double t = res ? (double)((wgs + 255) / 256) * tiles * slab + 0.045 * (double)wgs
                           : ((double)wgs * tiles / 256.0 + tiles) * slab;              // (dispatched as CUs free up: the work per CU + one workgroup's duration as the tail)
      if (have && t >= best) continue;
      best = t; have = true;
      pl.resident = res; pl.TN = c[0]; pl.WN = c[1]; pl.QB = qb; pl.gy = gy;
      pl.lds = dense_lds(c[0], c[1]);
    }
  pl.ok = have;
  if (!have) pl.bytes = 0;
  return pl;
}
// `order = 0` (automatic) takes the fan-out form whenever it applies.  Measured on MI355X (tools/topk_sizes.py, k = 12, D = 768; fan-out / scan kernel):
// 1 000 rows x 256 queries 94 / 164 us, 10 000 x 256 140 / 228 us, 10^5 x 256 0.59 / 2.0 ms, 10^6 x 256 3.8 / 17.2 ms; 10 000 x 16 90 / 158 us
// (profiles/r5_topk_fanout_vs_scan_by_size.txt).  (Until the |q|^2 pre-pass became a wave per query -- it took 91 us as a thread per query -- the two forms were
// equal at 10 000 rows and the switch sat at 32 768.)
inline bool mfma_auto(long long n_rows, int nq, int k, int dim) { (void)n_rows; return mfma_applies(nq, k, dim); }

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
  const int64_t scan = kTicketBytes + (int64_t)n_queries * slices * 4 * 64 * (int64_t)sizeof(Cand);
  const int64_t fan = n_queries >= 16 ? kTicketBytes + (int64_t)plan_mfma(n_rows, n_queries).bytes : 0;   // either form fits (the `order` argument picks one)
  const int64_t dense = n_queries >= 16 ? kTicketBytes + (int64_t)plan_dense(n_rows, n_queries).bytes : 0;  // (0 bytes when the one-launch plan does not apply)
  const int64_t m = scan > fan ? scan : fan;
  return m > dense ? m : dense;
}

extern "C" int mrag_topk_f32(void* stream, const float* db, const int32_t* group, int64_t n_rows, int32_t dim, const float* queries,
                             const int32_t* exclude, int32_t n_queries, int32_t k, int32_t metric, int32_t* out_rows, float* out_dist,
                             void* workspace, int64_t workspace_bytes, int32_t postfilter, int32_t order) {
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
  if (order < 0 || order > 4) return MRAG_EINVAL;
  if (order >= 2 && !mfma_applies(n_queries, k, dim)) return MRAG_ENOTSUP;
  if (order >= 2 || (order == 0 && mfma_auto(n_rows, n_queries, k, dim))) {
    // ---- the fan-out form in ONE launch: tables whose grid is resident at once (order 3 = never, order 4 = this form without waiting: diagnostics)
    if (order != 3) {
      const DensePlan dp = plan_dense(n_rows, n_queries, dim);
      TopkDP d{};
      d.db = db; d.group = p.group; d.q = queries; d.excl = exclude; d.n_rows = n_rows; d.dim = dim; d.nq = n_queries; d.nslab = (dim + 31) / 32;
      d.dist = (float*)((char*)workspace + kTicketBytes); d.sync = (unsigned*)workspace + 8;
      d.gmin = d.dist + (size_t)n_queries * dp.ld;
      d.ld = dp.ld; d.total = dp.resident ? dp.parts * dp.gy : 0;
      d.spin_limit = order == 4 ? 0 : 40000;                // x ~1 us of s_sleep: a workgroup that has not seen the grid arrive by then leaves (the last arriver finishes alone)
      d.mp = p; d.mp.rescore = metric == 0 ? 1 : 0;
      int done = 0;
#define MRAG_TOPK_DENSE(M, T, W, U)                                                                                            \
      if (dp.ok && !done && metric == M && dp.TN == T && dp.WN == W && dp.QB == U) {                                           \
        auto kfn = topk_dense_kernel<M, T, W, U>;                                                                                  \
        static int occ = -1;                                                                                                    \
        if (occ < 0) {                                                                                                          \
          hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp.lds);       \
          if (e != hipSuccess) return (int)e;                                                                                  \
          int o = 0;                                                                                                            \
          e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)kfn, 256 * W, dp.lds);                             \
          if (e != hipSuccess) return (int)e;                                                                                  \
          occ = o;                                                                                                              \
        }                                                                                                                       \
        if (d.total && (long long)d.total > (long long)dense_cus() * occ) d.total = 0;   /* (fewer resident workgroups than planned: no wait, two launches) */ \
        if (d.total || order != 4) {                                                                                            \
          MRAG_LAUNCH(kfn, dim3(dp.parts, dp.gy), dim3(256 * W), dp.lds, s, d);                                                 \
          done = 1;                                                                                                             \
        }                                                                                                                       \
      }
      MRAG_TOPK_DENSE(0, 1, 1, 32) MRAG_TOPK_DENSE(0, 2, 1, 64) MRAG_TOPK_DENSE(0, 2, 2, 128) MRAG_TOPK_DENSE(0, 2, 2, 96)
      MRAG_TOPK_DENSE(1, 1, 1, 32) MRAG_TOPK_DENSE(1, 2, 1, 64) MRAG_TOPK_DENSE(1, 2, 2, 128) MRAG_TOPK_DENSE(1, 2, 2, 96)
#undef MRAG_TOPK_DENSE
      if (done) {
        MRAG_LAUNCH_CHECK();
        MRAG_COUNT(MRAG_K_TOPK_DENSE);
        if (d.total == 0) {
          MRAG_LAUNCH(topk_dense_finish_kernel, dim3(n_queries), dim3(256), 0, s, d);
          MRAG_LAUNCH_CHECK();
          MRAG_COUNT(MRAG_K_TOPK_DENSE_FINISH);
        }
        return MRAG_OK;
      }
      if (order == 4) return MRAG_ENOTSUP;
    }
    // ---- the fan-out form: one fp32 MFMA pass over the table per 256 queries
    const MfmaPlan pl = plan_mfma(n_rows, n_queries);
    TopkMP m{};
    m.db = db; m.group = p.group; m.q = queries; m.excl = exclude; m.n_rows = n_rows; m.dim = dim; m.nq = n_queries;
    m.k = metric == 0 ? 16 : k;      // "l2": the second scoring takes the 16 nearest under the first score, so the lists (and their thresholds) are 16 deep
    m.nparts = pl.parts; m.rows_per_part = pl.rows_per_part; m.nslab = (dim + 31) / 32;
    const size_t qbytes = ((size_t)n_queries * sizeof(float) + 255) / 256 * 256;
    float* qq = (float*)((char*)workspace + kTicketBytes);
    m.qq = qq;
    m.tau_g = (unsigned*)((char*)workspace + kTicketBytes + qbytes);
    m.lists = (Cand*)((char*)workspace + kTicketBytes + 2 * qbytes);
    MRAG_LAUNCH(topk_qq_kernel, dim3(n_queries), dim3(64), 0, s, queries, qq, m.tau_g, n_queries, dim, metric == 0 ? 1 : 0);
    MRAG_LAUNCH_CHECK();
    const dim3 mgrid(pl.parts, pl.gy);
#define MRAG_TOPK_MFMA(M, T, W)                                                                                        \
    if (metric == M && pl.TN == T * W && pl.WN == W) {                                                                 \
      auto kfn = topk_mfma_kernel<M, T, W>;                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);   \
      if (e != hipSuccess) return (int)e;                                                                              \
      MRAG_LAUNCH(kfn, mgrid, dim3(256 * W), pl.lds, s, m);                                                            \
    }
    MRAG_TOPK_MFMA(0, 1, 1) MRAG_TOPK_MFMA(0, 2, 1) MRAG_TOPK_MFMA(0, 4, 1) MRAG_TOPK_MFMA(0, 4, 2) MRAG_TOPK_MFMA(0, 2, 2)   // (256 queries per workgroup: eight waves of four tiles)
    MRAG_TOPK_MFMA(1, 1, 1) MRAG_TOPK_MFMA(1, 2, 1) MRAG_TOPK_MFMA(1, 4, 1) MRAG_TOPK_MFMA(1, 4, 2) MRAG_TOPK_MFMA(1, 2, 2)
#undef MRAG_TOPK_MFMA
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_TOPK_MFMA);
    p.ws = m.lists; p.slices = pl.parts; p.wpb = 1; p.rescore = metric == 0 ? 1 : 0;
    MRAG_LAUNCH(topk_merge_kernel<16>, dim3(n_queries), dim3(256), 0, s, p);
    MRAG_LAUNCH_CHECK();
    MRAG_COUNT(MRAG_K_TOPK_MERGE);
    return MRAG_OK;
  }
  const int QT = pick_qt(n_queries), nj = (dim + 63) / 64;
  // blocks of 64 floats per register-ring step: 4 for the single query, 2 for query tiles (their chains need the registers)
  const int JCsel = QT == 1 ? (nj % 4 == 0 ? 4 : 1) : (nj % 2 == 0 ? 2 : 1);
  size_t lds = (QT == 1 ? (size_t)nj * 64 : (size_t)nj * 16 * (4 * QT + 4)) * sizeof(float);
  const dim3 grid(p.slices, (n_queries + QT - 1) / QT), block(256);
  // <= 4 queries (the interactive search of rag.py:63-80): ONE launch, the last workgroup to arrive merges (needs the first 64 workspace
  // bytes ZERO on entry -- see the header; the kernel leaves them zero)
  const bool fused = n_queries <= 4 && ((uintptr_t)workspace & 15) == 0;
  if (small && !fused) return MRAG_EINVAL;                                          // the small-database plan exists only in the fused form
  if (fused) {
    const size_t need = (size_t)(QT * 3 * 64 > 257 ? QT * 3 * 64 : 257) * sizeof(Cand);   // the four-wave pre-merge ([QT][3][64]) / the last arriver's merge scratch
    if (lds < need) lds = need;
    p.wpb = 1;                                                                          // one list per workgroup leaves the fused kernel
  }
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
  MRAG_LAUNCH(topk_merge_kernel<64>, dim3(n_queries), dim3(256), 0, s, p);
  MRAG_LAUNCH_CHECK();
  MRAG_COUNT(MRAG_K_TOPK_MERGE);
  return MRAG_OK;
}
