/* ORACLE (test infrastructure only -- never linked or called by the product path).
 *
 * Plain-C restatement of the retrieval step of MCG-NJU/MotionRAG:
 *   RAGDatabase.text_search -> vector_search (src/data/rag.py:63-80,36-61), i.e.
 *   table.search(vec, "text_embedding").limit(k)[.where('video != "<self>"')]
 * The arithmetic lives in the third-party dependency lancedb==0.14.0 (requirements.txt:18, Rust `lance`
 * core), absent from /root/reference.  Its published flat-scan algorithm is restated here: without an
 * index (tools/build_rag_database.py:51-52 only builds one above 1 M rows) every row is scored and the
 * k smallest distances are returned in ascending order; `_distance` is the squared L2 distance (LanceDB
 * default metric "l2") or 1 - dot for metric "dot" (the metric the reference's index uses).  The
 * Filter order: lancedb's `LanceQueryBuilder.where(where, prefilter=False)` -- the call src/data/rag.py:57-58 makes with 0.14.0's default --
 * applies the filter to the RESULT of the vector search (postfilter = 1 below: the k nearest rows are selected, rows failing the filter
 * are dropped, fewer than k rows may remain); `prefilter=True` removes rows before selection (postfilter = 0).  Both are restated; the
 * caller over-fetches k+3 and keeps k (src/data/datamodule.py:234, src/data/dataset.py:296), so the two orders give the same first 9 rows
 * unless more than 3 of the 12 nearest rows belong to the query's own video.  PARITY UNPINNED against LanceDB itself: the reference holds
 * no test or golden vector for retrieval, and lancedb is not installed here (the default is cited from the 0.14.0 package's
 * `lancedb/query.py`, `def where(self, where: str, prefilter: bool = False)`).
 *
 * Three scoring modes:
 *   mode 0: float32, 16 interleaved fmaf chains + a fixed pairwise tree (chain16 below) -- the exact evaluation
 *           order of the HIP scan kernel (fewer than 16 queries per call), so distances and indices are BIT-EXACT comparable;
 *   mode 1: float64 accumulation (canonical math), used to check that modes 0 and 2 rank the same rows;
 *   mode 2: float32 in the order of the HIP fan-out kernel (16 or more queries per call: the scan as an fp32 matrix product on
 *           v_mfma_f32_32x32x2_f32, whose result is bit for bit a k-ordered fmaf chain): ONE chain per (query, row) over the
 *           features in the order 8c, 8c+4, 8c+1, 8c+5, 8c+2, 8c+6, 8c+3, 8c+7 (c = 0, 1, ..: an MFMA takes one feature from
 *           each half of a 32-byte block); metric "dot": 1 - dot; metric "l2" through the expansion
 *           |q - x|^2 = (|q|^2 + |x|^2) - 2 q.x = fmaf(-2, dot, qq + xx), the squared norms as TWO chains (features 8c + t and
 *           8c + 4 + t, t = 0..3, c ascending) added once (mfma_dot / mfma_sq below).  The expansion SELECTS only: its absolute error is
 *           an ulp of |q|^2 + |x|^2 (a row's distance to itself comes out as ~1e-3 on unnormalised 768-d data, near-duplicates as noise
 *           or negative), so for metric "l2" the kernel's merge step scores the 16 nearest candidates under the expansion (ties by row;
 *           fewer when fewer rows pass a prefilter) AGAIN in the mode-0 form and re-ranks them by (mode-0 distance, row); the first k
 *           are the result (round 6).  Mode 2 restates exactly that; BIT-EXACT comparable as well.
 *   mode 3: the expansion alone (mode 2 without the second scoring) -- what the fan-out kernel returned in round 5; kept so that a test can show
 *           what the second scoring removes.
 * Ties: (distance asc, row asc).  Missing results: row = -1, dist = +inf.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { double d; int32_t r; } cand_t;

static int cand_less(cand_t a, cand_t b) { return a.d < b.d || (a.d == b.d && a.r < b.r); }

/* mode 0: the fp32 evaluation order the HIP kernel uses (motionrag_amd/csrc/topk.hip): 16 interleaved fmaf chains -- chain l
 * runs over k = 64 j + 4 l + c (j = 0.., c = 0..3) -- folded by a fixed pairwise tree p[l] += p[l ^ 8], ^4, ^2, ^1.  (A SIMD
 * flat scan such as lance's also keeps per-lane partial sums and adds them horizontally at the end; the exact lane count of
 * lance's kernels is not pinned, see the header.) */
static float chain16(const float* q, const float* x, int dim, int metric) {
  float p[16];
  for (int l = 0; l < 16; ++l) {
    float acc = 0.0f;
    for (int j = 0; 64 * j + 4 * l < dim; ++j)
      for (int c = 0; c < 4; ++c) {
        const int k = 64 * j + 4 * l + c;
        if (k >= dim) break;
        if (metric == 0) { const float df = q[k] - x[k]; acc = fmaf(df, df, acc); }
        else acc = fmaf(q[k], x[k], acc);
      }
    p[l] = acc;
  }
  for (int m = 8; m >= 1; m >>= 1) {
    float t[16];
    for (int l = 0; l < 16; ++l) t[l] = p[l] + p[l ^ m];
    for (int l = 0; l < 16; ++l) p[l] = t[l];
  }
  return p[0];
}

/* mode 2: the fan-out kernel's orders (see the header) */
static float mfma_dot(const float* q, const float* x, int dim) {
  float acc = 0.0f;
  for (int c = 0; 8 * c < dim; ++c)
    for (int t = 0; t < 4; ++t) {
      const int k0 = 8 * c + t, k1 = 8 * c + 4 + t;
      if (k0 < dim) acc = fmaf(x[k0], q[k0], acc);
      if (k1 < dim) acc = fmaf(x[k1], q[k1], acc);
    }
  return acc;
}
static float mfma_sq(const float* x, int dim) {
  float lo = 0.0f, hi = 0.0f;
  for (int c = 0; 8 * c < dim; ++c)
    for (int t = 0; t < 4; ++t) {
      const int k0 = 8 * c + t, k1 = 8 * c + 4 + t;
      if (k0 < dim) lo = fmaf(x[k0], x[k0], lo);
      if (k1 < dim) hi = fmaf(x[k1], x[k1], hi);
    }
  return lo + hi;
}

static double score(const float* q, const float* x, int dim, int metric, int mode) {
  if (mode == 0) {
    const float acc = chain16(q, x, dim, metric);
    return metric == 0 ? (double)acc : (double)(1.0f - acc);
  } else if (mode == 2 || mode == 3) {
    const float dot = mfma_dot(q, x, dim);
    if (metric != 0) return (double)(1.0f - dot);
    const float t = mfma_sq(q, dim) + mfma_sq(x, dim);
    return (double)fmaf(-2.0f, dot, t);
  } else {
    double acc = 0.0;
    if (metric == 0) { for (int d = 0; d < dim; ++d) { double df = (double)q[d] - (double)x[d]; acc += df * df; } return acc; }
    for (int d = 0; d < dim; ++d) acc += (double)q[d] * (double)x[d];
    return 1.0 - acc;
  }
}

/* out_rows [nq, k] int32, out_dist [nq, k] double */
int topk_oracle(const float* db, const int32_t* group, int64_t n_rows, int dim, const float* queries, const int32_t* exclude,
                int nq, int k, int metric, int mode, int postfilter, int32_t* out_rows, double* out_dist) {
  if (k <= 0 || n_rows <= 0 || nq <= 0) return -1;
  const int rescore = mode == 2 && metric == 0;      /* the fan-out form's second scoring (header, mode 2) */
  const int keep = rescore && k < 16 ? 16 : k;       /* candidates selected by the first score */
  cand_t* best = (cand_t*)malloc(sizeof(cand_t) * (size_t)keep);
  if (!best) return -2;
  for (int qi = 0; qi < nq; ++qi) {
    int nb = 0;
    const float* q = queries + (size_t)qi * dim;
    for (int64_t r = 0; r < n_rows; ++r) {
      if (!postfilter && exclude && group && group[r] == exclude[qi]) continue;
      cand_t c; c.d = score(q, db + (size_t)r * dim, dim, metric, mode); c.r = (int32_t)r;
      if (nb == keep && !cand_less(c, best[keep - 1])) continue;
      int pos = nb < keep ? nb : keep - 1;     /* insertion into the sorted list */
      while (pos > 0 && cand_less(c, best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
      best[pos] = c;
      if (nb < keep) ++nb;
    }
    if (rescore) {                               /* mode-0 distances of the selected candidates, re-ranked by (distance, row); the first k stay */
      for (int j = 0; j < nb; ++j) best[j].d = (double)chain16(q, db + (size_t)best[j].r * dim, dim, 0);
      for (int j = 1; j < nb; ++j) {
        const cand_t c = best[j];
        int pos = j;
        while (pos > 0 && cand_less(c, best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
        best[pos] = c;
      }
      if (nb > k) nb = k;
    }
    if (postfilter && exclude && group) {        /* drop the excluded rows from the selected list; survivors keep their order */
      int w = 0;
      for (int j = 0; j < nb; ++j)
        if (group[best[j].r] != exclude[qi]) best[w++] = best[j];
      nb = w;
    }
    for (int j = 0; j < k; ++j) {
      out_rows[(size_t)qi * k + j] = j < nb ? best[j].r : -1;
      out_dist[(size_t)qi * k + j] = j < nb ? best[j].d : INFINITY;
    }
  }
  free(best);
  return 0;
}
