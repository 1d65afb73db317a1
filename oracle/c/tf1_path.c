/*
 * tf1_path.c — C restatement of the TF1 CPU execution of Recommender.messagePropagate
 * (reference model.py:80-92). TEST / BASELINE INFRASTRUCTURE ONLY: used by tests/ as a second
 * oracle and by bench.py's cpu_baseline leg ("port"); never linked into libsagnn.so.
 *
 * It keeps TF's two-op structure on purpose — that is what the reference pays for on a CPU:
 *   1. GatherV2      G[e, :] = src[col(e), :]        materialises [nnz, d]   (model.py:86)
 *   2. SegmentSum    S[r, :] = sum_{e: row(e)=r} G[e, :], rows = max(row)+1, each segment
 *                    accumulated sequentially in edge order                 (model.py:87)
 *   3. Pad(+100 rows), GatherV2(range(n_out)), Maximum(leaky*x, x)          (model.py:87-92)
 * TF 1.14 runs GatherV2 sharded over its intra-op pool and SegmentSum on one thread; here both
 * loops are OpenMP-parallel (segments are never split across threads, so every row is still
 * summed in edge order) and the caller states the thread count it used.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* indices: [nnz, 2] int32 (row, col) sorted by row, as transToLsts returns them
 * (DataHandler.py:47-69). scratch: [nnz * d] floats. out: [n_out * d] floats.
 * returns 0, or -1 if n_out > max(row)+1+100 (TF-CPU raises InvalidArgument there) when
 * strict != 0; with strict == 0 the missing rows are zero-filled (what TF-GPU returns). */
int tf1_message_propagate(const int32_t* indices, int64_t nnz, const float* src, int64_t d,
                          int64_t n_out, float leaky, float* out, float* scratch, int threads,
                          int strict) {
  if (threads > 0) omp_set_num_threads(threads);
  /* 1. GatherV2 */
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < nnz; ++e)
    memcpy(scratch + e * d, src + (int64_t)indices[2 * e + 1] * d, (size_t)d * sizeof(float));

  const int64_t seg_rows = nnz > 0 ? (int64_t)indices[2 * (nnz - 1)] + 1 : 0;
  if (strict && n_out > seg_rows + 100) return -1;

  /* 2+3. SegmentSum into the output rows, then leaky. Rows >= seg_rows are the pad. */
  memset(out, 0, (size_t)n_out * (size_t)d * sizeof(float));
#pragma omp parallel
  {
    const int nt = omp_get_num_threads(), me = omp_get_thread_num();
    int64_t lo = nnz * me / nt, hi = nnz * (me + 1) / nt;
    /* move both cuts forward to the next segment start so no segment is shared */
    while (lo > 0 && lo < nnz && indices[2 * lo] == indices[2 * (lo - 1)]) ++lo;
    while (hi > 0 && hi < nnz && indices[2 * hi] == indices[2 * (hi - 1)]) ++hi;
    for (int64_t e = lo; e < hi; ++e) {
      const int64_t r = indices[2 * e];
      if (r >= n_out) continue;
      float* o = out + r * d;
      const float* g = scratch + e * d;
      for (int64_t k = 0; k < d; ++k) o[k] += g[k];
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_out * d; ++i) {
    const float x = out[i], y = leaky * x;
    out[i] = y > x ? y : x;
  }
  return 0;
}

int tf1_max_threads(void) { return omp_get_max_threads(); }
