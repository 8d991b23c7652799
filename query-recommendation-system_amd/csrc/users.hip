// users.hip -- N4: the deterministic half of Recommender.compute_userSimilarities (recommender.py:263-288).
//
// The reference clusters the users with scikit-learn (StandardScaler -> PCA -> BIRCH, :226-261; host side here
// too: it is the reference's own library call) and then, inside every cluster, centres each user's row on the
// mean of its non-zero ratings and takes the cosine of every pair of rows.  The centring happens IN PLACE IN AN
// INTEGER ARRAY (:268-272: np.array(self.ratings[...]) keeps the integer dtype), so the centred values are
// truncated toward zero -- reproduced here exactly: mean = (double)sum / (double)count (np.mean of integers is an
// exact sum divided once), c = (int)((double)x - mean).  With integer rows the rest is the hot path's own
// machinery: pairs of users that share a cluster label are candidate pairs of a one-band bucket structure
// (qrlsh_bucket_pairs_emit on the labels), their cosine is qrlsh_score_pairs on the centred rows (exact integer
// dot, float64 divide, rint(1000 cos)), and the per-user cut is the top-K of qrlsh_topk_select_*.
#include "common.h"

// one workgroup per user row (a row of 100 000 ratings is 400 KB: one wave walking it alone is latency-bound):
// non-zero mean, then the truncated centred row (zeros stay zero), row stride nq_stride (the padding columns are
// written as zeros)
__global__ __launch_bounds__(256) void center_rows_kernel(const int32_t *__restrict__ ratings, int64_t nu, int64_t nq,
                                                          int64_t nq_stride, int32_t *__restrict__ out) {
  __shared__ int64_t ssum[256 / WAVE], scnt[256 / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int64_t u = blockIdx.x;
  const int32_t *row = ratings + u * nq;
  int64_t sum = 0, cnt = 0;
  for (int64_t c = threadIdx.x; c < nq; c += 256) {
    const int32_t x = row[c];
    sum += x;
    cnt += x != 0;
  }
#pragma unroll
  for (int m = 1; m < WAVE; m <<= 1) {
    sum += __shfl_xor(sum, m, WAVE);
    cnt += __shfl_xor(cnt, m, WAVE);
  }
  if (lane == 0) {
    ssum[w] = sum;
    scnt[w] = cnt;
  }
  __syncthreads();
  sum = cnt = 0;
#pragma unroll
  for (int i = 0; i < 256 / WAVE; ++i) {  // exact integers: any order
    sum += ssum[i];
    cnt += scnt[i];
  }
  const double mean = cnt ? (double)sum / (double)cnt : 0.0;
  int32_t *dst = out + u * nq_stride;
  for (int64_t c = threadIdx.x; c < nq_stride; c += 256) {
    int32_t v = 0;
    if (c < nq) {
      const int32_t x = row[c];
      if (x != 0) v = (int32_t)((double)x - mean);  // float64 -> integer assignment truncates toward zero
    }
    dst[c] = v;
  }
}

QRLSH_EXPORT int qrlsh_center_rows(const int32_t *ratings, int64_t nu, int64_t nq, int64_t nq_stride, int32_t *out,
                                   void *stream) {
  QR_CHECK_ARG(nu >= 0 && nq >= 0 && nq_stride >= nq, "qrlsh_center_rows: bad sizes nu=%lld nq=%lld stride=%lld",
               (long long)nu, (long long)nq, (long long)nq_stride);
  if (nu == 0 || nq_stride == 0) return QRLSH_OK;
  QR_CHECK_ARG(ratings && out, "qrlsh_center_rows: null pointer");
  QR_CHECK_ARG(nu < (1ll << 31), "qrlsh_center_rows: nu=%lld", (long long)nu);
  QR_LAUNCH("center_rows", center_rows_kernel, dim3((unsigned)nu), dim3(256), 0,
            static_cast<hipStream_t>(stream), ratings, nu, nq, nq_stride, out);
  QR_LAUNCH_CHECK("qrlsh_center_rows");
  return QRLSH_OK;
}
