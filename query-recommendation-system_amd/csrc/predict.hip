// predict.hip -- N1: the hybrid prediction loop (consumer of the hot path's output).
//
// Reference: Recommender.compute_scores, recommender.py:301-331, and weighted_average,
// recommender.py:36-47.  Every zero cell (user i, query j) of the utility matrix gets
//     qp = weighted_average(ratings[i],    query_sims[j])      (content-based, :313-317)
//     up = weighted_average(ratings[:, j], user_sims[i])       (collaborative,  :320)
// blended by the rules of :324-331 and rounded with Python's round() (half to even).
// One thread per cell; every cell is independent.  The arithmetic is kept bit-compatible with
// the reference's float64 evaluation: this file is compiled with -ffp-contract=off, sums follow
// numpy's pairwise order (what np.sum does in weighted_average), the blend is evaluated in
// the reference's operand order, and rint() is round-half-to-even like round().
#include "common.h"

constexpr int PRED_MAXK = 64;  // longest neighbour list handled (K = round(log_1.5 n) <= 51 for n < 1e9)

// numpy's float64 pairwise_sum for n <= 128 (loops_utils.h.src)
__device__ static inline double np_sum_order(const double *a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += a[i];
  return res;
}

// weighted_average (recommender.py:36-47): rating[k] = the user's / query's rating of neighbour k
__device__ static inline double weighted_average(const int32_t *rating, const double *sims, int n) {
  double prod[PRED_MAXK], w[PRED_MAXK];
  int nw = 0;
  for (int k = 0; k < n; ++k) {
    prod[k] = (double)rating[k] * sims[k];
    if (rating[k] != 0) w[nw++] = sims[k];
  }
  const double wsum = np_sum_order(w, nw);
  if (wsum == 0.0) return 0.0;
  return np_sum_order(prod, n) / wsum;
}

__global__ __launch_bounds__(256) void predict_kernel(const int32_t *__restrict__ ratings, int64_t nu, int64_t nq,
                                                      const int64_t *__restrict__ q_off,
                                                      const int32_t *__restrict__ q_idx,
                                                      const double *__restrict__ q_val,
                                                      const int32_t *__restrict__ u_idx,
                                                      const double *__restrict__ u_val, int ku, double qw, double uw,
                                                      double dmean, int32_t *__restrict__ out) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= nu * nq) return;
  const int64_t i = cell / nq, j = cell - i * nq;
  const int32_t own = ratings[cell];
  if (own != 0) {
    out[cell] = own;
    return;
  }
  int32_t rt[PRED_MAXK];
  double sv[PRED_MAXK];
  // query side: neighbours of query j, this user's ratings of them
  const int64_t lo = q_off[j];
  int n = (int)(q_off[j + 1] - lo);
  if (n > PRED_MAXK) n = PRED_MAXK;
  for (int k = 0; k < n; ++k) {
    rt[k] = ratings[i * nq + q_idx[lo + k]];
    sv[k] = q_val[lo + k];
  }
  const double qp = weighted_average(rt, sv, n);
  // user side: neighbours of user i, their ratings of query j
  int m = 0;
  for (int k = 0; k < ku && k < PRED_MAXK; ++k) {
    const int32_t u = u_idx[i * ku + k];
    if (u < 0) break;  // -1 padding
    rt[m] = ratings[(int64_t)u * nq + j];
    sv[m] = u_val[i * ku + k];
    ++m;
  }
  const double up = weighted_average(rt, sv, m);
  double r;
  if (up == 0.0 && qp == 0.0) r = 0.0;
  else if (up == 0.0) r = qp * (qw + (uw * 0.5)) + dmean * (uw * 0.5);
  else if (qp == 0.0) r = up * (uw + (qw * 0.5)) + dmean * (qw * 0.5);
  else r = qp * qw + up * uw;
  out[cell] = (int32_t)rint(r);
}

QRLSH_EXPORT int qrlsh_predict(const int32_t *ratings, int64_t nu, int64_t nq, const int64_t *q_off,
                               const int32_t *q_idx, const double *q_val, const int32_t *u_idx, const double *u_val,
                               int32_t ku, double query_weight, double user_weight, double default_mean,
                               int32_t *out, void *stream) {
  QR_CHECK_ARG(nu >= 0 && nq >= 0 && ku >= 0 && ku <= PRED_MAXK, "qrlsh_predict: bad sizes nu=%lld nq=%lld ku=%d",
               (long long)nu, (long long)nq, ku);
  if (nu == 0 || nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(ratings && q_off && out && (ku == 0 || (u_idx && u_val)), "qrlsh_predict: null pointer");
  QR_LAUNCH("predict", predict_kernel, dim3((unsigned)ceil_div64(nu * nq, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), ratings, nu, nq, q_off, q_idx, q_val, u_idx, u_val, ku, query_weight,
            user_weight, default_mean, out);
  QR_LAUNCH_CHECK("qrlsh_predict");
  return QRLSH_OK;
}
