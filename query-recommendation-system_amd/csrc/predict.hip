// predict.hip -- N1: the hybrid prediction loop (consumer of the hot path's output).
//
// Reference: Recommender.compute_scores, recommender.py:301-331, and weighted_average,
// recommender.py:36-47.  Every zero cell (user i, query j) of the utility matrix gets
//     qp = weighted_average(ratings[i],    query_sims[j])      (content-based, :313-317)
//     up = weighted_average(ratings[:, j], user_sims[i])       (collaborative,  :320)
// blended by the rules of :324-331 and rounded with Python's round() (half to even).
// One thread per cell; every cell is independent.  float64 throughout, compiled with
// -ffp-contract=off, the blend evaluated in the reference's operand order, rint() = round-half-to-even.
//
// Order of the two sums inside weighted_average (np.sum(ur * vals) and np.sum(vals[ur != 0])):
//   QRLSH_SUM_PAIRWISE   numpy's pairwise_sum (8 running sums, combined as a tree, tail added one by one):
//                        what np.sum does when weighted_average runs as plain Python.  The golden fixtures
//                        (tests/golden/cfg1*_scores.npz) were captured that way -- numba is not installable
//                        here, tools/make_golden.py replaces @jit by the identity -- so THIS order is the one
//                        parity is pinned for.
//   QRLSH_SUM_SEQUENTIAL one accumulator, index order: what numba's nopython np.sum compiles to, i.e. what
//                        the reference computes where numba is installed.  Unpinned (no fixture can be made
//                        here); differs from the pairwise order by an ulp of the sums for lists of 8 or more,
//                        which can flip a cell whose blend lies exactly on .5.
// No per-thread arrays: the products are accumulated while the ratings are gathered (the position of a
// product in the sum is its neighbour index), the non-zero flags of the ratings are kept as one 64-bit mask,
// and the weight sum walks the mask's set bits (the position of a weight in the compacted list is its rank
// in the mask) -- so a neighbour list of up to 64 entries needs 8 accumulators, not four 64-entry arrays.
#include "common.h"

constexpr int PRED_MAXK = 64;  // longest neighbour list handled (K = round(log_1.5 n) <= 51 for n < 1e9)

struct Pairwise8 {
  // numpy's float64 pairwise_sum for n <= 128 (loops_utils.h.src), fed one element at a time: element t of n.
  // n < 8: plain running sum.  Otherwise elements [0, n - n % 8) go round-robin into 8 accumulators, which
  // are combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) when the first tail element (or the
  // end) arrives; tail elements are then added one by one.
  double r0, r1, r2, r3, r4, r5, r6, r7, res;
  int n, body, t;
  bool seq;
  __device__ void begin(int n_, bool sequential) {
    n = n_;
    seq = sequential || n_ < 8;
    body = n_ - (n_ % 8);
    t = 0;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = 0.0;
    res = 0.0;
  }
  __device__ void add(double v) {
    if (seq) {
      res += v;
    } else if (t < body) {
      const bool first = t < 8;
      switch (t & 7) {
        case 0: r0 = first ? v : r0 + v; break;
        case 1: r1 = first ? v : r1 + v; break;
        case 2: r2 = first ? v : r2 + v; break;
        case 3: r3 = first ? v : r3 + v; break;
        case 4: r4 = first ? v : r4 + v; break;
        case 5: r5 = first ? v : r5 + v; break;
        case 6: r6 = first ? v : r6 + v; break;
        default: r7 = first ? v : r7 + v; break;
      }
      if (t + 1 == body) res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    } else {
      res += v;
    }
    ++t;
  }
  // elements [k0, k0 + 8) of the list at once (k0 a multiple of 8; v[u] of elements past n are ignored): the
  // accumulator of element k0 + u is r<u>, known at compile time -- no per-element dispatch
  __device__ void add8(const double (&v)[8], int k0) {
    if (!seq && k0 < body) {   // a whole chunk of the round-robin part (body is a multiple of 8)
      if (k0 == 0) {
        r0 = v[0]; r1 = v[1]; r2 = v[2]; r3 = v[3]; r4 = v[4]; r5 = v[5]; r6 = v[6]; r7 = v[7];
      } else {
        r0 += v[0]; r1 += v[1]; r2 += v[2]; r3 += v[3]; r4 += v[4]; r5 += v[5]; r6 += v[6]; r7 += v[7];
      }
      if (k0 + 8 == body) res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < n) res += v[u];
    }
    t = k0 + 8;
  }
  __device__ double end() const { return res; }
};

// weighted_average (recommender.py:36-47) over a neighbour list of n entries: rating(k) = the user's / query's
// rating of neighbour k, sim(k) its similarity.  RatingAt / SimAt are callables (gathers).
template <typename RatingAt, typename SimAt>
__device__ static inline double weighted_average(int n, bool sequential, RatingAt rating, SimAt sim) {
  if (n == 0) return 0.0;
  Pairwise8 prod;
  prod.begin(n, sequential);
  unsigned long long nz = 0;
  // eight neighbours at a time: their list entries, then their ratings, are independent loads in flight together
  // (a dependent index -> rating round trip per neighbour otherwise); the additions keep the list order
  for (int k0 = 0; k0 < n; k0 += 8) {
    int32_t r[8];
    double sv[8], pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      r[u] = 0;
      sv[u] = 0.0;
      if (k0 + u < n) {
        sv[u] = sim(k0 + u);
        r[u] = rating(k0 + u);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pv[u] = (double)r[u] * sv[u];
      if (r[u] != 0) nz |= 1ull << (k0 + u);   // r[u] = 0 past the end
    }
    prod.add8(pv, k0);
  }
  // the weights of the rated neighbours, in list order, eight at a time as well
  Pairwise8 w;
  const int m = __popcll(nz);
  w.begin(m, sequential);
  for (int k0 = 0; k0 < m; k0 += 8) {
    double wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      wv[u] = 0.0;
      if (nz) {
        wv[u] = sim(__ffsll((long long)nz) - 1);
        nz &= nz - 1;
      }
    }
    w.add8(wv, k0);
  }
  const double wsum = w.end();
  if (wsum == 0.0) return 0.0;
  return prod.end() / wsum;
}

// TL: the query neighbour lists come transposed and padded, idx_t / val_t [kq][nq] (entry k of query j at
// k * nq + j; predict_lists_transpose_kernel): consecutive lanes = consecutive queries then read consecutive words,
// where the CSR form costs a separate cache line per lane for every list entry (q_idx / q_val = idx_t / val_t).
template <bool TL>
__global__ __launch_bounds__(256) void predict_kernel(const int32_t *__restrict__ ratings, int64_t nu, int64_t nq,
                                                      const int64_t *__restrict__ q_off,
                                                      const int32_t *__restrict__ q_idx,
                                                      const double *__restrict__ q_val,
                                                      const int32_t *__restrict__ u_idx,
                                                      const double *__restrict__ u_val, int ku, double qw, double uw,
                                                      double dmean, int sequential, int nlimit,
                                                      int32_t *__restrict__ out) {
  // consecutive lanes = consecutive queries of ONE user: the user-side gathers ratings[u][j] are coalesced
  // across the wave and the user's neighbour list is wave-uniform; the query-side gathers stay inside the
  // user's own row (4 nq bytes, cache-resident while the row's workgroups run)
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= nu * nq) return;
  const int64_t i = cell / nq, j = cell - i * nq;
  const int32_t own = ratings[cell];
  if (own != 0) {
    out[cell] = own;
    return;
  }
  // query side: neighbours of query j, this user's ratings of them
  const int64_t lo = q_off[j];
  const int64_t n64 = q_off[j + 1] - lo;
  // a list longer than the limit (PRED_MAXK entries, or the kq rows of the transposed workspace) is never walked:
  // predict_check_kernel has raised *too_long_out for it, the cell gets 0 and the caller discards the result
  if (n64 < 0 || n64 > nlimit) {
    out[cell] = 0;
    return;
  }
  const int n = (int)n64;
  const int32_t *row = ratings + i * nq;
  const double qp = TL ? weighted_average(
                             n, sequential != 0, [&](int k) { return row[q_idx[(int64_t)k * nq + j]]; },
                             [&](int k) { return q_val[(int64_t)k * nq + j]; })
                       : weighted_average(
                             n, sequential != 0, [&](int k) { return row[q_idx[lo + k]]; },
                             [&](int k) { return q_val[lo + k]; });
  // user side: neighbours of user i (-1 padded), their ratings of query j
  const int32_t *ui = u_idx + i * ku;
  const double *uv = u_val + i * ku;
  int m = 0;
  while (m < ku && ui[m] >= 0) ++m;
  const double up = weighted_average(
      m, sequential != 0, [&](int k) { return ratings[(int64_t)ui[k] * nq + j]; }, [&](int k) { return uv[k]; });
  double r;
  if (up == 0.0 && qp == 0.0) r = 0.0;
  else if (up == 0.0) r = qp * (qw + (uw * 0.5)) + dmean * (uw * 0.5);
  else if (qp == 0.0) r = up * (uw + (qw * 0.5)) + dmean * (qw * 0.5);
  else r = qp * qw + up * uw;
  out[cell] = (int32_t)rint(r);
}

// longest query neighbour list (host pre-check of the 64-entry limit without a read-back: the kernel below
// raises a device flag, the caller reads it together with the result)
__global__ __launch_bounds__(256) void predict_check_kernel(const int64_t *__restrict__ q_off, int64_t nq, int maxlen,
                                                            uint32_t *__restrict__ too_long) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nq && q_off[j + 1] - q_off[j] > maxlen) atomicOr(too_long, 1u);
}

// CSR neighbour lists -> [kq][nq], coalesced writes (one thread per output word; -1 / 0.0 past a list's end)
__global__ __launch_bounds__(256) void predict_lists_transpose_kernel(const int64_t *__restrict__ q_off,
                                                                      const int32_t *__restrict__ q_idx,
                                                                      const double *__restrict__ q_val, int64_t nq,
                                                                      int kq, int32_t *__restrict__ idx_t,
                                                                      double *__restrict__ val_t) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)kq * nq) return;
  const int64_t k = t / nq, j = t - k * nq;
  const int64_t lo = q_off[j];
  const bool in = k < q_off[j + 1] - lo;
  idx_t[t] = in ? q_idx[lo + k] : -1;
  val_t[t] = in ? q_val[lo + k] : 0.0;
}

QRLSH_EXPORT size_t qrlsh_predict_workspace_bytes(int64_t nq, int32_t kq) {
  if (nq <= 0 || kq <= 0) return 0;
  return (size_t)kq * (size_t)nq * (sizeof(double) + sizeof(int32_t)) + 16;
}

QRLSH_EXPORT int qrlsh_predict(const int32_t *ratings, int64_t nu, int64_t nq, const int64_t *q_off,
                               const int32_t *q_idx, const double *q_val, const int32_t *u_idx, const double *u_val,
                               int32_t ku, double query_weight, double user_weight, double default_mean,
                               int32_t sum_order, int32_t *out, uint32_t *too_long_out, int32_t kq, void *workspace,
                               size_t workspace_bytes, void *stream) {
  QR_CHECK_ARG(nu >= 0 && nq >= 0 && ku >= 0 && ku <= PRED_MAXK, "qrlsh_predict: bad sizes nu=%lld nq=%lld ku=%d (<= %d)",
               (long long)nu, (long long)nq, ku, PRED_MAXK);
  QR_CHECK_ARG(sum_order == QRLSH_SUM_PAIRWISE || sum_order == QRLSH_SUM_SEQUENTIAL, "qrlsh_predict: bad sum_order %d",
               sum_order);
  QR_CHECK_ARG(too_long_out, "qrlsh_predict: too_long_out is required");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(too_long_out, 0, sizeof(uint32_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_predict: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  if (nu == 0 || nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(ratings && q_off && out && (ku == 0 || (u_idx && u_val)), "qrlsh_predict: null pointer");
  const bool tl = kq > 0 && workspace != nullptr;
  QR_CHECK_ARG(kq >= 0 && kq <= PRED_MAXK && (!tl || workspace_bytes >= qrlsh_predict_workspace_bytes(nq, kq)),
               "qrlsh_predict: kq=%d (<= %d) needs %zu workspace bytes, got %zu", kq, PRED_MAXK,
               qrlsh_predict_workspace_bytes(nq, kq), workspace_bytes);
  QR_LAUNCH("predict_check", predict_check_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, st, q_off, nq,
            tl ? (int)kq : PRED_MAXK, too_long_out);
  if (tl) {
    // lists transposed once ([kq][nq], doubles first: 8-byte aligned), then read coalesced by every user's sweep
    double *val_t = static_cast<double *>(workspace);
    int32_t *idx_t = reinterpret_cast<int32_t *>(val_t + (size_t)kq * nq);
    QR_LAUNCH("predict_lists", predict_lists_transpose_kernel, dim3((unsigned)ceil_div64((int64_t)kq * nq, 256)),
              dim3(256), 0, st, q_off, q_idx, q_val, nq, (int)kq, idx_t, val_t);
    QR_LAUNCH("predict", predict_kernel<true>, dim3((unsigned)ceil_div64(nu * nq, 256)), dim3(256), 0, st, ratings, nu,
              nq, q_off, (const int32_t *)idx_t, (const double *)val_t, u_idx, u_val, ku, query_weight, user_weight,
              default_mean, (int)(sum_order == QRLSH_SUM_SEQUENTIAL), (int)kq, out);
  } else {
    QR_LAUNCH("predict", predict_kernel<false>, dim3((unsigned)ceil_div64(nu * nq, 256)), dim3(256), 0, st, ratings, nu,
              nq, q_off, q_idx, q_val, u_idx, u_val, ku, query_weight, user_weight, default_mean,
              (int)(sum_order == QRLSH_SUM_SEQUENTIAL), PRED_MAXK, out);
  }
  QR_LAUNCH_CHECK("qrlsh_predict");
  return QRLSH_OK;
}
