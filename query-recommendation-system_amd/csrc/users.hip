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

// ---- N4, the clustering features (round 4): StandardScaler + PCA scores on the device ---------------------------------
// recommender.py:226-234: normScores = StandardScaler().fit_transform(ratings); PCA(min(r, c, 200)).fit(.).transform(.)
// on a (users x queries) matrix with users << queries (2000 x 100 000 on the bench shape: 6.9 s of scikit-learn on
// the host, of a step whose device part takes 3 ms).  The PCA scores of a matrix with far more columns than rows come
// from its GRAM matrix: with Z the standardized matrix (column means 0), G = Z Z^T (users x users) = U S^2 U^T and the
// scores are U_k S_k -- a 2000 x 100 000 x 2000 float64 product, the one GEMM-shaped piece of this repository, done on
// the matrix cores (v_mfma_f64_16x16x4_f64) with the standardization fused into the operand staging (the ratings are
// read as the integers they are: Z is never materialised).  The eigen-decomposition of the small G and BIRCH stay with
// library calls (qrlsh/users.py).
//
// column statistics exactly as scikit-learn's StandardScaler computes them (sklearn.utils.extmath.
// _incremental_mean_and_var, first batch): T = sum / n (rows added in order), d = x - T, var = (sum d^2 - (sum d)^2 / n) / n;
// a feature is constant when var <= n eps var + (n mean eps)^2 (sklearn.preprocessing._data._is_constant_feature) and
// keeps scale 1; scale = sqrt(var), and a scale below 10 eps becomes 1 too.  inv_scale = 1 / scale.
__global__ __launch_bounds__(256) void column_stats_kernel(const int32_t *__restrict__ ratings, int64_t nu, int64_t nq,
                                                           double *__restrict__ mean, double *__restrict__ inv_scale) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nq) return;
  double sum = 0.0;
  for (int64_t r = 0; r < nu; ++r) sum += (double)ratings[r * nq + c];
  const double n = (double)nu, T = sum / n;
  double corr = 0.0, ss = 0.0;
  for (int64_t r = 0; r < nu; ++r) {
    const double d = (double)ratings[r * nq + c] - T;
    corr += d;
    ss += d * d;
  }
  double var = (ss - corr * corr / n) / n;
  const double eps = 2.220446049250313e-16;
  const double ub = n * eps * var + (n * T * eps) * (n * T * eps);
  double scale = var <= ub ? 1.0 : sqrt(var);
  if (scale < 10.0 * eps) scale = 1.0;
  mean[c] = T;
  inv_scale[c] = 1.0 / scale;
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int GR_T = 128;          // users per tile side (a workgroup owns a GR_T x GR_T tile of G over one K slice)
constexpr int GR_K = 32;           // columns staged per step
constexpr int GR_RS = GR_T + 1;    // row stride (doubles) of the [k][user] images: the staging writes of a wave land in 32 banks

// partial[s][i][j] (tile pairs ti <= tj only) = sum over the columns of slice s of z_i z_j, z = (x - mean) * inv_scale.
// 256 threads = 4 waves, wave w owns the 64 x 64 quadrant (w >> 1, w & 1) as 4 x 4 MFMA tiles of 16 x 16.
// f64 MFMA operand maps: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15]; D: col = lane & 15,
// row = (lane >> 4) + 4 * reg.
__global__ __launch_bounds__(256, 2) void user_gram_kernel(const int32_t *__restrict__ ratings, int64_t nu, int64_t nq,
                                                           const double *__restrict__ mean,
                                                           const double *__restrict__ inv_scale, int64_t slice_cols,
                                                           int ntile, double *__restrict__ partial) {
  __shared__ double za[GR_K][GR_RS], zb[GR_K][GR_RS];
  // tile pair (ti <= tj) from the linear index
  int ti = 0, rem = blockIdx.x;
  while (rem >= ntile - ti) {
    rem -= ntile - ti;
    ++ti;
  }
  const int tj = ti + rem;
  const int64_t i0 = (int64_t)ti * GR_T, j0 = (int64_t)tj * GR_T;
  const int64_t c_lo = (int64_t)blockIdx.y * slice_cols, c_hi = min(nq, c_lo + slice_cols);
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  const int wi = w >> 1, wj = w & 1;
  const bool diag = ti == tj;
  f64x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int kk = t & (GR_K - 1), r0 = t >> 5;   // staging: thread = (column kk of the step, rows r0, r0 + 8, ...)
  constexpr int RPT = GR_T / (256 / GR_K);      // rows a thread stages per step (16)
  // the ratings of the NEXT step are requested before the current step's MFMAs and standardized into LDS after them:
  // the trip to memory runs behind 128 matrix instructions per wave instead of in front of them
  int32_t xa[RPT], xb[RPT];
  double mu = 0.0, is = 0.0;
  // 32-bit element offsets from the two tile bases (128 rows x nq columns < 2^32: the host checks nq <= 2^24)
  const int32_t *ta = ratings + (size_t)i0 * nq, *tb = ratings + (size_t)j0 * nq;
  const uint32_t rstep = (uint32_t)(256 / GR_K) * (uint32_t)nq, rbase = (uint32_t)r0 * (uint32_t)nq;
  auto request = [&](int64_t c0) {
    const int64_t col = c0 + kk;
    const bool cok = col < c_hi;
    mu = cok ? mean[col] : 0.0;
    is = cok ? inv_scale[col] : 0.0;   // (0 for columns beyond the slice: they stage zeros)
    const uint32_t off = rbase + (uint32_t)col;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = r0 + q * (256 / GR_K);
      xa[q] = (cok && i0 + r < nu) ? ta[off + (uint32_t)q * rstep] : 0;
      xb[q] = (!diag && cok && j0 + r < nu) ? tb[off + (uint32_t)q * rstep] : 0;
    }
  };
  request(c_lo);
  for (int64_t c0 = c_lo; c0 < c_hi; c0 += GR_K) {
    {
      const bool cok = c0 + kk < c_hi;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = r0 + q * (256 / GR_K);
        za[kk][r] = (cok && i0 + r < nu) ? ((double)xa[q] - mu) * is : 0.0;
        if (!diag) zb[kk][r] = (cok && j0 + r < nu) ? ((double)xb[q] - mu) * is : 0.0;
      }
    }
    __syncthreads();
    if (c0 + GR_K < c_hi) request(c0 + GR_K);
    const double (*zbb)[GR_RS] = diag ? za : zb;
#pragma unroll
    for (int ks = 0; ks < GR_K / 4; ++ks) {
      const int k = ks * 4 + (lane >> 4);
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = za[k][wi * 64 + m * 16 + (lane & 15)];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = zbb[k][wj * 64 + n * 16 + (lane & 15)];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
  double *out = partial + (size_t)blockIdx.y * (size_t)nu * (size_t)nu;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = i0 + wi * 64 + m * 16 + (lane >> 4) + 4 * reg, colj = j0 + wj * 64 + n * 16 + (lane & 15);
        if (row < nu && colj < nu) out[row * nu + colj] = acc[m][n][reg];
      }
}

// G[i][j] = G[j][i] = sum over the slices, in slice order (deterministic), of the tile pairs ti <= tj
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ partial, int64_t nu, int nslices,
                                                          double *__restrict__ gram) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nu * nu) return;
  const int64_t i = idx / nu, j = idx % nu;
  const int64_t a = i / GR_T <= j / GR_T ? i : j, b = i / GR_T <= j / GR_T ? j : i;   // the stored orientation
  double s = 0.0;
  for (int k = 0; k < nslices; ++k) s += partial[(size_t)k * nu * nu + a * nu + b];
  gram[idx] = s;
}

QRLSH_EXPORT size_t qrlsh_user_gram_workspace_bytes(int64_t nu, int64_t nq) {
  if (nu <= 0 || nq <= 0) return 16;
  const int64_t slices = nq >= 65536 ? 16 : nq >= 4096 ? 4 : 1;
  return (size_t)slices * (size_t)nu * (size_t)nu * sizeof(double);
}

QRLSH_EXPORT int qrlsh_user_gram(const int32_t *ratings, int64_t nu, int64_t nq, double *mean_out, double *inv_scale_out,
                                 double *gram_out, void *workspace, size_t workspace_bytes, void *stream) {
  QR_CHECK_ARG(nu > 0 && nq > 0 && nu < (1ll << 20) && nq <= (1ll << 24), "qrlsh_user_gram: bad sizes nu=%lld nq=%lld",
               (long long)nu, (long long)nq);
  QR_CHECK_ARG(ratings && mean_out && inv_scale_out && gram_out && workspace, "qrlsh_user_gram: null pointer");
  if (workspace_bytes < qrlsh_user_gram_workspace_bytes(nu, nq)) {
    qrlsh_set_error("qrlsh_user_gram: workspace %zu < %zu bytes", workspace_bytes, qrlsh_user_gram_workspace_bytes(nu, nq));
    return QRLSH_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int slices = nq >= 65536 ? 16 : nq >= 4096 ? 4 : 1;
  const int64_t slice_cols = ceil_div64(ceil_div64(nq, slices), GR_K) * GR_K;
  const int ntile = (int)ceil_div64(nu, GR_T);
  QR_LAUNCH("user_colstats", column_stats_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, st, ratings, nu, nq,
            mean_out, inv_scale_out);
  QR_LAUNCH("user_gram", user_gram_kernel, dim3((unsigned)(ntile * (ntile + 1) / 2), (unsigned)slices), dim3(256), 0, st,
            ratings, nu, nq, (const double *)mean_out, (const double *)inv_scale_out, slice_cols, ntile,
            static_cast<double *>(workspace));
  QR_LAUNCH("user_gram_reduce", gram_reduce_kernel, dim3((unsigned)ceil_div64(nu * nu, 256)), dim3(256), 0, st,
            (const double *)workspace, nu, slices, gram_out);
  QR_LAUNCH_CHECK("qrlsh_user_gram");
  return QRLSH_OK;
}
