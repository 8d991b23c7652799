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
                                                      int32_t *__restrict__ out,
                                                      const uint32_t *__restrict__ only_if = nullptr) {
  if (only_if && *only_if == 0) return;  // the tile form has done the work
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

// Row form of the sweep (transposed lists only): one 1024-thread workgroup per (user, slice of the user's cells).
// Every query-side gather of a cell reads the SAME user's row -- ratings[i][q_idx[..]], a random word of a
// 4 nq-byte row per neighbour, each its own trip to L2 in the kernel above.  Ratings are small non-negative
// integers (0 .. 100 in the reference's data), so the workgroup first stages its user's whole row in LDS as BYTES
// (nq <= 128 K queries; a value outside 0 .. 255 anywhere in the row makes the workgroup read the row from memory
// instead -- same results) and the neighbour gathers become LDS byte reads.  The user-side gathers
// (ratings[u_k][j], consecutive lanes = consecutive j) stay coalesced global reads.
constexpr int PR_THREADS = 1024;
constexpr int PR_MAXQ = 128 * 1024;
__global__ __launch_bounds__(PR_THREADS) void predict_row_kernel(const int32_t *__restrict__ ratings, int64_t nu,
                                                                 int64_t nq, const int64_t *__restrict__ q_off,
                                                                 const int32_t *__restrict__ idx_t,
                                                                 const double *__restrict__ val_t,
                                                                 const int32_t *__restrict__ u_idx,
                                                                 const double *__restrict__ u_val, int ku, double qw,
                                                                 double uw, double dmean, int sequential, int nlimit,
                                                                 int32_t *__restrict__ out,
                                                                 const uint32_t *__restrict__ only_if = nullptr) {
  __shared__ uint8_t lrow[PR_MAXQ];
  __shared__ int wide;
  if (only_if && *only_if == 0) return;  // the tile form has done the work
  const int64_t i = blockIdx.y;
  const int t = threadIdx.x;
  const int32_t *row = ratings + i * nq;
  if (t == 0) wide = 0;
  __syncthreads();
  bool bad = false;
  for (int64_t j = t; j < nq; j += PR_THREADS) {
    const int32_t v = row[j];
    bad |= (v < 0) | (v > 255);
    lrow[j] = (uint8_t)v;
  }
  if (bad) wide = 1;
  __syncthreads();
  const bool in_lds = wide == 0;  // uniform
  const int32_t *ui = u_idx + i * ku;
  const double *uv = u_val + i * ku;
  int m = 0;
  while (m < ku && ui[m] >= 0) ++m;  // uniform: the user's own neighbour list
  // this workgroup's slice of the row
  const int64_t per = (nq + gridDim.x - 1) / gridDim.x;
  const int64_t j0 = (int64_t)blockIdx.x * per, j1 = min(nq, j0 + per);
  for (int64_t j = j0 + t; j < j1; j += PR_THREADS) {
    const int64_t cell = i * nq + j;
    const int32_t own = in_lds ? (int32_t)lrow[j] : row[j];
    if (own != 0) {
      out[cell] = own;
      continue;
    }
    const int64_t n64 = q_off[j + 1] - q_off[j];
    if (n64 < 0 || n64 > nlimit) {  // flagged by predict_check_kernel; never walked
      out[cell] = 0;
      continue;
    }
    const double qp = weighted_average(
        (int)n64, sequential != 0,
        [&](int k) {
          const int32_t q = idx_t[(int64_t)k * nq + j];
          return in_lds ? (int32_t)lrow[q] : row[q];
        },
        [&](int k) { return val_t[(int64_t)k * nq + j]; });
    const double up = weighted_average(
        m, sequential != 0, [&](int k) { return ratings[(int64_t)ui[k] * nq + j]; }, [&](int k) { return uv[k]; });
    double r;
    if (up == 0.0 && qp == 0.0) r = 0.0;
    else if (up == 0.0) r = qp * (qw + (uw * 0.5)) + dmean * (uw * 0.5);
    else if (qp == 0.0) r = up * (uw + (qw * 0.5)) + dmean * (qw * 0.5);
    else r = qp * qw + up * uw;
    out[cell] = (int32_t)rint(r);
  }
}

// ---- tile form ------------------------------------------------------------------------------------------
// Both sweeps above walk a cell's two neighbour lists with per-lane loads: every user's pass over the row re-reads
// ALL query lists (kq x nq x 12 bytes, 2000 times on the bench shape: 67 GB out of L2) -- the lists, not the ratings,
// are what those kernels move.  Here the matrix is first transposed to BYTES, rt[q][u] (ratings are small
// non-negative integers; a value outside 0 .. 255 raises a flag and the row form runs instead), and a workgroup
// owns a tile of 64 users x 16 queries: wave w = query j0 + w, lane = user u0 + lane.  Then
//   * a wave's query list (index, similarity) is the same for all its lanes: read once from LDS, broadcast;
//   * the query-side gathers ratings[u][idx] = rt[idx][u0 + lane] are 64 consecutive bytes per neighbour;
//   * the user-side gathers ratings[u_k][j] = rt[j][u_k] stay inside ONE 'nu'-byte row of rt per wave;
//   * every list is read once per tile (64 users' lists, 16 queries' lists: LDS), not once per cell;
//   * the output tile is turned in LDS and written as 64-byte row segments.
// The arithmetic per cell is the same weighted_average, in the same order.
constexpr int PT_USERS = 64;
constexpr int PT_QUERIES = 16;
constexpr int PT_THREADS = PT_USERS * PT_QUERIES;
constexpr int PT_MAXKU = 32;   // user lists beyond this take the row form

// ratings int32 [nu][nq] -> rt uint8 [nq][nus] (64 x 64 tiles through LDS: coalesced both ways); *wide |= 1 when a
// value does not fit a byte
__global__ __launch_bounds__(256) void predict_transpose_u8_kernel(const int32_t *__restrict__ ratings, int64_t nu,
                                                                   int64_t nq, int64_t nus, uint8_t *__restrict__ rt,
                                                                   uint32_t *__restrict__ wide) {
  __shared__ uint8_t tile[64][65];
  const int64_t q0 = (int64_t)blockIdx.x * 64, u0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  bool bad = false;
  for (int r = ty; r < 64; r += 4) {
    const int64_t u = u0 + r, q = q0 + tx;
    int32_t v = 0;
    if (u < nu && q < nq) v = ratings[u * nq + q];
    bad |= (v < 0) | (v > 255);
    tile[r][tx] = (uint8_t)v;
  }
  if (bad) atomicOr(wide, 1u);
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t q = q0 + r, u = u0 + tx;
    if (q < nq && u < nus) rt[q * nus + u] = tile[tx][r];
  }
}

__global__ __launch_bounds__(PT_THREADS, 8) void predict_tile_kernel(const uint8_t *__restrict__ rt, int64_t nus,
                                                                  const int32_t *__restrict__ ratings, int64_t nu,
                                                                  int64_t nq, const int64_t *__restrict__ q_off,
                                                                  const int32_t *__restrict__ q_idx,
                                                                  const double *__restrict__ q_val,
                                                                  const int32_t *__restrict__ u_idx,
                                                                  const double *__restrict__ u_val, int ku, double qw,
                                                                  double uw, double dmean, int sequential, int nlimit,
                                                                  const uint32_t *__restrict__ wide,
                                                                  int32_t *__restrict__ out) {
  __shared__ double lval[PT_QUERIES][PRED_MAXK];
  __shared__ int32_t lidx[PT_QUERIES][PRED_MAXK];
  __shared__ double uval[PT_USERS][PT_MAXKU + 1];
  __shared__ int32_t uidx[PT_USERS][PT_MAXKU + 1];
  __shared__ int32_t otile[PT_USERS][PT_QUERIES + 1];
  if (*wide) return;  // uniform: some rating does not fit a byte, the row form (launched next) does the work
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  const int64_t j0 = (int64_t)blockIdx.x * PT_QUERIES, u0 = (int64_t)blockIdx.y * PT_USERS;
  // the tile's lists: 16 query lists (CSR) and 64 user lists (padded [nu][ku]), each read once
  for (int e = t; e < PT_QUERIES * PRED_MAXK; e += PT_THREADS) {
    const int jj = e / PRED_MAXK, k = e - jj * PRED_MAXK;
    const int64_t j = j0 + jj;
    int32_t ix = 0;
    double vv = 0.0;
    if (j < nq) {
      const int64_t lo = q_off[j], n = q_off[j + 1] - lo;
      if (k < n && n <= nlimit) {
        ix = q_idx[lo + k];
        vv = q_val[lo + k];
      }
    }
    lidx[jj][k] = ix;
    lval[jj][k] = vv;
  }
  for (int e = t; e < PT_USERS * PT_MAXKU; e += PT_THREADS) {
    const int ul = e / PT_MAXKU, k = e - ul * PT_MAXKU;
    const int64_t u = u0 + ul;
    int32_t ix = -1;
    double vv = 0.0;
    if (u < nu && k < ku) {
      ix = u_idx[u * ku + k];
      vv = u_val[u * ku + k];
    }
    uidx[ul][k] = ix;
    uval[ul][k] = vv;
  }
  __syncthreads();
  const int64_t j = j0 + w, u = u0 + lane;
  int32_t res = 0;
  if (j < nq && u < nu) {
    const uint8_t *rj = rt + j * nus;
    const int32_t own = (int32_t)rj[u];
    res = own;
    const int64_t n64 = q_off[j + 1] - q_off[j];  // (wave-uniform)
    if (own == 0 && n64 >= 0 && n64 <= nlimit) {
      const double qp = weighted_average(
          (int)n64, sequential != 0, [&](int k) { return (int32_t)rt[(int64_t)lidx[w][k] * nus + u]; },
          [&](int k) { return lval[w][k]; });
      int m = 0;
      while (m < ku && uidx[lane][m] >= 0) ++m;
      const double up = weighted_average(
          m, sequential != 0, [&](int k) { return (int32_t)rj[uidx[lane][k]]; }, [&](int k) { return uval[lane][k]; });
      double r;
      if (up == 0.0 && qp == 0.0) r = 0.0;
      else if (up == 0.0) r = qp * (qw + (uw * 0.5)) + dmean * (uw * 0.5);
      else if (qp == 0.0) r = up * (uw + (qw * 0.5)) + dmean * (qw * 0.5);
      else r = qp * qw + up * uw;
      res = (int32_t)rint(r);
    }
  }
  otile[lane][w] = res;
  __syncthreads();
  {
    // 16 consecutive threads write one user's 16 cells (64 bytes)
    const int ul = t >> 4, jj = t & 15;
    const int64_t uu = u0 + ul, jw = j0 + jj;
    if (uu < nu && jw < nq) out[uu * nq + jw] = otile[ul][jj];
  }
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
                                                                      double *__restrict__ val_t,
                                                                      const uint32_t *__restrict__ only_if = nullptr) {
  if (only_if && *only_if == 0) return;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)kq * nq) return;
  const int64_t k = t / nq, j = t - k * nq;
  const int64_t lo = q_off[j];
  const bool in = k < q_off[j + 1] - lo;
  idx_t[t] = in ? q_idx[lo + k] : -1;
  val_t[t] = in ? q_val[lo + k] : 0.0;
}

static int64_t pt_stride(int64_t nu) { return (nu + 63) / 64 * 64; }

// workspace: [transposed query lists: kq x nq doubles, kq x nq int32 (row / cell forms)][16 B: flags]
//            [byte matrix rt: nq x round_up(nu, 64) (tile form)]
QRLSH_EXPORT size_t qrlsh_predict_workspace_bytes(int64_t nu, int64_t nq, int32_t kq) {
  if (nu <= 0 || nq <= 0 || kq <= 0) return 0;
  return (size_t)kq * (size_t)nq * (sizeof(double) + sizeof(int32_t)) + 16 + (size_t)nq * (size_t)pt_stride(nu);
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
  QR_CHECK_ARG(kq >= 0 && kq <= PRED_MAXK && (!tl || workspace_bytes >= qrlsh_predict_workspace_bytes(nu, nq, kq)),
               "qrlsh_predict: kq=%d (<= %d) needs %zu workspace bytes, got %zu", kq, PRED_MAXK,
               qrlsh_predict_workspace_bytes(nu, nq, kq), workspace_bytes);
  QR_LAUNCH("predict_check", predict_check_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, st, q_off, nq,
            tl ? (int)kq : PRED_MAXK, too_long_out);
  const int seq = (int)(sum_order == QRLSH_SUM_SEQUENTIAL);
  if (!tl) {
    QR_LAUNCH("predict", predict_kernel<false>, dim3((unsigned)ceil_div64(nu * nq, 256)), dim3(256), 0, st, ratings, nu,
              nq, q_off, q_idx, q_val, u_idx, u_val, ku, query_weight, user_weight, default_mean, seq, PRED_MAXK, out);
    QR_LAUNCH_CHECK("qrlsh_predict");
    return QRLSH_OK;
  }
  double *val_t = static_cast<double *>(workspace);
  int32_t *idx_t = reinterpret_cast<int32_t *>(val_t + (size_t)kq * nq);
  uint32_t *wide = reinterpret_cast<uint32_t *>(idx_t + (size_t)kq * nq);   // 4-byte aligned; 16 bytes reserved
  uint8_t *rt = reinterpret_cast<uint8_t *>(wide) + 16;
  const bool tile = ku <= PT_MAXKU && ceil_div64(nq, PT_QUERIES) <= 2147483647ll && ceil_div64(nu, PT_USERS) <= 65535 &&
                    ceil_div64(nu, 64) <= 65535;
  const bool row = nq <= PR_MAXQ && nu <= 65535;
  if (tile) {
    // tile form; the row (or cell) form follows and does nothing unless a rating did not fit a byte
    const int64_t nus = pt_stride(nu);
    if (hipMemsetAsync(wide, 0, 16, st) != hipSuccess) {
      qrlsh_set_error("qrlsh_predict: hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    QR_LAUNCH("predict_transpose", predict_transpose_u8_kernel,
              dim3((unsigned)ceil_div64(nq, 64), (unsigned)ceil_div64(nu, 64)), dim3(256), 0, st, ratings, nu, nq, nus, rt,
              wide);
    QR_LAUNCH("predict", predict_tile_kernel, dim3((unsigned)ceil_div64(nq, PT_QUERIES), (unsigned)ceil_div64(nu, PT_USERS)),
              dim3(PT_THREADS), 0, st, (const uint8_t *)rt, nus, ratings, nu, nq, q_off, q_idx, q_val, u_idx, u_val, ku,
              query_weight, user_weight, default_mean, seq, (int)kq, (const uint32_t *)wide, out);
  }
  // lists transposed once ([kq][nq], doubles first: 8-byte aligned), then read coalesced by every user's sweep
  const uint32_t *only_if = tile ? (const uint32_t *)wide : nullptr;
  QR_LAUNCH("predict_lists", predict_lists_transpose_kernel, dim3((unsigned)ceil_div64((int64_t)kq * nq, 256)), dim3(256), 0,
            st, q_off, q_idx, q_val, nq, (int)kq, idx_t, val_t, only_if);
  if (row) {
    // row form: slices so that a few thousand workgroups exist whatever the number of users
    int64_t slices = ceil_div64(4096, nu);
    const int64_t most = ceil_div64(nq, PR_THREADS);
    if (slices > most) slices = most;
    if (slices < 1) slices = 1;
    QR_LAUNCH("predict_rows", predict_row_kernel, dim3((unsigned)slices, (unsigned)nu), dim3(PR_THREADS), 0, st, ratings, nu,
              nq, q_off, (const int32_t *)idx_t, (const double *)val_t, u_idx, u_val, ku, query_weight, user_weight,
              default_mean, seq, (int)kq, out, only_if);
  } else {
    QR_LAUNCH("predict_cells", predict_kernel<true>, dim3((unsigned)ceil_div64(nu * nq, 256)), dim3(256), 0, st, ratings, nu,
              nq, q_off, (const int32_t *)idx_t, (const double *)val_t, u_idx, u_val, ku, query_weight, user_weight,
              default_mean, seq, (int)kq, out, only_if);
  }
  QR_LAUNCH_CHECK("qrlsh_predict");
  return QRLSH_OK;
}
