// answers.hip -- N2: answer sets of conjunctive attribute=value queries (the producer of the hot
// path's input).
//
// Reference: Recommender.compute_shingles, recommender.py:68-103 -- per query a pandas boolean
// mask per constrained feature (dataset[feature] == value), AND-ed (:87-89), and the indices
// of the surviving rows (:91).  Here every (feature, value) of the table has one bitmap over
// the D table rows (the query x item incidence structure, one bit per cell); a query is the AND
// of at most nfeat bitmap rows.  One wave per query sweeps the D/32 words 64 at a time
// (coalesced 256-B reads per bitmap row, L2-resident), popcounts for the size pass, and in the
// fill pass turns set bits into ascending row ids with a wave prefix sum -- the CSR
// (offsets, rows) that qrlsh_minhash consumes, built without leaving the device.
#include "common.h"

// qrows[q][f] : bitmap row of query q's value for feature f, or -1 when the feature is
//               unconstrained ("" in the reference, :86); a value absent from the table must
//               point at an all-zero bitmap row.
// MODE 0: sizes only.  MODE 1: write the rows at offsets[q] (needs the scanned sizes).
// MODE 2: ONE sweep that records the size AND parks the first AS_SLOT row ids of every query in a
//         padded slot (tmp[q][AS_SLOT]); answer_sets_compact_kernel then moves the slots to their
//         CSR positions, and only queries with more than AS_SLOT rows (rare) are swept again.
constexpr int AS_SLOT = 64;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// VEC = words per lane per step: 4 (16-byte loads; needs words_per_row % 4 == 0 and 16-byte aligned
// bitmap rows) or 1.
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void answer_sets_kernel(const uint32_t *__restrict__ bitmaps, int64_t wpr,
                                                          int64_t D, const int32_t *__restrict__ qrows, int64_t nq,
                                                          int nfeat, int32_t *__restrict__ sizes,
                                                          const int64_t *__restrict__ offsets,
                                                          int32_t *__restrict__ rows) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= nq) return;  // wave-uniform
  // the (at most 64) bitmap rows of this query, one per lane
  const int32_t myrow = lane < nfeat ? qrows[q * nfeat + lane] : -1;
  constexpr bool FILL = MODE == 1;
  int64_t out = FILL ? offsets[q] : 0;
  uint32_t total = 0;
  for (int64_t w0 = 0; w0 < wpr; w0 += WAVE * VEC) {
    const int64_t w = w0 + (int64_t)lane * VEC;  // first of this lane's VEC consecutive words
    uint32_t word[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int64_t we = w + e;
      // rows >= D never match (an unconstrained query keeps exactly the D table rows)
      word[e] = 0u;
      if (we < wpr && we * 32 < D) word[e] = (D - we * 32 >= 32) ? 0xFFFFFFFFu : ((1u << (D - we * 32)) - 1u);
    }
    for (int f = 0; f < nfeat; ++f) {
      const int32_t r = __shfl(myrow, f, WAVE);
      if (r >= 0 && w < wpr) {
        const uint32_t *src = bitmaps + (size_t)r * wpr + w;
        if (VEC == 4) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(src);
          word[0] &= v.x; word[1 % VEC] &= v.y; word[2 % VEC] &= v.z; word[3 % VEC] &= v.w;
        } else {
          word[0] &= src[0];
        }
      }
    }
    uint32_t pc = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) pc += (uint32_t)__popc(word[e]);
    if (MODE == 0) {
      total += pc;
    } else {
      // exclusive prefix of the lanes' popcounts: where this lane's rows start
      uint32_t inc = pc;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += o;
      }
      const uint32_t step_total = __shfl(inc, WAVE - 1, WAVE);
      if (MODE == 2) {
        uint32_t p = total + (inc - pc);  // total is wave-uniform here
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          uint32_t x = word[e];
          while (x && p < (uint32_t)AS_SLOT) {
            const int bit = __ffs(x) - 1;
            rows[q * AS_SLOT + p++] = (int32_t)((w + e) * 32 + bit);
            x &= x - 1;
          }
        }
        total += step_total;
      } else {
        int64_t p = out + (inc - pc);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          uint32_t x = word[e];
          while (x) {
            const int bit = __ffs(x) - 1;
            rows[p++] = (int32_t)((w + e) * 32 + bit);
            x &= x - 1;
          }
        }
        out += step_total;
      }
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) total += __shfl_xor(total, m, WAVE);
    if (lane == 0) sizes[q] = (int32_t)total;
  } else if (MODE == 2) {
    if (lane == 0) sizes[q] = (int32_t)total;
  }
}

// slots -> CSR: one wave per query copies its (at most AS_SLOT) parked row ids; larger answer sets
// are flagged (big[q] = 1 via sizes) for the caller's second sweep.
__global__ __launch_bounds__(256) void answer_sets_compact_kernel(const int32_t *__restrict__ tmp,
                                                                  const int64_t *__restrict__ offsets, int64_t nq,
                                                                  int32_t *__restrict__ rows) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= nq) return;
  const int64_t lo = offsets[q];
  const int64_t n = offsets[q + 1] - lo;
  if (n <= AS_SLOT && lane < n) rows[lo + lane] = tmp[q * AS_SLOT + lane];
}

static bool answers_vec4(const uint32_t *bitmaps, int64_t wpr) {
  return (wpr % 4) == 0 && (((uintptr_t)bitmaps) & 15) == 0;
}

static int answers_check(const uint32_t *bitmaps, int64_t wpr, int64_t D, const int32_t *qrows, int64_t nq,
                         int32_t nfeat, const char *name) {
  QR_CHECK_ARG(nq >= 0 && wpr > 0 && nfeat > 0 && nfeat <= WAVE && D > 0 && D <= wpr * 32,
               "%s: bad sizes nq=%lld words_per_row=%lld D=%lld nfeat=%d", name, (long long)nq, (long long)wpr,
               (long long)D, nfeat);
  QR_CHECK_ARG(nq == 0 || (bitmaps && qrows), "%s: null pointer", name);
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_answer_sets_count(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                                         const int32_t *qrows, int64_t nq, int32_t nfeat, int32_t *sizes_out,
                                         void *stream) {
  const int rc = answers_check(bitmaps, words_per_row, D, qrows, nq, nfeat, "qrlsh_answer_sets_count");
  if (rc != QRLSH_OK) return rc;
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(sizes_out, "qrlsh_answer_sets_count: null output");
  if (answers_vec4(bitmaps, words_per_row))
    QR_LAUNCH("answers_count", (answer_sets_kernel<0, 4>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, sizes_out,
              (const int64_t *)nullptr, (int32_t *)nullptr);
  else
    QR_LAUNCH("answers_count", (answer_sets_kernel<0, 1>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, sizes_out,
              (const int64_t *)nullptr, (int32_t *)nullptr);
  QR_LAUNCH_CHECK("qrlsh_answer_sets_count");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_answer_sets_fill(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                                        const int32_t *qrows, int64_t nq, int32_t nfeat, const int64_t *offsets,
                                        int32_t *rows_out, void *stream) {
  const int rc = answers_check(bitmaps, words_per_row, D, qrows, nq, nfeat, "qrlsh_answer_sets_fill");
  if (rc != QRLSH_OK) return rc;
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(offsets, "qrlsh_answer_sets_fill: null offsets");
  if (answers_vec4(bitmaps, words_per_row))
    QR_LAUNCH("answers_fill", (answer_sets_kernel<1, 4>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, (int32_t *)nullptr, offsets,
              rows_out);
  else
    QR_LAUNCH("answers_fill", (answer_sets_kernel<1, 1>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, (int32_t *)nullptr, offsets,
              rows_out);
  QR_LAUNCH_CHECK("qrlsh_answer_sets_fill");
  return QRLSH_OK;
}

// One-sweep form: sizes_out[q] = |A(q)| and the first 64 row ids of every query parked in
// slots_out[nq][64]; follow with an exclusive scan of the sizes and qrlsh_answer_sets_compact.
QRLSH_EXPORT int qrlsh_answer_sets_sweep(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                                         const int32_t *qrows, int64_t nq, int32_t nfeat, int32_t *sizes_out,
                                         int32_t *slots_out, void *stream) {
  const int rc = answers_check(bitmaps, words_per_row, D, qrows, nq, nfeat, "qrlsh_answer_sets_sweep");
  if (rc != QRLSH_OK) return rc;
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(sizes_out && slots_out, "qrlsh_answer_sets_sweep: null output");
  if (answers_vec4(bitmaps, words_per_row))
    QR_LAUNCH("answers_sweep", (answer_sets_kernel<2, 4>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, sizes_out,
              (const int64_t *)nullptr, slots_out);
  else
    QR_LAUNCH("answers_sweep", (answer_sets_kernel<2, 1>), dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
              static_cast<hipStream_t>(stream), bitmaps, words_per_row, D, qrows, nq, nfeat, sizes_out,
              (const int64_t *)nullptr, slots_out);
  QR_LAUNCH_CHECK("qrlsh_answer_sets_sweep");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_answer_sets_compact(const int32_t *slots, const int64_t *offsets, int64_t nq,
                                           int32_t *rows_out, void *stream) {
  QR_CHECK_ARG(nq >= 0, "qrlsh_answer_sets_compact: bad nq");
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(slots && offsets && rows_out, "qrlsh_answer_sets_compact: null pointer");
  QR_LAUNCH("answers_compact", answer_sets_compact_kernel, dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0,
            static_cast<hipStream_t>(stream), slots, offsets, nq, rows_out);
  QR_LAUNCH_CHECK("qrlsh_answer_sets_compact");
  return QRLSH_OK;
}
