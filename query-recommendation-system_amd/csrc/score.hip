// score.hip -- a5: cosine of the MinHash-value vectors of every candidate pair.
//
// Reference: recommender.py:203-204,
//     values = np.around(cosine_similarity([sig_i, sig_j1, ...])[0][1:], 3)
// (sklearn builds the whole (k+1)^2 Gram matrix in float64 and keeps row 0).  Here each
// unique pair is scored once: exact integer dot product of the two signature rows read
// from HBM (16 B per lane, 16 lanes per pair, 4 pairs per wave-instruction), the squared
// norms come precomputed from the MinHash kernel, and
//     cos = dot / (sqrt(na) * sqrt(nb))   in float64,  milli = rint(cos * 1000)
// so that milli / 1000.0 == np.around(cos, 3).  Integer dot/norms are exact (|sig| < 2^31,
// P * D^2 < 2^63), so the only rounding is the final divide -- within 2 ulp of sklearn's
// normalise-then-multiply and identical after the rounding to 3 decimals (checked on every
// golden pair).  HBM-bound: 2 * 4P bytes of rows per pair.
#include <stdlib.h>

#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SCORE_LPP = 16;  // lanes per pair

// value of a compact uint16 signature entry: 0xFFFF is the -1 of an empty answer set
__device__ static inline int64_t c16(unsigned short v) { return v == 0xFFFFu ? -1 : (int64_t)v; }

// SigT = int32_t (reference layout) or uint16_t (compact rows, D <= 65535: half the HBM bytes)
template <typename SigT, bool VECLOAD>
__global__ __launch_bounds__(256) void score_pairs_kernel(const SigT *__restrict__ sig,
                                                          const int64_t *__restrict__ norm2, int P,
                                                          const uint64_t *__restrict__ pairs, int64_t n,
                                                          int32_t *__restrict__ milli, double *__restrict__ cosv,
                                                          uint64_t *__restrict__ edges, int id_bits,
                                                          uint32_t *__restrict__ edge_dst,
                                                          const SigT *__restrict__ sig_b,
                                                          const int64_t *__restrict__ norm2_b, uint32_t split,
                                                          int rev_only) {
  constexpr bool IS16 = sizeof(SigT) == 2;
  constexpr int VEC = IS16 ? 8 : 4;
  const int lane = threadIdx.x & (WAVE - 1);
  const int lig = lane & (SCORE_LPP - 1);
  const int64_t group = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SCORE_LPP);
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / SCORE_LPP;
  const int64_t iters = (n + ngroups - 1) / ngroups;  // uniform trip count for the shuffles
  // software prefetch of the next pair word
  uint64_t pr_next = (group < n) ? pairs[group] : 0;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t t = it * ngroups + group;
    const bool live = t < n;
    const uint64_t pr = pr_next;
    const int64_t tn = t + ngroups;
    pr_next = (tn < n) ? pairs[tn] : 0;
    const uint32_t i = (uint32_t)(pr >> 32), j = (uint32_t)pr;
    // row table in two pieces (sharded driver): rows below `split` are the rank's own, the rest were fetched
    const SigT *a = i < split ? sig + (size_t)i * P : sig_b + (size_t)(i - split) * P;
    const SigT *c = j < split ? sig + (size_t)j * P : sig_b + (size_t)(j - split) * P;
    // norm2 == NULL: the two squared norms are summed from the rows themselves, which are in registers anyway -- a
    // precomputed norm is one more random 64-B sector per row fetched (a quarter of a compact 256-B row on top)
    const bool own_norms = norm2 == nullptr;
    int64_t na = 0, nb = 0;
    if (!own_norms && live && lig == 0) {
      na = i < split ? norm2[i] : norm2_b[i - split];
      nb = j < split ? norm2[j] : norm2_b[j - split];
    }
    int64_t dot = 0;
    if (live) {
      if (VECLOAD) {
        for (int col = lig * VEC; col < P; col += SCORE_LPP * VEC) {
          if (IS16) {
            const u16x8 x = *reinterpret_cast<const u16x8 *>(a + col);
            const u16x8 y = *reinterpret_cast<const u16x8 *>(c + col);
            // no value of either chunk has bit 15 set (always so for D <= 32768 and non-empty answer sets; 0xFFFF,
            // the -1 of an empty set, has it): four products then fit 32 bits and v_dot2_u32_u16 sums two element
            // pairs per instruction on the packed words as loaded -- ~20 instructions per lane instead of ~45
            // (unpack, -1 test, 64-bit multiply-add per element).  Decided per chunk from the data, same integers.
            const u32x4 xw = __builtin_bit_cast(u32x4, x), yw = __builtin_bit_cast(u32x4, y);
            const uint32_t hi = (xw[0] | xw[1] | xw[2] | xw[3] | yw[0] | yw[1] | yw[2] | yw[3]) & 0x80008000u;
            if (hi == 0) {
              uint32_t s0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 0, 1),
                                                   __builtin_shufflevector(y, y, 0, 1), 0u, false);
              uint32_t s1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 2, 3),
                                                   __builtin_shufflevector(y, y, 2, 3), 0u, false);
              s0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(y, y, 4, 5),
                                          s0, false);
              s1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(y, y, 6, 7),
                                          s1, false);
              dot += (int64_t)((uint64_t)s0 + (uint64_t)s1);
              if (own_norms) {
                uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                a0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(x, x, 0, 1), a0, false);
                a1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(x, x, 2, 3), a1, false);
                a0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(x, x, 4, 5), a0, false);
                a1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(x, x, 6, 7), a1, false);
                b0 = __builtin_amdgcn_udot2(__builtin_shufflevector(y, y, 0, 1), __builtin_shufflevector(y, y, 0, 1), b0, false);
                b1 = __builtin_amdgcn_udot2(__builtin_shufflevector(y, y, 2, 3), __builtin_shufflevector(y, y, 2, 3), b1, false);
                b0 = __builtin_amdgcn_udot2(__builtin_shufflevector(y, y, 4, 5), __builtin_shufflevector(y, y, 4, 5), b0, false);
                b1 = __builtin_amdgcn_udot2(__builtin_shufflevector(y, y, 6, 7), __builtin_shufflevector(y, y, 6, 7), b1, false);
                na += (int64_t)((uint64_t)a0 + (uint64_t)a1);
                nb += (int64_t)((uint64_t)b0 + (uint64_t)b1);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int64_t xe = c16(x[e]), ye = c16(y[e]);
                dot += xe * ye;
                if (own_norms) {
                  na += xe * xe;
                  nb += ye * ye;
                }
              }
            }
          } else {
            const i32x4 x = *reinterpret_cast<const i32x4 *>(a + col);
            const i32x4 y = *reinterpret_cast<const i32x4 *>(c + col);
            dot += (int64_t)x.x * y.x + (int64_t)x.y * y.y + (int64_t)x.z * y.z + (int64_t)x.w * y.w;
            if (own_norms) {
              na += (int64_t)x.x * x.x + (int64_t)x.y * x.y + (int64_t)x.z * x.z + (int64_t)x.w * x.w;
              nb += (int64_t)y.x * y.x + (int64_t)y.y * y.y + (int64_t)y.z * y.z + (int64_t)y.w * y.w;
            }
          }
        }
      } else {
        for (int col = lig; col < P; col += SCORE_LPP) {
          const int64_t xe = IS16 ? c16((unsigned short)a[col]) : (int64_t)a[col];
          const int64_t ye = IS16 ? c16((unsigned short)c[col]) : (int64_t)c[col];
          dot += xe * ye;
          if (own_norms) {
            na += xe * xe;
            nb += ye * ye;
          }
        }
      }
    }
#pragma unroll
    for (int m = 1; m < SCORE_LPP; m <<= 1) dot += __shfl_xor(dot, m, WAVE);
    if (own_norms) {  // uniform
#pragma unroll
      for (int m = 1; m < SCORE_LPP; m <<= 1) {
        na += __shfl_xor(na, m, WAVE);
        nb += __shfl_xor(nb, m, WAVE);
      }
    }
    if (live && lig == 0) {
      double cs = 0.0;
      if (na != 0 && nb != 0) cs = (double)dot / (sqrt((double)na) * sqrt((double)nb));
      const int32_t mi = (int32_t)rint(cs * 1000.0);
      milli[t] = mi;
      if (cosv) cosv[t] = cs;
      if (edges) {
        const uint64_t inv = (uint64_t)(1000 - mi);
        if (rev_only) {  // only the reverse edge (src = j), one word per pair: the select form of the top-K
          if (edge_dst) {
            edges[t] = ((uint64_t)j << 11) | inv;
            edge_dst[t] = i;
          } else {
            edges[t] = ((uint64_t)j << (id_bits + 11)) | (inv << id_bits) | i;
          }
        } else if (edge_dst) {  // wide ids: src << 11 | inv in the key, dst as the payload
          edges[2 * t] = ((uint64_t)i << 11) | inv;
          edges[2 * t + 1] = ((uint64_t)j << 11) | inv;
          edge_dst[2 * t] = j;
          edge_dst[2 * t + 1] = i;
        } else {
          edges[2 * t] = ((uint64_t)i << (id_bits + 11)) | (inv << id_bits) | j;
          edges[2 * t + 1] = ((uint64_t)j << (id_bits + 11)) | (inv << id_bits) | i;
        }
      }
    }
  }
}

// ---- the run form (round 4): compact uint16 rows of 128 or 256 values, precomputed norms ------------------------------
// The candidate pairs are SORTED (i, j): consecutive pairs share their first row.  A 16-lane group (one DPP row) takes
// 16 CONSECUTIVE pairs at a time instead of every ngroups-th one:
//   * one coalesced 128-B load brings the group its 16 pair words, lane u holding pair u; both norms of pair u are
//     loaded by lane u itself -- 3 vector-memory instructions per 64 pairs of a wave where the form above issues 3 per
//     FOUR pairs;
//   * row i stays in registers while i does not change (a new a-row is loaded for ~1 pair in 4 on the bench data);
//   * the second row of pair u + PF is requested before pair u is summed (PF rows in flight per lane behind the
//     arithmetic);
//   * the 16-lane sum of a pair is a DPP all-reduce (two quad permutes, half-row mirror, row mirror: no LDS pipe) and
//     lane u keeps pair u's total, so the float64 square roots, the divide and the rounding run ONCE per chunk on all
//     64 lanes (one pair each) instead of per pair on 4 lanes of 64, and scores / edge words leave as whole lines.
// Same integers, same float64 expression: bit-identical to score_pairs_kernel.  Pairs beyond n are computed on row 0
// and never stored.
template <int U> __device__ static inline uint32_t row_bcast_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x150 + U, 0xF, 0xF, false);   // row_newbcast:U
}
template <int CTRL> __device__ static inline uint64_t dpp_add_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
  return v + (((uint64_t)hi << 32) | lo);
}
// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ static inline uint64_t row_sum_u64(uint64_t v) {
  v = dpp_add_u64<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add_u64<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add_u64<0x141>(v);  // row_half_mirror
  v = dpp_add_u64<0x140>(v);  // row_mirror
  return v;
}

// a chunk with a value >= 0x8000 (D > 32768, or the 0xFFFF = -1 of an empty answer set): element by element in 64 bits.
// Not inlined: inlined, the compiler unpacks the first row's elements where that row is LOADED (it changes only every few
// pairs), which costs eight registers and -- worse -- a wait for ALL outstanding loads right behind the predicated load.
__device__ static __attribute__((noinline)) uint64_t chunk_dot_wide(u16x8 x, u16x8 y) {
  int64_t d = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) d += c16(x[e]) * c16(y[e]);
  return (uint64_t)d;
}

// exact dot product of two 8-value chunks of compact rows (see score_pairs_kernel)
__device__ static inline uint64_t chunk_dot(u16x8 x, u16x8 y) {
  const u32x4 xw = __builtin_bit_cast(u32x4, x), yw = __builtin_bit_cast(u32x4, y);
  const uint32_t hi = (xw[0] | xw[1] | xw[2] | xw[3] | yw[0] | yw[1] | yw[2] | yw[3]) & 0x80008000u;
  if (hi == 0) {
    uint32_t s0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(y, y, 0, 1), 0u, false);
    uint32_t s1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(y, y, 2, 3), 0u, false);
    s0 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(y, y, 4, 5), s0, false);
    s1 = __builtin_amdgcn_udot2(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(y, y, 6, 7), s1, false);
    return (uint64_t)s0 + (uint64_t)s1;
  }
  return chunk_dot_wide(x, y);
}

#ifndef QR_SCORE_PF
#define QR_SCORE_PF 1   // pairs whose rows are requested ahead of the one being summed (1: 2.25 ms at 10 M, 2 - 4: 2.45)
#endif
constexpr int SCORE_PF = QR_SCORE_PF;

#ifndef QR_SCORE_OCC
#define QR_SCORE_OCC 8   // workgroups of 4 waves per CU the register budget is held to (8: 64 VGPRs)
#endif
#ifndef QR_SCORE_OCC2
#define QR_SCORE_OCC2 5  // the same for rows of 256 values (twice the row registers: 5 -> 96 VGPRs)
#endif
template <int CH>
__global__ __launch_bounds__(256, CH == 1 ? QR_SCORE_OCC : QR_SCORE_OCC2) void score_runs_kernel(const uint16_t *__restrict__ sig, const int64_t *__restrict__ norm2,
                                                         const uint64_t *__restrict__ pairs, int64_t n,
                                                         int32_t *__restrict__ milli, double *__restrict__ cosv,
                                                         uint64_t *__restrict__ edges, int id_bits,
                                                         uint32_t *__restrict__ edge_dst,
                                                         const uint16_t *__restrict__ sig_b,
                                                         const int64_t *__restrict__ norm2_b, uint32_t split, int rev_only) {
  constexpr int P = CH * 128;
  constexpr int NB = SCORE_PF + 1;   // row buffers in rotation
  const int lig = threadIdx.x & (SCORE_LPP - 1);
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SCORE_LPP;
  const int64_t nchunks = (n + SCORE_LPP - 1) / SCORE_LPP;
  auto row_of = [&](uint32_t id) -> const uint16_t * {
    return (id < split ? sig + (size_t)id * P : sig_b + (size_t)(id - split) * P) + lig * 8;
  };
  auto load_row = [&](u16x8 (&dst)[CH], const uint16_t *p) {
#pragma unroll
    for (int k = 0; k < CH; ++k) dst[k] = *reinterpret_cast<const u16x8 *>(p + k * 128);
  };
  // ONE chunk per group and launch: no chunk loop for the compiler to hoist 16 lane masks and half a dozen 64-bit
  // induction variables out of (they did not fit 64 registers next to the row buffers and went to scratch: 5.8 ms)
  {
    const int64_t c = group;
    if (c >= nchunks) return;   // (whole DPP rows leave together)
    const int64_t t = c * SCORE_LPP + lig;
    const bool live = t < n;
    const uint64_t pw = live ? pairs[t] : 0ull;
    const uint32_t my_i = (uint32_t)(pw >> 32), my_j = (uint32_t)pw;
    int64_t na = 0, nb = 0;
    if (live) {
      na = my_i < split ? norm2[my_i] : norm2_b[my_i - split];
      nb = my_j < split ? norm2[my_j] : norm2_b[my_j - split];
    }
    uint64_t my_dot = 0;
    // rows in rotation: pair V's two rows are requested together, PF pairs ahead of the one being summed (the counter
    // that orders the waits is in issue order: a first row requested LATER than the second rows already in flight
    // would make its consumer wait for all of them).  The first row of pair V is a copy of pair V - 1's unless i changed.
    u16x8 ab[NB][CH], cb[NB][CH];
#define QR_ROW_I(U) row_bcast_u32<(U)>(my_i)
#define QR_ROW_J(U) row_bcast_u32<(U)>(my_j)
#ifndef QR_SCORE_AREUSE
#define QR_SCORE_AREUSE 0
#endif
#if QR_SCORE_AREUSE   /* first row copied from the previous pair's unless i changed (the copy waits for that row) */
#define QR_REQ(V)                                                                           \
    if ((V) < SCORE_LPP) {                                                                  \
      if ((V) == 0) {                                                                       \
        load_row(ab[0], row_of(QR_ROW_I(0)));                                               \
      } else {                                                                              \
        const uint32_t vi = QR_ROW_I((V) % SCORE_LPP), pi = QR_ROW_I(((V) + SCORE_LPP - 1) % SCORE_LPP); \
        _Pragma("unroll") for (int k = 0; k < CH; ++k) ab[(V) % NB][k] = ab[((V) + NB - 1) % NB][k]; \
        if (vi != pi) load_row(ab[(V) % NB], row_of(vi));                                   \
      }                                                                                     \
      load_row(cb[(V) % NB], row_of(QR_ROW_J((V) % SCORE_LPP)));                            \
    }
#else                 /* both rows of every pair requested (a repeated first row is a cache hit) */
#define QR_REQ(V)                                                                           \
    if ((V) < SCORE_LPP) {                                                                  \
      load_row(ab[(V) % NB], row_of(QR_ROW_I((V) % SCORE_LPP)));                            \
      load_row(cb[(V) % NB], row_of(QR_ROW_J((V) % SCORE_LPP)));                            \
    }
#endif
    static_assert(SCORE_PF >= 1 && SCORE_PF <= 4, "prefetch depth");
    QR_REQ(0)
    if (SCORE_PF > 1) { QR_REQ(1) }
    if (SCORE_PF > 2) { QR_REQ(2) }
    if (SCORE_PF > 3) { QR_REQ(3) }
#define QR_STEP(U)                                                                          \
    {                                                                                       \
      QR_REQ((U) + SCORE_PF)                                                                \
      uint64_t part = 0;                                                                    \
      _Pragma("unroll") for (int k = 0; k < CH; ++k) part += chunk_dot(ab[(U) % NB][k], cb[(U) % NB][k]); \
      const uint64_t tot = row_sum_u64(part);                                               \
      if (lig == (U)) my_dot = tot;                                                         \
      asm volatile("" : "+v"(my_dot)); /* materialised now: the compiler would otherwise keep all 16 totals and \
                                          pick one at the end (32 registers, spilled) */    \
      __builtin_amdgcn_sched_barrier(0); /* the loads stay where they are written: PF pairs ahead, not all 16 */ \
    }
    QR_STEP(0) QR_STEP(1) QR_STEP(2) QR_STEP(3) QR_STEP(4) QR_STEP(5) QR_STEP(6) QR_STEP(7)
    QR_STEP(8) QR_STEP(9) QR_STEP(10) QR_STEP(11) QR_STEP(12) QR_STEP(13) QR_STEP(14) QR_STEP(15)
#undef QR_STEP
#undef QR_REQ
#undef QR_ROW_I
#undef QR_ROW_J
    if (live) {
      const int64_t dot = (int64_t)my_dot;
      double cs = 0.0;
      if (na != 0 && nb != 0) cs = (double)dot / (sqrt((double)na) * sqrt((double)nb));
      const int32_t mi = (int32_t)rint(cs * 1000.0);
      milli[t] = mi;
      if (cosv) cosv[t] = cs;
      if (edges) {
        const uint64_t inv = (uint64_t)(1000 - mi), i = my_i, j = my_j;
        if (rev_only) {
          if (edge_dst) {
            edges[t] = (j << 11) | inv;
            edge_dst[t] = (uint32_t)i;
          } else {
            edges[t] = (j << (id_bits + 11)) | (inv << id_bits) | i;
          }
        } else if (edge_dst) {
          edges[2 * t] = (i << 11) | inv;
          edges[2 * t + 1] = (j << 11) | inv;
          edge_dst[2 * t] = (uint32_t)j;
          edge_dst[2 * t + 1] = (uint32_t)i;
        } else {
          edges[2 * t] = (i << (id_bits + 11)) | (inv << id_bits) | j;
          edges[2 * t + 1] = (j << (id_bits + 11)) | (inv << id_bits) | i;
        }
      }
    }
  }
}

static int g_score_runs = -1;   // -1: read QRLSH_SCORE_RUNS on first use (default on)
// 1: compact 128 / 256-value rows are scored by the run form (default), 0: by the generic form (same results; an A/B
// and test knob).  Returns the previous setting.
QRLSH_EXPORT int qrlsh_set_score_runs(int on) {
  const int old = g_score_runs;
  g_score_runs = on != 0;
  return old;
}

static int score_launch(const void *sig, const void *sig_b, int64_t split, int32_t sig_dtype, const int64_t *norm2,
                        const int64_t *norm2_b, int32_t P, const uint64_t *pairs, int64_t n, int32_t *milli_out,
                        double *cos_out, uint64_t *edge_out, int32_t id_bits, uint32_t *edge_dst_out, void *stream,
                        int rev_only = 0) {
  const int64_t groups_per_block = 256 / SCORE_LPP;
  int64_t blocks = ceil_div64(n, groups_per_block);
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), block(256);
  const bool aligned = ((((uintptr_t)sig | (uintptr_t)sig_b) & 15) == 0);
  const uint32_t sp = split >= (1ll << 32) ? 0xFFFFFFFFu : (uint32_t)split;
  if (sig_dtype == QRLSH_SIG_U16) {
    const uint16_t *s16 = static_cast<const uint16_t *>(sig), *s16b = static_cast<const uint16_t *>(sig_b);
    // the run form: compact rows of 128 / 256 values, precomputed norms (QRLSH_SCORE_RUNS=0: the generic form)
    if (g_score_runs < 0) {
      const char *e = getenv("QRLSH_SCORE_RUNS");
      g_score_runs = !(e && e[0] == '0');
    }
    if (g_score_runs && aligned && norm2 && (P == 128 || P == 256) && (sig_b == nullptr || norm2_b)) {
      const int64_t rb = ceil_div64(ceil_div64(n, SCORE_LPP), groups_per_block);   // one chunk of 16 pairs per group
      QR_CHECK_ARG(rb <= 2147483647ll, "qrlsh_score_pairs: too many pairs for one launch");
      const dim3 rgrid((unsigned)rb);
      if (P == 128)
        QR_LAUNCH("score_pairs", (score_runs_kernel<1>), rgrid, block, 0, st, s16, norm2, pairs, n, milli_out, cos_out,
                  edge_out, id_bits, edge_dst_out, s16b, norm2_b, sp, rev_only);
      else
        QR_LAUNCH("score_pairs", (score_runs_kernel<2>), rgrid, block, 0, st, s16, norm2, pairs, n, milli_out, cos_out,
                  edge_out, id_bits, edge_dst_out, s16b, norm2_b, sp, rev_only);
      QR_LAUNCH_CHECK("qrlsh_score_pairs");
      return QRLSH_OK;
    }
    if (aligned && P % 8 == 0)
      QR_LAUNCH("score_pairs", (score_pairs_kernel<uint16_t, true>), grid, block, 0, st, s16, norm2, P, pairs, n, milli_out,
                cos_out, edge_out, id_bits, edge_dst_out, s16b, norm2_b, sp, rev_only);
    else
      QR_LAUNCH("score_pairs", (score_pairs_kernel<uint16_t, false>), grid, block, 0, st, s16, norm2, P, pairs, n, milli_out,
                cos_out, edge_out, id_bits, edge_dst_out, s16b, norm2_b, sp, rev_only);
  } else {
    const int32_t *s32 = static_cast<const int32_t *>(sig), *s32b = static_cast<const int32_t *>(sig_b);
    if (aligned && P % 4 == 0)
      QR_LAUNCH("score_pairs", (score_pairs_kernel<int32_t, true>), grid, block, 0, st, s32, norm2, P, pairs, n, milli_out,
                cos_out, edge_out, id_bits, edge_dst_out, s32b, norm2_b, sp, rev_only);
    else
      QR_LAUNCH("score_pairs", (score_pairs_kernel<int32_t, false>), grid, block, 0, st, s32, norm2, P, pairs, n, milli_out,
                cos_out, edge_out, id_bits, edge_dst_out, s32b, norm2_b, sp, rev_only);
  }
  QR_LAUNCH_CHECK("qrlsh_score_pairs");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_score_pairs(const void *sig, int32_t sig_dtype, const int64_t *norm2, int32_t P,
                                   const uint64_t *pairs, int64_t n, int32_t *milli_out, double *cos_out,
                                   uint64_t *edge_out, int32_t id_bits, uint32_t *edge_dst_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && P > 0, "qrlsh_score_pairs: bad sizes n=%lld P=%d", (long long)n, P);
  QR_CHECK_ARG(sig_dtype == QRLSH_SIG_I32 || sig_dtype == QRLSH_SIG_U16, "qrlsh_score_pairs: bad sig_dtype %d", sig_dtype);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(sig && pairs && milli_out, "qrlsh_score_pairs: null pointer");
  if (edge_out && !edge_dst_out)
    QR_CHECK_ARG(id_bits > 0 && id_bits <= 26, "qrlsh_score_pairs: id_bits=%d must be in [1,26] without edge_dst_out", id_bits);
  QR_CHECK_ARG(!edge_dst_out || edge_out, "qrlsh_score_pairs: edge_dst_out needs edge_out");
  return score_launch(sig, nullptr, 1ll << 32, sig_dtype, norm2, nullptr, P, pairs, n, milli_out, cos_out, edge_out,
                      id_bits, edge_dst_out, stream);
}

// Scores + the REVERSE edge word of every pair only (rev_out[t] = j << (id_bits + 11) | inv << id_bits | i, or
// with rev_dst_out the key + payload form j << 11 | inv, i): the input of qrlsh_topk_select_* once sorted on j.
QRLSH_EXPORT int qrlsh_score_pairs_rev(const void *sig, int32_t sig_dtype, const int64_t *norm2, int32_t P,
                                       const uint64_t *pairs, int64_t n, int32_t *milli_out, uint64_t *rev_out,
                                       int32_t id_bits, uint32_t *rev_dst_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && P > 0, "qrlsh_score_pairs_rev: bad sizes n=%lld P=%d", (long long)n, P);
  QR_CHECK_ARG(sig_dtype == QRLSH_SIG_I32 || sig_dtype == QRLSH_SIG_U16, "qrlsh_score_pairs_rev: bad sig_dtype %d",
               sig_dtype);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(sig && pairs && milli_out && rev_out, "qrlsh_score_pairs_rev: null pointer");
  QR_CHECK_ARG(rev_dst_out || (id_bits > 0 && id_bits <= 26),
               "qrlsh_score_pairs_rev: id_bits=%d must be in [1,26] without rev_dst_out", id_bits);
  return score_launch(sig, nullptr, 1ll << 32, sig_dtype, norm2, nullptr, P, pairs, n, milli_out, nullptr, rev_out, id_bits,
                      rev_dst_out, stream, 1);
}

// The same scores against a row table that comes in two pieces: row index x < split_rows is row x of
// sig / norm2 (the rank's own queries), the others row x - split_rows of sig_b / norm2_b (rows fetched from
// other ranks) -- the sharded driver scores without first copying both into one buffer.
QRLSH_EXPORT int qrlsh_score_pairs_split(const void *sig, const int64_t *norm2, int64_t split_rows, const void *sig_b,
                                         const int64_t *norm2_b, int32_t sig_dtype, int32_t P, const uint64_t *pairs,
                                         int64_t n, int32_t *milli_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && P > 0 && split_rows >= 0 && split_rows < (1ll << 32), "qrlsh_score_pairs_split: bad sizes");
  QR_CHECK_ARG(sig_dtype == QRLSH_SIG_I32 || sig_dtype == QRLSH_SIG_U16, "qrlsh_score_pairs_split: bad sig_dtype %d",
               sig_dtype);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(pairs && milli_out && (split_rows == 0 || sig) && ((norm2 == nullptr) == (norm2_b == nullptr) || !sig_b),
               "qrlsh_score_pairs_split: null pointer (norm2 and norm2_b: both or neither)");
  return score_launch(sig, sig_b, split_rows, sig_dtype, norm2, norm2_b, P, pairs, n, milli_out, nullptr, nullptr, 0,
                      nullptr, stream, 0);
}

// Exact candidate test for wide bands (r > 4, hashed bucket ids): flags[t] = 1 iff the pair
// shares at least one band whose r low-16 values are all equal and not all -1 -- the condition
// lsh.py:31-53 implements with string keys.  One thread per pair; only used on the r > 4 path.
template <typename SigT>
__global__ __launch_bounds__(256) void verify_pairs_kernel(const SigT *__restrict__ sig, int P, int b,
                                                           const uint64_t *__restrict__ pairs, int64_t n,
                                                           uint8_t *__restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t pr = pairs[t];
  const SigT *a = sig + (size_t)(uint32_t)(pr >> 32) * P;
  const SigT *c = sig + (size_t)(uint32_t)pr * P;
  const int r = P / b;
  uint8_t hit = 0;
  for (int band = 0; band < b && !hit; ++band) {
    bool eq = true, empty = true;
    for (int k = 0; k < r; ++k) {
      const uint32_t x = (uint32_t)a[band * r + k] & 0xFFFFu, y = (uint32_t)c[band * r + k] & 0xFFFFu;
      eq &= x == y;
      empty &= x == 0xFFFFu;
    }
    hit = eq && !empty;
  }
  flags[t] = hit;
}

QRLSH_EXPORT int qrlsh_verify_pairs(const void *sig, int32_t sig_dtype, int32_t P, int32_t b, const uint64_t *pairs,
                                    int64_t n, uint8_t *flags_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && P > 0 && b > 0 && P % b == 0, "qrlsh_verify_pairs: bad sizes n=%lld P=%d b=%d", (long long)n, P, b);
  QR_CHECK_ARG(sig_dtype == QRLSH_SIG_I32 || sig_dtype == QRLSH_SIG_U16, "qrlsh_verify_pairs: bad sig_dtype %d", sig_dtype);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(sig && pairs && flags_out, "qrlsh_verify_pairs: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)ceil_div64(n, 256)), block(256);
  if (sig_dtype == QRLSH_SIG_U16)
    QR_LAUNCH("verify_pairs", (verify_pairs_kernel<uint16_t>), grid, block, 0, st, static_cast<const uint16_t *>(sig), P, b,
              pairs, n, flags_out);
  else
    QR_LAUNCH("verify_pairs", (verify_pairs_kernel<int32_t>), grid, block, 0, st, static_cast<const int32_t *>(sig), P, b,
              pairs, n, flags_out);
  QR_LAUNCH_CHECK("qrlsh_verify_pairs");
  return QRLSH_OK;
}

// ---- multi-GPU glue (qrlsh/dist.py): two small fused kernels instead of ~25 elementwise launches -------
// An owner scores pairs (i local, j anywhere) against a row table [local rows | fetched remote rows]:
// out[t] = (i - q0) << 32 | slot(j), slot(j) = j - q0 for a local j, nql + position of j in the ascending
// list `need` of fetched ids otherwise.
__global__ __launch_bounds__(256) void remap_pairs_kernel(const uint64_t *__restrict__ pairs, int64_t n, uint64_t q0,
                                                          uint64_t nql, const uint64_t *__restrict__ need,
                                                          int64_t n_need, uint64_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t pr = pairs[t], i = pr >> 32, j = pr & 0xFFFFFFFFull;
  uint64_t slot;
  if (j >= q0 && j < q0 + nql) {
    slot = j - q0;
  } else {
    int64_t lo = 0, hi = n_need;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (need[mid] < j) lo = mid + 1;
      else hi = mid;
    }
    slot = nql + (uint64_t)lo;
  }
  out[t] = (i - q0) << 32 | slot;
}

// directed edge keys of scored pairs, forward (src = i) and reverse (src = j) in separate arrays, in the
// packed format (id_bits > 0: src << (id_bits+11) | inv << id_bits | dst) or the key + payload one
// (id_bits == 0: src << 11 | inv, dst)
__global__ __launch_bounds__(256) void pair_edges_kernel(const uint64_t *__restrict__ pairs,
                                                         const int32_t *__restrict__ milli, int64_t n, int id_bits,
                                                         uint64_t *__restrict__ fwd, uint64_t *__restrict__ rev,
                                                         uint32_t *__restrict__ fwd_dst,
                                                         uint32_t *__restrict__ rev_dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t pr = pairs[t], i = pr >> 32, j = pr & 0xFFFFFFFFull, inv = (uint64_t)(1000 - milli[t]);
  // rev == NULL: both edges of pair t next to each other in fwd (2 t, 2 t + 1), as qrlsh_score_pairs writes them
  uint64_t *f = rev ? fwd + t : fwd + 2 * t, *r = rev ? rev + t : fwd + 2 * t + 1;
  if (id_bits > 0) {
    *f = i << (id_bits + 11) | inv << id_bits | j;
    *r = j << (id_bits + 11) | inv << id_bits | i;
  } else {
    *f = i << 11 | inv;
    *r = j << 11 | inv;
    if (rev) {
      fwd_dst[t] = (uint32_t)j;
      rev_dst[t] = (uint32_t)i;
    } else {
      fwd_dst[2 * t] = (uint32_t)j;
      fwd_dst[2 * t + 1] = (uint32_t)i;
    }
  }
}

QRLSH_EXPORT int qrlsh_remap_pairs(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, const uint64_t *need,
                                   int64_t n_need, uint64_t *out, void *stream) {
  QR_CHECK_ARG(n >= 0 && q0 >= 0 && nql >= 0 && n_need >= 0, "qrlsh_remap_pairs: bad sizes");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(pairs && out && (n_need == 0 || need), "qrlsh_remap_pairs: null pointer");
  QR_LAUNCH("remap_pairs", remap_pairs_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), pairs, n, (uint64_t)q0, (uint64_t)nql, need, n_need, out);
  QR_LAUNCH_CHECK("qrlsh_remap_pairs");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_pair_edges(const uint64_t *pairs, const int32_t *milli, int64_t n, int32_t id_bits,
                                  uint64_t *fwd_out, uint64_t *rev_out, uint32_t *fwd_dst_out, uint32_t *rev_dst_out,
                                  void *stream) {
  QR_CHECK_ARG(n >= 0 && id_bits >= 0 && id_bits <= 26, "qrlsh_pair_edges: bad arguments (id_bits=%d)", id_bits);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(pairs && milli && fwd_out, "qrlsh_pair_edges: null pointer");
  QR_CHECK_ARG(id_bits > 0 || (fwd_dst_out && (rev_dst_out || !rev_out)), "qrlsh_pair_edges: id_bits == 0 needs the dst outputs");
  QR_LAUNCH("pair_edges", pair_edges_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), pairs, milli, n, id_bits, fwd_out, rev_out, fwd_dst_out, rev_dst_out);
  QR_LAUNCH_CHECK("qrlsh_pair_edges");
  return QRLSH_OK;
}
