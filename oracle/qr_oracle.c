/*
 * qr_oracle.c -- CPU restatement of the reference's MinHash-LSH hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing
 * in the product package (query-recommendation-system_amd/) may import, link or
 * call it.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors
 * captured from the reference's own Python code (tools/make_golden.py ->
 * tests/golden/ (npz); tests/test_oracle_golden.py).
 *
 * Reference = wamuumu/query-recommendation-system @ 2025-01-14, files lsh.py and
 * recommender.py; each function cites the lines it restates.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared; see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QRO_API __attribute__((visibility("default")))

QRO_API int qro_version(void) { return 1; }

QRO_API int qro_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

QRO_API void qro_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

QRO_API void qro_free(void *p) { free(p); }

/* ------------------------------------------------------------------------
 * a1. MinHash signatures -- recommender.py:105-143.
 *
 * The reference walks the table rows in the order of permutation p
 * (sorted_indexes = argsort(perm), :121) and stores perm[ind] for the first row
 * that belongs to a query (:125-129), i.e.  sig[q][p] = min_{d in A(q)} perm_p[d];
 * a query with an empty answer set keeps the initial -1 (:116).  The result is
 * returned transposed, one row per query (:139).
 *
 * perm  : [P][D] int32, perm[p] = the p-th consecutive np.random.permutation(D)
 * A(q)  : rows[offsets[q] .. offsets[q+1])   (CSR form of compute_shingles' output)
 * sig   : [nq][P] int32
 * ---------------------------------------------------------------------- */
QRO_API void qro_minhash(const int64_t *offsets, const int32_t *rows, int64_t nq,
                         const int32_t *perm, int32_t P, int32_t D, int32_t *sig) {
  /* row-major copy [D][P] so the inner loop over p is contiguous */
  int32_t *pt = (int32_t *)malloc((size_t)D * (size_t)P * sizeof(int32_t));
#pragma omp parallel for schedule(static)
  for (int64_t d = 0; d < D; ++d)
    for (int32_t p = 0; p < P; ++p) pt[(size_t)d * P + p] = perm[(size_t)p * D + d];

#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t q = 0; q < nq; ++q) {
    int32_t *s = sig + (size_t)q * P;
    int64_t lo = offsets[q], hi = offsets[q + 1];
    if (hi <= lo) {
      for (int32_t p = 0; p < P; ++p) s[p] = -1;
      continue;
    }
    const int32_t *t0 = pt + (size_t)rows[lo] * P;
    for (int32_t p = 0; p < P; ++p) s[p] = t0[p];
    for (int64_t k = lo + 1; k < hi; ++k) {
      const int32_t *t = pt + (size_t)rows[k] * P;
      for (int32_t p = 0; p < P; ++p) s[p] = t[p] < s[p] ? t[p] : s[p];
    }
  }
  free(pt);
}

/* ------------------------------------------------------------------------
 * a2. Band keys -- lsh.py:17-38.
 *
 * make_subvecs splits a signature into b bands of r = P / b values and casts
 * them to int16 (:28; wraps mod 2^16).  compute_buckets joins the decimal strings
 * with ',' (:33-34) and uses that as the dict key of band i.  Two bands get the
 * same string iff their int16 tuples are equal, so for r <= 4 the tuple packed
 * into 64 bits is an exact stand-in for the string:
 *     key = sum_k (sig[band*r + k] & 0xFFFF) << (16*k)
 * Returns 0, or -1 if P % b != 0 (the reference asserts, :20) or r > 4.
 * keys : [nq][b] uint64
 * ---------------------------------------------------------------------- */
QRO_API int qro_band_keys(const int32_t *sig, int64_t nq, int32_t P, int32_t b, uint64_t *keys) {
  if (b <= 0 || P % b != 0) return -1;
  int32_t r = P / b;
  if (r > 4) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < nq; ++q)
    for (int32_t i = 0; i < b; ++i) {
      uint64_t k = 0;
      for (int32_t j = 0; j < r; ++j)
        k |= (uint64_t)((uint32_t)sig[(size_t)q * P + i * r + j] & 0xFFFFu) << (16 * j);
      keys[(size_t)q * b + i] = k;
    }
  return 0;
}

/* the key of a band whose r values are all int16 -1: get_candidates skips such
 * buckets (lsh.py:45-47: keySet != {'-1'}) */
static inline uint64_t empty_key(int32_t r) { return r >= 4 ? ~0ull : ((1ull << (16 * r)) - 1); }

typedef struct { uint64_t key; uint32_t id; } rec_t;

static int cmp_rec(const void *a, const void *b) {
  const rec_t *x = (const rec_t *)a, *y = (const rec_t *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id);
}
static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y);
}

/* LSD radix sort of 64-bit words (used so the CPU baseline is not a qsort strawman) */
static void radix_sort_u64(uint64_t *a, int64_t n) {
  if (n < 2) return;
  uint64_t *tmp = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
  uint64_t *src = a, *dst = tmp;
  uint64_t orall = 0, andall = ~0ull;
  for (int64_t i = 0; i < n; ++i) { orall |= a[i]; andall &= a[i]; }
  uint64_t diff = orall ^ andall;
  for (int shift = 0; shift < 64; shift += 8) {
    if (((diff >> shift) & 0xFF) == 0) continue;
    int64_t cnt[257] = {0};
    for (int64_t i = 0; i < n; ++i) cnt[((src[i] >> shift) & 0xFF) + 1]++;
    for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
    for (int64_t i = 0; i < n; ++i) dst[cnt[(src[i] >> shift) & 0xFF]++] = src[i];
    uint64_t *t = src; src = dst; dst = t;
  }
  if (src != a) memcpy(a, src, (size_t)n * sizeof(uint64_t));
  free(tmp);
}

/* ------------------------------------------------------------------------
 * Parallel helpers of the candidate stage (so that the all-core CPU baseline is not gated by one thread):
 * words / records are dealt into NB buckets by a counting pass over per-thread chunks, then the buckets are
 * worked independently under OpenMP.
 * ---------------------------------------------------------------------- */
static inline uint64_t oracle_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int bucket_bits_for(int64_t n, int64_t per_bucket, int max_bits) {
  int bits = 0;
  while (bits < max_bits && (n >> bits) > per_bucket) ++bits;
  return bits;
}

/* sort + unique of n 64-bit words on all cores: the words are dealt into 2^bits ranges of their top `top_bits`
 * significant bits (a counting pass per thread chunk), every range is radix-sorted and de-duplicated by one thread,
 * the ranges are closed up.  The result is the sorted unique array (in `a`, count returned) -- what one serial
 * radix sort + unique gives. */
static int64_t parallel_sort_unique_u64(uint64_t *a, int64_t n) {
  if (n < 2) return n;
  uint64_t maxw = 0;
#pragma omp parallel for reduction(max : maxw) schedule(static)
  for (int64_t i = 0; i < n; ++i) maxw = a[i] > maxw ? a[i] : maxw;
  int sig_bits = 0;
  while (sig_bits < 64 && (maxw >> sig_bits) != 0) ++sig_bits;
  int bits = bucket_bits_for(n, 1 << 15, 14);
  if (bits > sig_bits) bits = sig_bits;
  if (bits == 0) {
    radix_sort_u64(a, n);
    int64_t u = 0;
    for (int64_t i = 0; i < n; ++i)
      if (i == 0 || a[i] != a[i - 1]) a[u++] = a[i];
    return u;
  }
  const int shift = sig_bits - bits;
  const int64_t NB = (int64_t)1 << bits;
  const int nt = qro_max_threads();
  int64_t *cnt = (int64_t *)calloc((size_t)nt * (size_t)NB, sizeof(int64_t));
  int64_t *bstart = (int64_t *)malloc((size_t)(NB + 1) * sizeof(int64_t));
  int64_t *bcount = (int64_t *)malloc((size_t)NB * sizeof(int64_t));
  uint64_t *tmp = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
    const int tid = 0, nth = 1;
#endif
    const int64_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
    int64_t *c = cnt + (size_t)tid * NB;
    for (int64_t i = lo; i < hi; ++i) c[a[i] >> shift]++;
#pragma omp barrier
#pragma omp single
    {
      int64_t run = 0;
      for (int64_t q = 0; q < NB; ++q) {
        bstart[q] = run;
        for (int t = 0; t < nth; ++t) {
          const int64_t v = cnt[(size_t)t * NB + q];
          cnt[(size_t)t * NB + q] = run;
          run += v;
        }
      }
      bstart[NB] = run;
    }
    for (int64_t i = lo; i < hi; ++i) tmp[c[a[i] >> shift]++] = a[i];
#pragma omp barrier
#pragma omp for schedule(dynamic, 4)
    for (int64_t q = 0; q < NB; ++q) {
      uint64_t *w = tmp + bstart[q];
      const int64_t m = bstart[q + 1] - bstart[q];
      radix_sort_u64(w, m);
      int64_t u = 0;
      for (int64_t i = 0; i < m; ++i)
        if (i == 0 || w[i] != w[i - 1]) w[u++] = w[i];
      bcount[q] = u;
    }
  }
  int64_t total = 0;
  for (int64_t q = 0; q < NB; ++q) { const int64_t u = bcount[q]; bcount[q] = total; total += u; }
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t q = 0; q < NB; ++q) {
    const int64_t u = (q + 1 < NB ? bcount[q + 1] : total) - bcount[q];
    if (u) memcpy(a + bcount[q], tmp + bstart[q], (size_t)u * sizeof(uint64_t));
  }
  free(tmp); free(cnt); free(bstart); free(bcount);
  return total;
}

/* growable per-thread pair buffers */
typedef struct { uint64_t *p; int64_t n, cap; } pvec_t;
static inline void pvec_reserve(pvec_t *v, int64_t extra) {
  const int64_t need = v->n + extra;
  if (need <= v->cap) return;
  int64_t nc = v->cap ? v->cap : 1024;
  while (nc < need) nc *= 2;
  v->p = (uint64_t *)realloc(v->p, (size_t)nc * sizeof(uint64_t));
  v->cap = nc;
}

/* ------------------------------------------------------------------------
 * a3. Candidate pairs -- lsh.py:40-55.
 *
 * For every band, every bucket with more than one member whose key is not the
 * all -1 tuple contributes all combinations(hits, 2) (:47-49); hits are in
 * insertion (= query id) order so each pair is (i, j) with i < j; the Python set
 * removes duplicates across bands (:41, :53).  The `reversed(c) in candidates`
 * test (:52) compares an iterator object and is always False.
 *
 * All cores: per band the (key, id) records are dealt into hash ranges of the key (equal keys share a range), the
 * ranges are sorted and their buckets' combinations written into per-thread buffers by independent threads; the
 * emitted words are then sorted + de-duplicated range-parallel (parallel_sort_unique_u64).
 *
 * Output: *pairs_out = malloc'ed sorted unique array of (i << 32 | j); returns the
 * count, or -1 on bad arguments.  Free with qro_free.
 * ---------------------------------------------------------------------- */
QRO_API int64_t qro_candidates(const uint64_t *keys, int64_t nq, int32_t b, int32_t r,
                               uint64_t **pairs_out) {
  *pairs_out = NULL;
  if (nq < 0 || b <= 0 || r <= 0 || r > 4) return -1;
  const uint64_t ek = empty_key(r);
  const int nt = qro_max_threads();
  pvec_t *tv = (pvec_t *)calloc((size_t)nt, sizeof(pvec_t));
  const int bits = bucket_bits_for(nq, 1 << 13, 16);
  const int64_t NB = (int64_t)1 << bits;
  rec_t *rec = (rec_t *)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(rec_t));
  int64_t *cnt = (int64_t *)malloc((size_t)nt * (size_t)NB * sizeof(int64_t));
  int64_t *bstart = (int64_t *)malloc((size_t)(NB + 1) * sizeof(int64_t));

  for (int32_t band = 0; band < b; ++band) {
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
      const int tid = 0, nth = 1;
#endif
      const int64_t lo = nq * tid / nth, hi = nq * (tid + 1) / nth;
      int64_t *c = cnt + (size_t)tid * NB;
      memset(c, 0, (size_t)NB * sizeof(int64_t));
      if (bits)
        for (int64_t q = lo; q < hi; ++q) c[oracle_mix64(keys[(size_t)q * b + band]) >> (64 - bits)]++;
      else
        c[0] = hi - lo;
#pragma omp barrier
#pragma omp single
      {
        int64_t run = 0;
        for (int64_t k = 0; k < NB; ++k) {
          bstart[k] = run;
          for (int t = 0; t < nth; ++t) {
            const int64_t v = cnt[(size_t)t * NB + k];
            cnt[(size_t)t * NB + k] = run;
            run += v;
          }
        }
        bstart[NB] = run;
      }
      for (int64_t q = lo; q < hi; ++q) {
        const uint64_t key = keys[(size_t)q * b + band];
        rec_t *d = &rec[c[bits ? oracle_mix64(key) >> (64 - bits) : 0]++];
        d->key = key;
        d->id = (uint32_t)q;
      }
#pragma omp barrier
      pvec_t *v = &tv[tid];
#pragma omp for schedule(dynamic, 4)
      for (int64_t k = 0; k < NB; ++k) {
        rec_t *w = rec + bstart[k];
        const int64_t m_all = bstart[k + 1] - bstart[k];
        if (m_all < 2) continue;
        qsort(w, (size_t)m_all, sizeof(rec_t), cmp_rec);
        int64_t s = 0;
        while (s < m_all) {
          int64_t e = s + 1;
          while (e < m_all && w[e].key == w[s].key) ++e;
          const int64_t m = e - s;
          if (m > 1 && w[s].key != ek) {
            pvec_reserve(v, m * (m - 1) / 2);
            uint64_t *o = v->p + v->n;
            for (int64_t x = s; x < e; ++x)
              for (int64_t y = x + 1; y < e; ++y) *o++ = ((uint64_t)w[x].id << 32) | w[y].id;
            v->n += m * (m - 1) / 2;
          }
          s = e;
        }
      }
    }
  }
  free(rec); free(cnt); free(bstart);
  int64_t total = 0;
  for (int t = 0; t < nt; ++t) total += tv[t].n;
  uint64_t *all = (uint64_t *)malloc((size_t)(total > 0 ? total : 1) * sizeof(uint64_t));
  int64_t o = 0;
  for (int t = 0; t < nt; ++t) {          /* (copies run in parallel: each thread's buffer has its own place) */
    const int64_t at = o;
    o += tv[t].n;
    tv[t].cap = at;
  }
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < nt; ++t) {
    if (tv[t].n) memcpy(all + tv[t].cap, tv[t].p, (size_t)tv[t].n * sizeof(uint64_t));
    free(tv[t].p);
  }
  free(tv);
  const int64_t u = parallel_sort_unique_u64(all, total);
  *pairs_out = all;
  return u;
}

/* ------------------------------------------------------------------------
 * Helpers of the large sharded tests (not part of the reference's algorithm): the words of a sorted pair list that
 * a given rank of a sharded run scores -- the numpy twin tests/dist_worker.py:pair_host states the rule
 * (csrc/common.h: qr_pair_host: the owner of i or of j, picked by the top bit of mix64(pair)) -- and the sorted
 * union of word lists, both on all cores.
 * ---------------------------------------------------------------------- */
QRO_API int64_t qro_filter_pair_host(const uint64_t *pairs, int64_t n, uint32_t shard, uint32_t rank, uint64_t *out) {
  const int nt = qro_max_threads();
  int64_t *cnt = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
    const int tid = 0, nth = 1;
#endif
    const int64_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
    int64_t c = 0;
    for (int64_t t = lo; t < hi; ++t) {
      const uint64_t w = pairs[t];
      const uint64_t id = (oracle_mix64(w) >> 63) ? (w & 0xFFFFFFFFull) : (w >> 32);
      c += id / shard == rank;
    }
    cnt[tid + 1] = c;
#pragma omp barrier
#pragma omp single
    for (int t = 0; t < nth; ++t) cnt[t + 1] += cnt[t];
    int64_t o = cnt[tid];
    for (int64_t t = lo; t < hi; ++t) {
      const uint64_t w = pairs[t];
      const uint64_t id = (oracle_mix64(w) >> 63) ? (w & 0xFFFFFFFFull) : (w >> 32);
      if (id / shard == rank) out[o++] = w;
    }
  }
  int64_t total = 0;
  for (int t = 0; t <= nt; ++t) total = cnt[t] > total ? cnt[t] : total;
  free(cnt);
  return total;
}

QRO_API int64_t qro_sort_unique_u64(uint64_t *a, int64_t n) { return parallel_sort_unique_u64(a, n); }

/* Same as qro_candidates, for ANY band width r = P / b: buckets are grouped by comparing the
 * int16 tuples themselves (no 64-bit packing), straight from the signature matrix. */
typedef struct { const int32_t *sig; int32_t P, off, r; } tuple_ctx_t;
static __thread tuple_ctx_t g_tc;
static int cmp_tuple(const void *a, const void *b) {
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  const int32_t *sx = g_tc.sig + (size_t)x * g_tc.P + g_tc.off, *sy = g_tc.sig + (size_t)y * g_tc.P + g_tc.off;
  for (int32_t t = 0; t < g_tc.r; ++t) {
    const uint32_t vx = (uint32_t)sx[t] & 0xFFFFu, vy = (uint32_t)sy[t] & 0xFFFFu;
    if (vx != vy) return vx < vy ? -1 : 1;
  }
  return x < y ? -1 : (x > y);
}

QRO_API int64_t qro_candidates_from_sig(const int32_t *sig, int64_t nq, int32_t P, int32_t b, uint64_t **pairs_out) {
  *pairs_out = NULL;
  if (nq < 0 || b <= 0 || P % b != 0) return -1;
  const int32_t r = P / b;
  uint32_t *idx = (uint32_t *)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(uint32_t));
  uint64_t *all = NULL;
  int64_t n = 0, cap = 0;
  for (int32_t band = 0; band < b; ++band) {
    for (int64_t q = 0; q < nq; ++q) idx[q] = (uint32_t)q;
    g_tc.sig = sig; g_tc.P = P; g_tc.off = band * r; g_tc.r = r;
    qsort(idx, (size_t)nq, sizeof(uint32_t), cmp_tuple);
    int64_t s = 0;
    while (s < nq) {
      int64_t e = s + 1;
      const int32_t *ss = sig + (size_t)idx[s] * P + band * r;
      while (e < nq) {
        const int32_t *se = sig + (size_t)idx[e] * P + band * r;
        int same = 1;
        for (int32_t t = 0; t < r; ++t)
          if (((uint32_t)ss[t] & 0xFFFFu) != ((uint32_t)se[t] & 0xFFFFu)) { same = 0; break; }
        if (!same) break;
        ++e;
      }
      int empty = 1;
      for (int32_t t = 0; t < r; ++t)
        if (((uint32_t)ss[t] & 0xFFFFu) != 0xFFFFu) { empty = 0; break; }
      const int64_t m = e - s;
      if (m > 1 && !empty) {
        const int64_t need = n + m * (m - 1) / 2;
        if (need > cap) {
          cap = cap ? cap : 1024;
          while (cap < need) cap *= 2;
          all = (uint64_t *)realloc(all, (size_t)cap * sizeof(uint64_t));
        }
        for (int64_t x = s; x < e; ++x)
          for (int64_t y = x + 1; y < e; ++y) all[n++] = ((uint64_t)idx[x] << 32) | idx[y];
      }
      s = e;
    }
  }
  free(idx);
  if (!all) all = (uint64_t *)malloc(sizeof(uint64_t));
  const int64_t u = parallel_sort_unique_u64(all, n);
  *pairs_out = all;
  return u;
}

/* number of pairs before cross-band de-duplication (what the per-band
 * combinations() loops generate in total, lsh.py:49) */
QRO_API int64_t qro_emitted_pairs(const uint64_t *keys, int64_t nq, int32_t b, int32_t r) {
  if (nq <= 0 || b <= 0 || r <= 0 || r > 4) return 0;
  const uint64_t ek = empty_key(r);
  int64_t total = 0;
  uint64_t *col = (uint64_t *)malloc((size_t)nq * sizeof(uint64_t));
  for (int32_t band = 0; band < b; ++band) {
    for (int64_t q = 0; q < nq; ++q) col[q] = keys[(size_t)q * b + band];
    qsort(col, (size_t)nq, sizeof(uint64_t), cmp_u64);
    int64_t s = 0;
    while (s < nq) {
      int64_t e = s + 1;
      while (e < nq && col[e] == col[s]) ++e;
      if (col[s] != ek) total += (e - s) * (e - s - 1) / 2;
      s = e;
    }
  }
  free(col);
  return total;
}

/* ------------------------------------------------------------------------
 * a5 (first half). Pair scoring -- recommender.py:203-204.
 *
 *   values = np.around(cosine_similarity([sig_i, sig_j1, ...])[0][1:], 3)
 *
 * sklearn's cosine_similarity L2-normalises each row in float64 (a zero row is
 * left as zeros) and multiplies the normalised rows.  np.around(x, 3) is
 * rint(x * 1000) / 1000.
 *
 * mode 0 ("sklearn order"): normalise, then sum of products of the normalised
 *         entries -- the reference's arithmetic up to BLAS summation order.
 * mode 1 ("exact"): integer dot and squared norms (exact in int64), then
 *         dot / (sqrt(na) * sqrt(nb)) -- what the HIP kernel computes.
 * The two differ by a few ulp before rounding and are checked equal after it
 * (tests/test_oracle_golden.py) on every golden pair.
 *
 * milli[n] = (int) rint(cos * 1000);  cosv[n] (optional) = the unrounded cosine.
 * ---------------------------------------------------------------------- */
QRO_API void qro_score_pairs(const int32_t *sig, int32_t P, const uint64_t *pairs, int64_t n,
                             int32_t mode, int32_t *milli, double *cosv) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < n; ++t) {
    const int32_t *a = sig + (size_t)(pairs[t] >> 32) * P;
    const int32_t *c = sig + (size_t)(pairs[t] & 0xFFFFFFFFu) * P;
    int64_t dot = 0, na = 0, nb = 0;
    for (int32_t p = 0; p < P; ++p) {
      dot += (int64_t)a[p] * c[p];
      na += (int64_t)a[p] * a[p];
      nb += (int64_t)c[p] * c[p];
    }
    double cs;
    if (na == 0 || nb == 0) {
      cs = 0.0;
    } else if (mode == 0) {
      double ia = sqrt((double)na), ib = sqrt((double)nb), acc = 0.0;
      for (int32_t p = 0; p < P; ++p) acc += ((double)a[p] / ia) * ((double)c[p] / ib);
      cs = acc;
    } else {
      cs = (double)dot / (sqrt((double)na) * sqrt((double)nb));
    }
    milli[t] = (int32_t)rint(cs * 1000.0);
    if (cosv) cosv[t] = cs;
  }
}

/* ------------------------------------------------------------------------
 * a5 (second half). Per-query top-K -- recommender.py:185-210.
 *
 * Every candidate pair (i, j) makes j a neighbour of i and i a neighbour of j
 * (:198-199).  Per query the neighbours are ordered by rounded value, descending,
 * and the first K kept (:206).  The reference's order among equal values is
 * whatever np.argsort and Python set iteration produce (arbitrary); this
 * restatement fixes it: value descending, then neighbour id ascending.
 *
 * Output (caller allocates 2*n each): directed edges sorted by (src, value desc,
 * dst asc), at most K per src.  Returns the number of edges kept.
 * ---------------------------------------------------------------------- */
typedef struct { uint64_t hi, lo; } k2_t;
static int cmp_k2(const void *a, const void *b) {
  const k2_t *x = (const k2_t *)a, *y = (const k2_t *)b;
  if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
  return x->lo < y->lo ? -1 : (x->lo > y->lo);
}

QRO_API int64_t qro_topk(const uint64_t *pairs, const int32_t *milli, int64_t n, int32_t K,
                         int32_t *src, int32_t *dst, int32_t *val) {
  /* sortable 16-byte records: hi = src << 32 | (2000 - (milli + 1000)), lo = dst.  The records are
   * first dealt into NB ranges of src (a counting pass per thread chunk), the ranges sorted independently (OpenMP),
   * so the whole array ends up in (src, inv, dst) order -- same result as one qsort (the records are distinct), on
   * all cores; the cut to K per src is made per range (a src never straddles two ranges) and written in place. */
  const int64_t m = 2 * n;
  k2_t *k = (k2_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(k2_t));
  uint32_t maxid = 0;
#pragma omp parallel for reduction(max : maxid) schedule(static)
  for (int64_t t = 0; t < n; ++t) {
    uint32_t j = (uint32_t)(pairs[t] & 0xFFFFFFFFu); /* i < j */
    if (j > maxid) maxid = j;
  }
  enum { NB = 4096 };
  const uint64_t span = (uint64_t)maxid + 1;
  const int nt = qro_max_threads();
  int64_t *start = (int64_t *)calloc(NB + 1, sizeof(int64_t));
  int64_t *cnt = (int64_t *)calloc((size_t)nt * NB, sizeof(int64_t));
  int64_t *kept = (int64_t *)calloc(NB + 1, sizeof(int64_t));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
    const int tid = 0, nth = 1;
#endif
    const int64_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
    int64_t *c = cnt + (size_t)tid * NB;
    for (int64_t t = lo; t < hi; ++t) {
      uint32_t i = (uint32_t)(pairs[t] >> 32), j = (uint32_t)(pairs[t] & 0xFFFFFFFFu);
      c[(uint64_t)i * NB / span]++;
      c[(uint64_t)j * NB / span]++;
    }
#pragma omp barrier
#pragma omp single
    {
      int64_t run = 0;
      for (int q = 0; q < NB; ++q) {
        start[q] = run;
        for (int t = 0; t < nth; ++t) {
          const int64_t v = cnt[(size_t)t * NB + q];
          cnt[(size_t)t * NB + q] = run;
          run += v;
        }
      }
      start[NB] = run;
    }
    for (int64_t t = lo; t < hi; ++t) {
      uint32_t i = (uint32_t)(pairs[t] >> 32), j = (uint32_t)(pairs[t] & 0xFFFFFFFFu);
      uint32_t inv = (uint32_t)(1000 - milli[t]); /* milli in [-1000, 1000] -> inv in [0, 2000] */
      k2_t *a = &k[c[(uint64_t)i * NB / span]++], *e = &k[c[(uint64_t)j * NB / span]++];
      a->hi = ((uint64_t)i << 32) | inv; a->lo = j;
      e->hi = ((uint64_t)j << 32) | inv; e->lo = i;
    }
#pragma omp barrier
#pragma omp for schedule(dynamic, 8)
    for (int q = 0; q < NB; ++q) {
      k2_t *w = k + start[q];
      const int64_t len = start[q + 1] - start[q];
      if (len > 1) qsort(w, (size_t)len, sizeof(k2_t), cmp_k2);
      /* the cut: the first K records of every src, closed up at the front of the range */
      int64_t out = 0, run = 0;
      uint32_t cs = 0;
      for (int64_t t = 0; t < len; ++t) {
        uint32_t s = (uint32_t)(w[t].hi >> 32);
        if (t == 0 || s != cs) { cs = s; run = 0; }
        if (run < K) w[out++] = w[t];
        ++run;
      }
      kept[q + 1] = out;
    }
  }
  for (int q = 0; q < NB; ++q) kept[q + 1] += kept[q];
#pragma omp parallel for schedule(dynamic, 8)
  for (int q = 0; q < NB; ++q) {
    const k2_t *w = k + start[q];
    const int64_t at = kept[q], len = kept[q + 1] - kept[q];
    for (int64_t t = 0; t < len; ++t) {
      src[at + t] = (int32_t)(uint32_t)(w[t].hi >> 32);
      dst[at + t] = (int32_t)w[t].lo;
      val[at + t] = 1000 - (int32_t)(uint32_t)(w[t].hi & 0xFFFFFFFFu);
    }
  }
  const int64_t out = kept[NB];
  free(cnt); free(start); free(kept); free(k);
  return out;
}

/* ------------------------------------------------------------------------
 * Synthetic answer sets: a pure function of (seed, q) so
 * the CPU, the GPU and every shard regenerate identical rows.  Not part of the
 * reference; bench / test input only.  The HIP twin is qrlsh_synth_* and must be
 * bit-identical (tests/test_gpu_parity.py).
 *
 *   nb      = max(1, nq / cluster) base sets
 *   |base|  = clamp(Poisson(mean) via the caller's 32-bit CDF table, 1, 48)
 *   base[k] = uniform in [0, D)
 *   query q = base (q mod nb) with each element independently replaced by a
 *             uniform draw when a 24-bit draw < rep_thresh24; then sort-unique
 * ---------------------------------------------------------------------- */
#define QRO_MAXS 48

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t idx, uint64_t ctr) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (stream + 1);
  x = mix64(x ^ (idx * 0xD1342543DE82EF95ull));
  x = mix64(x + ctr * 0xA24BAED4963EE407ull + 0x9E3779B97F4A7C15ull);
  return x;
}
static inline uint32_t to_range(uint64_t r, uint32_t D) {
  return (uint32_t)(((r >> 32) * (uint64_t)D) >> 32);
}

static int synth_one(uint64_t seed, int64_t q, int64_t nb, uint32_t D, const uint32_t *cdf,
                     int32_t ncdf, uint32_t rep_thresh24, uint32_t *out) {
  uint64_t bidx = (uint64_t)(q % nb);
  uint32_t u = (uint32_t)(rnd(seed, 1, bidx, 0) >> 32);
  int32_t size = 0;
  while (size < ncdf && u >= cdf[size]) ++size;
  if (size < 1) size = 1;
  if (size > QRO_MAXS) size = QRO_MAXS;
  int n = 0;
  for (int32_t k = 0; k < size; ++k) {
    uint32_t e = to_range(rnd(seed, 2, bidx, (uint64_t)k), D);
    uint64_t rr = rnd(seed, 3, (uint64_t)q, (uint64_t)k);
    if ((uint32_t)(rr & 0xFFFFFFu) < rep_thresh24) e = to_range(rnd(seed, 4, (uint64_t)q, (uint64_t)k), D);
    /* insertion into sorted unique list */
    int pos = n;
    while (pos > 0 && out[pos - 1] > e) --pos;
    if (pos > 0 && out[pos - 1] == e) continue;
    for (int m = n; m > pos; --m) out[m] = out[m - 1];
    out[pos] = e;
    ++n;
  }
  return n;
}

QRO_API void qro_synth_sizes(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total,
                             int32_t cluster, uint32_t D, const uint32_t *cdf, int32_t ncdf,
                             uint32_t rep_thresh24, int32_t *sizes) {
  int64_t nb = nq_total / cluster; if (nb < 1) nb = 1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq_local; ++i) {
    uint32_t tmp[QRO_MAXS];
    sizes[i] = synth_one(seed, q0 + i, nb, D, cdf, ncdf, rep_thresh24, tmp);
  }
}

QRO_API void qro_synth_fill(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total,
                            int32_t cluster, uint32_t D, const uint32_t *cdf, int32_t ncdf,
                            uint32_t rep_thresh24, const int64_t *offsets, int32_t *rows) {
  int64_t nb = nq_total / cluster; if (nb < 1) nb = 1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nq_local; ++i) {
    uint32_t tmp[QRO_MAXS];
    int n = synth_one(seed, q0 + i, nb, D, cdf, ncdf, rep_thresh24, tmp);
    for (int k = 0; k < n; ++k) rows[offsets[i] + k] = (int32_t)tmp[k];
  }
}
