// pairs.hip -- a3 candidate-pair enumeration and the two stream compactions (unique, top-K).
//
// Reference: LSH.get_candidates, lsh.py:40-55 (per bucket: combinations(hits, 2), skipping
// single-member buckets and the all -1 key; a Python set removes cross-band duplicates),
// and the per-query cut argsort(values)[::-1][:K], recommender.py:206-210.
//
// Input of the enumeration is, per band, the keys sorted by the top 32 bits of mix64(key)
// together with their query ids (stable sort => ids ascending inside a run).  Record t
// pairs with every earlier record u of its mix-run whose FULL key equals its own, so the
// (rare) 32-bit mix collisions cost a compare and never a false pair.  Everything is
// count-then-fill with per-workgroup offsets; nothing is allocated here.
#include "common.h"

constexpr int PAIR_THREADS = 256;
constexpr int PAIR_IPT = 4;
constexpr int PAIR_TILE = PAIR_THREADS * PAIR_IPT;  // records per workgroup (blocked: thread t owns 4 consecutive)

constexpr int CMP_THREADS = 256;
constexpr int CMP_IPT = 8;
constexpr int CMP_TILE = CMP_THREADS * CMP_IPT;

constexpr int PAIR_HALO = 128;  // records before the tile that are staged in LDS as well

__device__ static inline uint32_t mix_hi(uint64_t k, int hsh) { return (uint32_t)(qr_mix64(k) >> hsh); }

// Stage keys (and their hash-run ids) of [tile_start - HALO, tile_start + TILE) in LDS.
__device__ static inline void stage_tile(const uint64_t *__restrict__ k, int64_t nq, int64_t tile_start, int hsh,
                                         uint64_t *sk, uint32_t *sh) {
  for (int idx = threadIdx.x; idx < PAIR_HALO + PAIR_TILE; idx += PAIR_THREADS) {
    const int64_t g = tile_start - PAIR_HALO + idx;
    uint64_t key = 0;
    if (g >= 0 && g < nq) key = k[g];
    sk[idx] = key;
    sh[idx] = mix_hi(key, hsh);
  }
}

// Number of earlier records of the same band whose key equals record t's.  The walk runs
// backwards through the hash-run in LDS; a run longer than the halo continues in global memory.
__device__ static inline uint32_t count_back(const uint64_t *__restrict__ k, const uint64_t *sk, const uint32_t *sh,
                                             int64_t tile_start, int tl, uint64_t ek, int hsh) {
  const uint64_t kt = sk[PAIR_HALO + tl];
  if (kt == ek) return 0;
  const uint32_t ht = sh[PAIR_HALO + tl];
  uint32_t c = 0;
  int idx = PAIR_HALO + tl - 1;
  const int lo = (tile_start >= PAIR_HALO) ? 0 : (int)(PAIR_HALO - tile_start);  // first staged index that exists
  for (; idx >= lo; --idx) {
    if (sk[idx] == kt) ++c;
    else if (sh[idx] != ht) return c;
  }
  for (int64_t u = tile_start - PAIR_HALO - 1; u >= 0; --u) {  // rare: run longer than the halo
    const uint64_t ku = k[u];
    if (ku == kt) ++c;
    else if (mix_hi(ku, hsh) != ht) break;
  }
  return c;
}

__global__ __launch_bounds__(PAIR_THREADS) void pairs_count_kernel(const uint64_t *__restrict__ keys, int64_t nq,
                                                                   int ntiles, uint64_t ek, int hsh,
                                                                   uint64_t *__restrict__ blk) {
  __shared__ uint64_t sk[PAIR_HALO + PAIR_TILE];
  __shared__ uint32_t sh[PAIR_HALO + PAIR_TILE];
  __shared__ uint64_t sm[4];
  const int tile = blockIdx.x, band = blockIdx.y;
  const uint64_t *k = keys + (size_t)band * nq;
  const int64_t tile_start = (int64_t)tile * PAIR_TILE;
  stage_tile(k, nq, tile_start, hsh, sk, sh);
  __syncthreads();
  const int tl0 = threadIdx.x * PAIR_IPT;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < PAIR_IPT; ++i)
    if (tile_start + tl0 + i < nq) c += count_back(k, sk, sh, tile_start, tl0 + i, ek, hsh);
  uint64_t total;
  (void)block_excl_scan_u64_256(c, sm, &total);
  if (threadIdx.x == 0) blk[(size_t)band * ntiles + tile] = total;
}

__global__ __launch_bounds__(PAIR_THREADS) void pairs_fill_kernel(const uint64_t *__restrict__ keys,
                                                                  const uint32_t *__restrict__ ids, int64_t nq,
                                                                  int ntiles, uint64_t ek, int hsh,
                                                                  const uint64_t *__restrict__ blk,
                                                                  uint64_t *__restrict__ out) {
  __shared__ uint64_t sk[PAIR_HALO + PAIR_TILE];
  __shared__ uint32_t sh[PAIR_HALO + PAIR_TILE];
  __shared__ uint32_t si[PAIR_HALO + PAIR_TILE];
  __shared__ uint64_t sm[4];
  const int tile = blockIdx.x, band = blockIdx.y;
  const uint64_t *k = keys + (size_t)band * nq;
  const uint32_t *id = ids + (size_t)band * nq;
  const int64_t tile_start = (int64_t)tile * PAIR_TILE;
  stage_tile(k, nq, tile_start, hsh, sk, sh);
  for (int idx = threadIdx.x; idx < PAIR_HALO + PAIR_TILE; idx += PAIR_THREADS) {
    const int64_t g = tile_start - PAIR_HALO + idx;
    si[idx] = (g >= 0 && g < nq) ? id[g] : 0u;
  }
  __syncthreads();
  const int tl0 = threadIdx.x * PAIR_IPT;
  uint32_t c[PAIR_IPT];
  uint64_t mine = 0;
#pragma unroll
  for (int i = 0; i < PAIR_IPT; ++i) {
    c[i] = (tile_start + tl0 + i < nq) ? count_back(k, sk, sh, tile_start, tl0 + i, ek, hsh) : 0;
    mine += c[i];
  }
  uint64_t total;
  uint64_t pos = blk[(size_t)band * ntiles + tile] + block_excl_scan_u64_256(mine, sm, &total);
#pragma unroll
  for (int i = 0; i < PAIR_IPT; ++i) {
    if (c[i] == 0) continue;
    const int tl = tl0 + i;
    const uint64_t kt = sk[PAIR_HALO + tl];
    const uint32_t it = si[PAIR_HALO + tl];
    uint32_t left = c[i];
    for (int idx = PAIR_HALO + tl - 1; left > 0 && idx >= 0; --idx) {
      if (sk[idx] == kt) {
        const uint32_t iu = si[idx];
        out[pos++] = ((uint64_t)(iu < it ? iu : it) << 32) | (iu < it ? it : iu);
        --left;
      }
    }
    for (int64_t u = tile_start - PAIR_HALO - 1; left > 0; --u) {  // rare: run longer than the halo
      if (k[u] == kt) {
        const uint32_t iu = id[u];
        out[pos++] = ((uint64_t)(iu < it ? iu : it) << 32) | (iu < it ? it : iu);
        --left;
      }
    }
  }
}

// ---- stream compaction over a sorted uint64 array -------------------------------------
// UNIQUE: keep the first of every run of equal words.
// TOPK  : keep a directed edge iff fewer than K earlier edges share its src (edges are sorted
//         by src, then value desc, then dst), i.e. iff t < K or src(a[t-K]) != src(a[t]).
enum { PRED_UNIQUE = 0, PRED_TOPK = 1 };

template <int PRED> __device__ static inline bool keep_at(const uint64_t *__restrict__ a, int64_t t, int K, int sh) {
  if (PRED == PRED_UNIQUE) return t == 0 || a[t] != a[t - 1];
  return t < K || (a[t - K] >> sh) != (a[t] >> sh);
}

// Tile = CMP_TILE words; wave w owns the contiguous slice [w*512, (w+1)*512) and sweeps it 64
// words at a time, so loads are coalesced and the kept words of a sweep take consecutive output
// positions (ballot + popcount): stores are coalesced too.
constexpr int CMP_WAVES = CMP_THREADS / WAVE;
constexpr int CMP_PER_WAVE = CMP_TILE / CMP_WAVES;

template <int PRED>
__global__ __launch_bounds__(CMP_THREADS) void compact_count_kernel(const uint64_t *__restrict__ a, int64_t n, int K,
                                                                    int sh, uint64_t *__restrict__ blk) {
  __shared__ uint32_t wtot[CMP_WAVES];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * CMP_TILE + (int64_t)w * CMP_PER_WAVE;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < CMP_PER_WAVE / WAVE; ++i) {
    const int64_t t = base + i * WAVE + lane;
    const bool keep = t < n && keep_at<PRED>(a, t, K, sh);
    c += (uint32_t)__popcll(__ballot(keep));
  }
  if (lane == 0) wtot[w] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t tot = 0;
    for (int i = 0; i < CMP_WAVES; ++i) tot += wtot[i];
    blk[blockIdx.x] = tot;
  }
}

template <int PRED>
__global__ __launch_bounds__(CMP_THREADS) void compact_fill_kernel(const uint64_t *__restrict__ a, int64_t n, int K,
                                                                   int sh, int id_bits,
                                                                   const uint32_t *__restrict__ vals,
                                                                   const uint64_t *__restrict__ blk,
                                                                   uint64_t *__restrict__ out_u64,
                                                                   int32_t *__restrict__ src_out,
                                                                   int32_t *__restrict__ dst_out,
                                                                   int32_t *__restrict__ milli_out,
                                                                   int32_t src_base = 0) {
  __shared__ uint32_t wtot[CMP_WAVES];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * CMP_TILE + (int64_t)w * CMP_PER_WAVE;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  uint64_t v[CMP_PER_WAVE / WAVE];
  uint64_t masks[CMP_PER_WAVE / WAVE];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < CMP_PER_WAVE / WAVE; ++i) {
    const int64_t t = base + i * WAVE + lane;
    v[i] = t < n ? a[t] : 0;
    const bool keep = t < n && keep_at<PRED>(a, t, K, sh);
    masks[i] = __ballot(keep);
    c += (uint32_t)__popcll(masks[i]);
  }
  if (lane == 0) wtot[w] = c;
  __syncthreads();
  uint64_t pos = blk[blockIdx.x];
  for (int i = 0; i < w; ++i) pos += wtot[i];
  const uint64_t idm = (1ull << id_bits) - 1ull;
#pragma unroll
  for (int i = 0; i < CMP_PER_WAVE / WAVE; ++i) {
    if ((masks[i] >> lane) & 1ull) {
      const uint64_t p = pos + (uint64_t)__popcll(masks[i] & lt_mask);
      if (PRED == PRED_UNIQUE) {
        out_u64[p] = v[i];
      } else {
        src_out[p] = (int32_t)(v[i] >> sh) + src_base;
        if (vals) {  // wide ids: key = src << 11 | inv, dst rides as the payload
          dst_out[p] = (int32_t)vals[base + i * WAVE + lane];
          milli_out[p] = 1000 - (int32_t)(v[i] & 0x7FFull);
        } else {
          dst_out[p] = (int32_t)(v[i] & idm);
          milli_out[p] = 1000 - (int32_t)((v[i] >> id_bits) & 0x7FFull);
        }
      }
    }
    pos += (uint64_t)__popcll(masks[i]);
  }
}

// ---------------------------------------------------------------------------------------
QRLSH_EXPORT size_t qrlsh_pairs_workspace_bytes(int64_t nq, int32_t b) {
  if (nq <= 0 || b <= 0) return 16;
  return (size_t)b * ceil_div64(nq, PAIR_TILE) * sizeof(uint64_t);
}

QRLSH_EXPORT int qrlsh_pairs_count(const uint64_t *sorted_keys, int64_t nq, int32_t b, int32_t r, int32_t hash_bits,
                                   void *workspace,
                                   size_t workspace_bytes, uint64_t *total_out, void *stream) {
  QR_CHECK_ARG(nq >= 0 && b > 0 && r > 0, "qrlsh_pairs_count: bad sizes nq=%lld b=%d r=%d", (long long)nq,
               b, r);
  QR_CHECK_ARG(hash_bits >= 8 && hash_bits <= 32, "qrlsh_pairs_count: hash_bits=%d not in [8,32]", hash_bits);
  QR_CHECK_ARG(total_out && workspace, "qrlsh_pairs_count: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nq == 0) {
    if (hipMemsetAsync(total_out, 0, sizeof(uint64_t), st) != hipSuccess) {
      qrlsh_set_error("hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    return QRLSH_OK;
  }
  QR_CHECK_ARG(sorted_keys, "qrlsh_pairs_count: null keys");
  if (workspace_bytes < qrlsh_pairs_workspace_bytes(nq, b)) {
    qrlsh_set_error("qrlsh_pairs_count: workspace %zu < %zu bytes", workspace_bytes,
                    qrlsh_pairs_workspace_bytes(nq, b));
    return QRLSH_EWORKSPACE;
  }
  const int ntiles = (int)ceil_div64(nq, PAIR_TILE);
  uint64_t *blk = static_cast<uint64_t *>(workspace);
  QR_LAUNCH("pairs_count", pairs_count_kernel, dim3(ntiles, b), dim3(PAIR_THREADS), 0, st, sorted_keys, nq, ntiles,
                     qr_empty_key(r), 64 - hash_bits, blk);
  QR_LAUNCH("scan_blocks", scan_u64_kernel, dim3(1), dim3(1024), 0, st, blk, (int64_t)ntiles * b, total_out);
  QR_LAUNCH_CHECK("qrlsh_pairs_count");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_pairs_fill(const uint64_t *sorted_keys, const uint32_t *sorted_ids, int64_t nq, int32_t b,
                                  int32_t r, int32_t hash_bits, const void *workspace, uint64_t *pairs_out,
                                  void *stream) {
  QR_CHECK_ARG(nq >= 0 && b > 0 && r > 0 && hash_bits >= 8 && hash_bits <= 32, "qrlsh_pairs_fill: bad sizes");
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(sorted_keys && sorted_ids && workspace && pairs_out, "qrlsh_pairs_fill: null pointer");
  const int ntiles = (int)ceil_div64(nq, PAIR_TILE);
  QR_LAUNCH("pairs_fill", pairs_fill_kernel, dim3(ntiles, b), dim3(PAIR_THREADS), 0, static_cast<hipStream_t>(stream),
                     sorted_keys, sorted_ids, nq, ntiles, qr_empty_key(r), 64 - hash_bits,
                     static_cast<const uint64_t *>(workspace),
                     pairs_out);
  QR_LAUNCH_CHECK("qrlsh_pairs_fill");
  return QRLSH_OK;
}

// workspace: per-tile counts [nblk] | chunk totals of the large-array scan
QRLSH_EXPORT size_t qrlsh_compact_workspace_bytes(int64_t n) {
  if (n <= 0) return 16;
  const int64_t nblk = ceil_div64(n, CMP_TILE);
  return (size_t)(nblk + ceil_div64(nblk, SCANL_CHUNK) + 1) * sizeof(uint64_t);
}

template <int PRED>
static int compact_count(const uint64_t *a, int64_t n, int K, int sh, void *workspace, size_t workspace_bytes,
                         uint64_t *total_out, void *stream, const char *name) {
  QR_CHECK_ARG(n >= 0 && total_out && workspace, "%s: bad arguments", name);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (hipMemsetAsync(total_out, 0, sizeof(uint64_t), st) != hipSuccess) {
      qrlsh_set_error("hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    return QRLSH_OK;
  }
  QR_CHECK_ARG(a, "%s: null input", name);
  if (workspace_bytes < qrlsh_compact_workspace_bytes(n)) {
    qrlsh_set_error("%s: workspace %zu < %zu bytes", name, workspace_bytes, qrlsh_compact_workspace_bytes(n));
    return QRLSH_EWORKSPACE;
  }
  const int64_t nblk = ceil_div64(n, CMP_TILE);
  uint64_t *blk = static_cast<uint64_t *>(workspace);
  QR_LAUNCH(PRED == PRED_UNIQUE ? "unique_count" : "topk_count", (compact_count_kernel<PRED>), dim3((unsigned)nblk), dim3(CMP_THREADS), 0, st, a, n, K, sh, blk);
  qr_scan_u64(blk, nblk, total_out, blk + nblk, st);
  QR_LAUNCH_CHECK(name);
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_unique_count(const uint64_t *sorted, int64_t n, void *workspace, size_t workspace_bytes,
                                    uint64_t *total_out, void *stream) {
  return compact_count<PRED_UNIQUE>(sorted, n, 0, 0, workspace, workspace_bytes, total_out, stream,
                                    "qrlsh_unique_count");
}

QRLSH_EXPORT int qrlsh_unique_fill(const uint64_t *sorted, int64_t n, const void *workspace, uint64_t *out,
                                   void *stream) {
  QR_CHECK_ARG(n >= 0, "qrlsh_unique_fill: bad n");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(sorted && workspace && out, "qrlsh_unique_fill: null pointer");
  QR_LAUNCH("unique_fill", (compact_fill_kernel<PRED_UNIQUE>), dim3((unsigned)ceil_div64(n, CMP_TILE)), dim3(CMP_THREADS), 0,
                     static_cast<hipStream_t>(stream), sorted, n, 0, 0, 0, (const uint32_t *)nullptr,
                     static_cast<const uint64_t *>(workspace), out, nullptr, nullptr, nullptr);
  QR_LAUNCH_CHECK("qrlsh_unique_fill");
  return QRLSH_OK;
}

// ---- a3 tail, fast form: sorted unique pairs from pairs GROUPED BY i ---------------------------
// The emitted pairs carry every candidate once per band it collides in (5.5x on the config-2
// workload), and a full (i, j) radix sort of all of them only to drop the repeats is the
// largest block of sort passes in the pipeline.  Here the pairs are sorted on i's bits only
// (ceil(id_bits / 8) passes instead of ceil(2 id_bits / 8)); a row (all pairs of one i) is then
// a short run -- tens of words -- that is de-duplicated and ordered by j inside LDS:
//   1. the row's own span of an LDS array serves as an open-addressing hash set of its j values
//      (as many slots as the row has words; ds_cmpst claims a slot or finds the value present);
//   2. a workgroup prefix sum over the occupied slots packs the distinct values, row by row;
//   3. the place of a distinct value is the number of smaller ones in its (now short) packed row.
//
// A workgroup owns the rows whose first word lies in its RD_C-word chunk; it loads RD_CAP words
// beyond the chunk so that the last owned row is complete (a longer overhang sets the overflow
// flag: the caller then uses the general sort + qrlsh_unique path).  The kept words of a
// workgroup are written, in order, into `tmp` starting at its first owned word (owned ranges
// tile the input, so these never overlap); per-workgroup counts are scanned and a second small
// kernel closes the gaps.
// A "row" may also be a GROUP of 2^gbits consecutive i (the words are then ordered by i >> gbits only,
// which can save the grouping sort its last pass): the value that is de-duplicated and ordered inside
// a row is then (i's low gbits, j) packed into 32 bits, j < 2^jbits.
struct RowSplit {
  int gbits, jbits;
  __device__ uint32_t row(uint64_t x) const { return (uint32_t)(x >> (32 + gbits)); }
  __device__ uint32_t val(uint64_t x) const {
    const uint32_t j = (uint32_t)x;
    return gbits ? ((uint32_t)(x >> 32) & ((1u << gbits) - 1u)) << jbits | j : j;
  }
  // the word of value v in the row that word x0 belongs to
  __device__ uint64_t word(uint64_t x0, uint32_t v) const {
    if (!gbits) return (x0 & 0xFFFFFFFF00000000ull) | v;
    const uint64_t i = ((x0 >> 32) & ~(uint64_t)((1u << gbits) - 1u)) | (v >> jbits);
    return i << 32 | (v & ((1u << jbits) - 1u));
  }
};

constexpr int RD_THREADS = 512;
constexpr int RD_C = 2048;
constexpr int RD_CAP = 1024;
constexpr int RD_IMG = RD_C + RD_CAP;
constexpr int RD_PER = RD_IMG / RD_THREADS;
constexpr uint32_t RD_EMPTY = 0xFFFFFFFFu;  // never a j (ids are non-negative int32)

__global__ __launch_bounds__(RD_THREADS) void row_unique_kernel(const uint64_t *__restrict__ in, int64_t n,
                                                                uint64_t *__restrict__ tmp,
                                                                uint64_t *__restrict__ counts,
                                                                uint64_t *__restrict__ starts,
                                                                uint64_t *__restrict__ longlist,
                                                                unsigned long long *__restrict__ nlong, int gbits,
                                                                int jbits) {
  __shared__ uint32_t lo[RD_IMG];    // j of every word; later: the packed distinct values
  __shared__ uint32_t tab[RD_IMG];   // hash sets, one per owned row, over the row's own span
  __shared__ uint16_t rs[RD_IMG];    // row start + 1 of the row a word belongs to, 0 = row began before the image
  __shared__ uint16_t re[RD_IMG];    // at a row's start: one past its last word
  __shared__ uint16_t pre[RD_IMG + 1];  // occupied slots before position p
  __shared__ uint16_t crow[RD_IMG];  // row start of every packed value
  __shared__ uint32_t wsum[RD_THREADS / WAVE];
  __shared__ uint32_t h0s, tail_open, long_s;
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  const RowSplit rsp{gbits, jbits};
  const int64_t c0 = (int64_t)blockIdx.x * RD_C;
  const int m = (int)min((int64_t)RD_IMG, n - c0);
  const int mc = min(RD_C, m);
  if (t == 0) {
    h0s = 0xFFFFFFFFu;
    tail_open = 0;
    long_s = 0xFFFFFFFFu;
  }
  __syncthreads();
  // load; a word whose i differs from its predecessor's starts a row (i itself is not kept in LDS:
  // the output step re-reads it, the lines are still in L2)
  {
    uint64_t x[RD_PER], xp[RD_PER];  // all global loads of the workgroup are issued before the first use
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = k * RD_THREADS + t;
      x[k] = p < m ? in[c0 + p] : 0;
    }
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = k * RD_THREADS + t;
      xp[k] = (lane == 0 && p < m && c0 + p > 0) ? in[c0 + p - 1] : 0;
    }
    const bool more = t == 0 && c0 + m < n;   // does the last word's row go on past the image?
    const uint64_t xlast = more ? in[c0 + m - 1] : 0, xnext = more ? in[c0 + m] : 1ull << 32;
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = k * RD_THREADS + t;
      const uint32_t h = rsp.row(x[k]);
      const uint32_t ph = __shfl_up(h, 1, WAVE);
      if (p < m) {
        lo[p] = rsp.val(x[k]);
        tab[p] = RD_EMPTY;
        const bool head = lane == 0 ? (c0 + p == 0 || rsp.row(xp[k]) != h) : ph != h;
        rs[p] = head ? (uint16_t)(p + 1) : (uint16_t)0;
      }
    }
    if (more && rsp.row(xlast) == rsp.row(xnext)) tail_open = 1;
  }
  __syncthreads();

  // row starts: running maximum of (head position + 1), blocked layout (RD_PER consecutive words per thread)
  const int b0 = t * RD_PER;
  {
    uint32_t run = 0;
    uint16_t loc[RD_PER];
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = b0 + k;
      if (p < m) run = max(run, (uint32_t)rs[p]);
      loc[k] = (uint16_t)run;
    }
    uint32_t inc = run;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc = max(inc, o);
    }
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t excl = __shfl_up(inc, 1, WAVE);
    if (lane == 0) excl = 0;
    for (int k = 0; k < w; ++k) excl = max(excl, wsum[k]);
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = b0 + k;
      if (p < m) rs[p] = (uint16_t)max((uint32_t)loc[k], excl);
    }
  }
  __syncthreads();
  for (int p = t; p < m; p += RD_THREADS) {
    const uint32_t s1 = rs[p];
    if (!s1) continue;
    if (s1 == (uint32_t)p + 1u && p < mc) atomicMin(&h0s, (uint32_t)p);
    const bool last = p + 1 == m;
    if (last || rs[p + 1] == (uint16_t)(p + 2)) re[s1 - 1] = (uint16_t)(p + 1);
    // the last owned row runs past the image: it is left to row_unique_long_kernel (it is
    // necessarily the LAST row that starts in this chunk, so its output follows this workgroup's)
    if (last && (int)s1 - 1 < mc && tail_open) long_s = s1 - 1;
  }
  __syncthreads();
  const int own_end = (int)min((uint32_t)mc, long_s);  // rows starting before this position are finished here

  // 1. hash-set insert of every owned word into its row's span of tab
  for (int p = t; p < m; p += RD_THREADS) {
    const uint32_t s1 = rs[p];
    if (!s1 || (int)s1 - 1 >= own_end) continue;
    const uint32_t s = s1 - 1, e = re[s], len = e - s, v = lo[p];
    uint32_t slot = s + __umulhi(v * 0x9E3779B1u, len);
    for (;;) {  // at most len probes: the row has len slots and at most len distinct values
      const uint32_t old = atomicCAS(&tab[slot], RD_EMPTY, v);
      if (old == RD_EMPTY || old == v) break;
      slot = slot + 1 == e ? s : slot + 1;
    }
  }
  __syncthreads();
  // 2. exclusive prefix sum over the occupied slots; pack the distinct values (lo is free now)
  uint32_t total;
  {
    uint32_t sum = 0, val[RD_PER];
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = b0 + k;
      val[k] = p < m ? tab[p] : RD_EMPTY;
      sum += val[k] != RD_EMPTY;
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc += o;
    }
    __syncthreads();  // every wave is done with wsum (row starts) and with lo
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    total = 0;
    for (int k = 0; k < RD_THREADS / WAVE; ++k) {
      if (k < w) run += wsum[k];
      total += wsum[k];
    }
#pragma unroll
    for (int k = 0; k < RD_PER; ++k) {
      const int p = b0 + k;
      if (p <= m) pre[p] = (uint16_t)run;   // p == m: the grand total (one thread reaches it)
      if (val[k] != RD_EMPTY) {
        lo[run] = val[k];
        crow[run] = (uint16_t)(rs[p] - 1);
        ++run;
      }
    }
    if (t == RD_THREADS - 1) pre[m] = (uint16_t)total;
  }
  __syncthreads();
  // 3. place of every distinct value inside its packed row; write out
  const uint32_t h0 = h0s == 0xFFFFFFFFu ? 0u : h0s;
  uint64_t *dst = tmp + c0 + h0;
  for (uint32_t k = t; k < total; k += RD_THREADS) {
    const uint32_t s = crow[k], cs = pre[s], ce = pre[re[s]], v = lo[k];
    uint32_t r = 0;
    for (uint32_t q = cs; q < ce; ++q) r += lo[q] < v;
    dst[cs + r] = rsp.word(in[c0 + s], v);
  }
  if (t == 0) {
    // two output segments per workgroup: its finished rows, then its long row (filled in later)
    counts[2 * (size_t)blockIdx.x] = total;
    starts[2 * (size_t)blockIdx.x] = (uint64_t)(c0 + h0);
    counts[2 * (size_t)blockIdx.x + 1] = 0;
    starts[2 * (size_t)blockIdx.x + 1] = (uint64_t)(c0 + (long_s == 0xFFFFFFFFu ? 0u : long_s));
    if (long_s != 0xFFFFFFFFu)
      longlist[__hip_atomic_fetch_add(nlong, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = blockIdx.x;
  }
}

// Rows too long for the chunk image (an i with thousands of emitted pairs; rare): one 1024-thread
// workgroup per row, same three steps with a table of RL_CAP slots.  The grid is fixed and walks the
// list the main kernel left, so nothing is read back to size the launch.  A row above RL_CAP words
// raises the overflow word (general path).
constexpr int RL_THREADS = 1024;
constexpr int RL_CAP = 12288;
constexpr int RL_PER = RL_CAP / RL_THREADS;
constexpr int RL_GRID = 512;

__global__ __launch_bounds__(RL_THREADS) void row_unique_long_kernel(const uint64_t *__restrict__ in, int64_t n,
                                                                     uint64_t *__restrict__ tmp,
                                                                     uint64_t *__restrict__ counts,
                                                                     const uint64_t *__restrict__ starts,
                                                                     const uint64_t *__restrict__ longlist,
                                                                     const unsigned long long *__restrict__ nlong,
                                                                     uint64_t *__restrict__ overflow, int gbits,
                                                                     int jbits) {
  __shared__ uint32_t tab[RL_CAP];
  __shared__ uint32_t pk[RL_CAP];
  __shared__ uint32_t wsum[RL_THREADS / WAVE];
  __shared__ long long s_end;
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  const unsigned long long nl = *nlong;
  const RowSplit rsp{gbits, jbits};
  for (unsigned long long e = blockIdx.x; e < nl; e += gridDim.x) {
    const uint64_t b = longlist[e];
    const int64_t s0 = (int64_t)starts[2 * b + 1];
    const uint64_t x0 = in[s0];
    if (t == 0) {  // end of the row: first position whose row id is larger (the words are ordered by it)
      const uint32_t r0 = rsp.row(x0);
      int64_t a = s0 + 1, z = n;
      while (a < z) {
        const int64_t mid = (a + z) >> 1;
        if (rsp.row(in[mid]) > r0) z = mid;
        else a = mid + 1;
      }
      s_end = a;
    }
#pragma unroll
    for (int k = 0; k < RL_PER; ++k) tab[t + k * RL_THREADS] = RD_EMPTY;
    __syncthreads();
    const int64_t len = s_end - s0;
    if (len > RL_CAP) {  // uniform
      if (t == 0) atomicOr((unsigned long long *)overflow, 1ull);
      __syncthreads();
      continue;
    }
    for (int64_t p = t; p < len; p += RL_THREADS) {
      const uint32_t v = rsp.val(in[s0 + p]);
      uint32_t slot = __umulhi(v * 0x9E3779B1u, (uint32_t)RL_CAP);
      for (;;) {
        const uint32_t old = atomicCAS(&tab[slot], RD_EMPTY, v);
        if (old == RD_EMPTY || old == v) break;
        slot = slot + 1 == (uint32_t)RL_CAP ? 0u : slot + 1;
      }
    }
    __syncthreads();
    // pack the distinct values
    const int b0 = t * RL_PER;
    uint32_t val[RL_PER], sum = 0;
#pragma unroll
    for (int k = 0; k < RL_PER; ++k) {
      val[k] = tab[b0 + k];
      sum += val[k] != RD_EMPTY;
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc += o;
    }
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - sum, u = 0;
    for (int k = 0; k < RL_THREADS / WAVE; ++k) {
      if (k < w) run += wsum[k];
      u += wsum[k];
    }
#pragma unroll
    for (int k = 0; k < RL_PER; ++k)
      if (val[k] != RD_EMPTY) pk[run++] = val[k];
    __syncthreads();
    for (uint32_t k = t; k < u; k += RL_THREADS) {
      const uint32_t v = pk[k];
      uint32_t r = 0;
#pragma unroll 8
      for (uint32_t q = 0; q < u; ++q) r += pk[q] < v;
      tmp[s0 + r] = rsp.word(x0, v);
    }
    if (t == 0) counts[2 * b + 1] = u;
    __syncthreads();  // tab / pk / wsum / s_end are reused by the next row
  }
}

// close the gaps: workgroup g copies its counts[g] kept words from tmp[starts[g]..] to out[offs[g]..]
__global__ __launch_bounds__(RD_THREADS) void row_unique_gather_kernel(const uint64_t *__restrict__ tmp,
                                                                       const uint64_t *__restrict__ offs,
                                                                       const uint64_t *__restrict__ starts,
                                                                       uint64_t *__restrict__ out) {
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {  // the workgroup's finished rows, then its long row (usually empty)
    const size_t g = 2 * (size_t)blockIdx.x + seg;
    const uint64_t o0 = offs[g], cnt = offs[g + 1] - o0;
    const uint64_t *src = tmp + starts[g];
    for (uint32_t k = threadIdx.x; k < cnt; k += RD_THREADS) out[o0 + k] = src[k];
  }
}

// workspace: counts[2 nblk + 1] | starts[2 nblk] | longlist[nblk] | nlong | chunk totals of the scan
QRLSH_EXPORT size_t qrlsh_row_unique_workspace_bytes(int64_t n) {
  const int64_t nblk = n > 0 ? ceil_div64(n, RD_C) : 0;
  return (size_t)(5 * nblk + 2 + ceil_div64(2 * nblk + 1, SCANL_CHUNK) + 1) * sizeof(uint64_t);
}

QRLSH_EXPORT int qrlsh_row_unique_count(const uint64_t *grouped, int64_t n, int32_t group_bits, int32_t id_bits,
                                        uint64_t *tmp, void *workspace, size_t workspace_bytes,
                                        uint64_t *total_overflow_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && total_overflow_out, "qrlsh_row_unique_count: bad arguments");
  QR_CHECK_ARG(group_bits >= 0 && group_bits <= 8 && id_bits >= 1 && id_bits <= 32 &&
                   (group_bits == 0 || group_bits + id_bits <= 32),
               "qrlsh_row_unique_count: group_bits=%d / id_bits=%d (need group_bits + id_bits <= 32)", group_bits,
               id_bits);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(total_overflow_out, 0, 2 * sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_row_unique_count: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(grouped && tmp && workspace, "qrlsh_row_unique_count: null pointer");
  if (workspace_bytes < qrlsh_row_unique_workspace_bytes(n)) {
    qrlsh_set_error("qrlsh_row_unique_count: workspace %zu < %zu bytes", workspace_bytes,
                    qrlsh_row_unique_workspace_bytes(n));
    return QRLSH_EWORKSPACE;
  }
  const int64_t nblk = ceil_div64(n, RD_C);
  uint64_t *counts = static_cast<uint64_t *>(workspace), *starts = counts + (2 * nblk + 1);
  uint64_t *longlist = starts + 2 * nblk, *nlong = longlist + nblk;
  if (hipMemsetAsync(counts + 2 * nblk, 0, sizeof(uint64_t), st) != hipSuccess ||
      hipMemsetAsync(nlong, 0, sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_row_unique_count: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  QR_LAUNCH("row_unique", row_unique_kernel, dim3((unsigned)nblk), dim3(RD_THREADS), 0, st, grouped, n, tmp, counts,
            starts, longlist, reinterpret_cast<unsigned long long *>(nlong), group_bits, id_bits);
  QR_LAUNCH("row_unique_long", row_unique_long_kernel, dim3((unsigned)(nblk < RL_GRID ? nblk : RL_GRID)),
            dim3(RL_THREADS), 0, st, grouped, n, tmp, counts, (const uint64_t *)starts, (const uint64_t *)longlist,
            (const unsigned long long *)nlong, total_overflow_out + 1, group_bits, id_bits);
  qr_scan_u64(counts, 2 * nblk + 1, total_overflow_out, nlong + 1, st);
  QR_LAUNCH_CHECK("qrlsh_row_unique_count");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_row_unique_fill(const uint64_t *tmp, int64_t n, const void *workspace, uint64_t *out,
                                       void *stream) {
  QR_CHECK_ARG(n >= 0, "qrlsh_row_unique_fill: bad n");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(tmp && workspace && out, "qrlsh_row_unique_fill: null pointer");
  const int64_t nblk = ceil_div64(n, RD_C);
  const uint64_t *offs = static_cast<const uint64_t *>(workspace), *starts = offs + (2 * nblk + 1);
  QR_LAUNCH("row_unique_gather", row_unique_gather_kernel, dim3((unsigned)nblk), dim3(RD_THREADS), 0,
            static_cast<hipStream_t>(stream), tmp, offs, starts, out);
  QR_LAUNCH_CHECK("qrlsh_row_unique_fill");
  return QRLSH_OK;
}

// ---- a3 tail, region form: sorted unique pairs from pairs grouped by i >> g, g up to 8 ---------------
// row_unique above finishes rows that a workgroup discovers inside a fixed chunk of the input; its cost is the
// bookkeeping of that discovery (row starts / ends / overhang, per-position arrays) and, at 2^24 ids, the three
// grouping passes that make single-i rows.  Here a REGION is the set of words whose i share their bits above
// g (2^g consecutive queries, a few thousand words): the grouping sort orders the words by i >> g only -- at
// 2^24 ids and g = 8 that is TWO radix passes -- and one workgroup finishes one region:
//   1. the words are streamed from global memory (never staged) into an open-addressing hash set in LDS keyed
//      by the 32-bit value (i's low g bits, j); a first insertion also counts the value for its i (256 counters);
//   2. the counters are scanned -> where each i's distinct values start in the output;
//   3. the occupied slots are dealt to their i's segment, then every value finds its place by counting the
//      smaller ones of its own i (a handful).
// Only the number of DISTINCT pairs of a region is bounded by LDS, not its word count, so an i with thousands
// of repeated emissions is no special case: about 5 K per region in the main kernel (two workgroups per CU),
// about 11 K in the big-image kernel that takes over the few regions beyond that; a region beyond THAT raises
// the overflow word and the caller takes the general path.  Region boundaries come from a binary search per region (the words are
// ordered by region), outputs are packed by the same count -> scan -> gather as above.
constexpr int RG_THREADS = 1024;
constexpr int RG_SEG = 6144;     // distinct pairs a region may hold
constexpr int RG_ROWS = 256;     // 2^g <= 256
constexpr int RG_LONGROW = 192;  // a query with more distinct neighbours than this is ranked through sub-buckets

__global__ __launch_bounds__(256) void region_bounds_kernel(const uint64_t *__restrict__ w, int64_t n, int shift,
                                                            int64_t nregions, uint64_t *__restrict__ starts) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f > nregions) return;
  int64_t a = 0, b = n;  // first position whose region is >= f
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if ((int64_t)(w[mid] >> shift) >= f) b = mid;
    else a = mid + 1;
  }
  starts[f] = (uint64_t)a;
}

// One region, finished by the calling workgroup (RG_THREADS threads).  TAB_LOG2 / SEG size the hash set and the
// segment array.  Returns false (uniform) when the region holds more than SEG distinct pairs.
template <int TAB_LOG2, int SEG>
__device__ static inline bool region_finish(const uint64_t *__restrict__ in, int64_t s0, int64_t s1, int64_t region,
                                            uint64_t *__restrict__ tmp, uint64_t *__restrict__ counts, int gbits,
                                            int jbits) {
  constexpr int TAB = 1 << TAB_LOG2;
  __shared__ uint32_t tab[TAB];
  __shared__ uint32_t seg[SEG];
  __shared__ uint32_t rowcnt[RG_ROWS], rowstart[RG_ROWS + 1], rowfill[RG_ROWS];
  __shared__ uint32_t sub[RG_ROWS], substart[RG_ROWS + 1], subfill[RG_ROWS];
  __shared__ uint64_t longmask[RG_ROWS / WAVE];
  __shared__ uint32_t wsum[RG_ROWS / WAVE];
  __shared__ uint32_t full, ndist;
  const int t = threadIdx.x, lane = t & (WAVE - 1), wv = t >> 6;
#pragma unroll
  for (int k = 0; k < TAB / RG_THREADS; ++k) tab[t + k * RG_THREADS] = RD_EMPTY;
  if (t < RG_ROWS) {
    rowcnt[t] = 0;
    rowfill[t] = 0;
  }
  if (t == 0) {
    full = 0;
    ndist = 0;
  }
  __syncthreads();
  const uint32_t gmask = (1u << gbits) - 1u, jmask = (1u << jbits) - 1u;  // jbits <= 31 (qrlsh_region_unique_count refuses 32)
  // 1. stream the words into the hash set, four independent loads in flight per thread
  for (int64_t p0 = s0 + t; p0 < s1; p0 += 4 * RG_THREADS) {
    uint64_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t p = p0 + (int64_t)k * RG_THREADS;
      x[k] = p < s1 ? in[p] : ~0ull;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p0 + (int64_t)k * RG_THREADS >= s1) continue;
      const uint32_t row = (uint32_t)(x[k] >> 32) & gmask;
      const uint32_t v = row << jbits | ((uint32_t)x[k] & jmask);
      uint32_t slot = (v * 0x9E3779B1u) >> (32 - TAB_LOG2);
      // the set never takes more than SEG values (SEG < TAB: a free slot always turns up); once it would, the
      // region is given up and the remaining words are skipped
      while (!*(volatile uint32_t *)&full) {
        const uint32_t old = atomicCAS(&tab[slot], RD_EMPTY, v);
        if (old == RD_EMPTY) {
          atomicAdd(&rowcnt[row], 1u);
          if (atomicAdd(&ndist, 1u) >= (uint32_t)SEG - RG_THREADS) full = 1;  // (up to RG_THREADS inserts are in flight)
          break;
        }
        if (old == v) break;
        slot = (slot + 1) & (TAB - 1);
      }
    }
    if (*(volatile uint32_t *)&full) break;
  }
  __syncthreads();
  // 2. where each i's distinct values start: exclusive scan of the 256 counters
  uint32_t c = 0, inc = 0;
  if (t < RG_ROWS) {
    c = rowcnt[t];
    inc = c;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc += o;
    }
    if (lane == WAVE - 1) wsum[wv] = inc;
  }
  __syncthreads();
  if (t < RG_ROWS) {
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < RG_ROWS / WAVE; ++k)
      if (k < wv) base += wsum[k];
    rowstart[t] = base + inc - c;
    if (t == RG_ROWS - 1) rowstart[RG_ROWS] = base + inc;
    const uint64_t lm = __ballot(c > (uint32_t)RG_LONGROW);  // which of this wave's 64 rows are popular queries
    if (lane == 0) longmask[wv] = lm;
  }
  __syncthreads();
  const uint32_t u = rowstart[RG_ROWS];
  const bool fits = !full;   // full: SEG - RG_THREADS distinct values were reached (u <= SEG either way)
  __syncthreads();  // every thread has read `full` / `u` before the arrays are touched again (or re-initialised)
  if (!fits) return false;
  // 3. deal the occupied slots to their i's segment ...
#pragma unroll
  for (int k = 0; k < TAB / RG_THREADS; ++k) {
    const uint32_t v = tab[t + k * RG_THREADS];
    if (v != RD_EMPTY) {
      const uint32_t row = v >> jbits;
      seg[rowstart[row] + atomicAdd(&rowfill[row], 1u)] = v;
    }
  }
  __syncthreads();
  // ... and place every value by the number of smaller ones of its own i
  const uint64_t ihigh = (uint64_t)region << gbits;
  uint64_t *dst = tmp + s0;
  for (uint32_t k = t; k < u; k += RG_THREADS) {
    const uint32_t v = seg[k], row = v >> jbits;
    const uint32_t rs = rowstart[row], re = rowstart[row + 1];
    if (re - rs > (uint32_t)RG_LONGROW) continue;  // a popular query: below
    uint32_t r = 0;
    for (uint32_t q = rs; q < re; ++q) r += seg[q] < v;
    dst[rs + r] = (ihigh | row) << 32 | (v & jmask);
  }
  // A popular query (hundreds to thousands of distinct neighbours) would cost its square that way.  Its values
  // are first dealt into 256 sub-buckets by the top bits of j (the same count -> scan -> deal as above, into the
  // hash table's space, which is dead by now) and then ranked inside their sub-bucket.
  const int sh = jbits > 8 ? jbits - 8 : 0;
  uint32_t *seg2 = tab;
  for (int part = 0; part < RG_ROWS / WAVE; ++part)
  for (uint64_t lm = longmask[part]; lm; lm &= lm - 1) {  // uniform: every thread reads the same masks
    const int row = part * WAVE + __ffsll((long long)lm) - 1;
    const uint32_t rs = rowstart[row], n = rowstart[row + 1] - rs;
    __syncthreads();  // the previous long row (or the short-row loop) is done with sub* / seg2
    if (t < RG_ROWS) {
      sub[t] = 0;
      subfill[t] = 0;
    }
    __syncthreads();
    for (uint32_t k = t; k < n; k += RG_THREADS) atomicAdd(&sub[(seg[rs + k] & jmask) >> sh], 1u);
    __syncthreads();
    uint32_t c2 = 0, inc2 = 0;
    if (t < RG_ROWS) {
      c2 = sub[t];
      inc2 = c2;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(inc2, d, WAVE);
        if (lane >= d) inc2 += o;
      }
      if (lane == WAVE - 1) wsum[wv] = inc2;
    }
    __syncthreads();
    if (t < RG_ROWS) {
      uint32_t base = 0;
#pragma unroll
      for (int k = 0; k < RG_ROWS / WAVE; ++k)
        if (k < wv) base += wsum[k];
      substart[t] = base + inc2 - c2;
      if (t == RG_ROWS - 1) substart[RG_ROWS] = base + inc2;
    }
    __syncthreads();
    for (uint32_t k = t; k < n; k += RG_THREADS) {
      const uint32_t v = seg[rs + k], b2 = (v & jmask) >> sh;
      seg2[substart[b2] + atomicAdd(&subfill[b2], 1u)] = v;
    }
    __syncthreads();
    for (uint32_t k = t; k < n; k += RG_THREADS) {
      const uint32_t v = seg2[k], b2 = (v & jmask) >> sh;
      const uint32_t bs = substart[b2], be = substart[b2 + 1];
      uint32_t r = 0;
      for (uint32_t q = bs; q < be; ++q) r += seg2[q] < v;
      dst[rs + bs + r] = (ihigh | (uint32_t)row) << 32 | (v & jmask);
    }
  }
  if (t == 0) counts[region] = u;
  __syncthreads();  // the big kernel re-uses the arrays for its next region
  return true;
}

__global__ __launch_bounds__(RG_THREADS, 8) void region_unique_kernel(const uint64_t *__restrict__ in,
                                                                      const uint64_t *__restrict__ starts,
                                                                      const uint64_t *__restrict__ ends,
                                                                      uint64_t *__restrict__ tmp,
                                                                      uint64_t *__restrict__ counts,
                                                                      uint64_t *__restrict__ biglist,
                                                                      unsigned long long *__restrict__ nbig, int gbits,
                                                                      int jbits) {
  const int64_t region = blockIdx.x;
  // words of the region: [starts[r], ends[r]) -- ends = starts + 1 for words sorted by region, its own array for the
  // fixed regions of qrlsh_pair_regions_scatter
  const int64_t s0 = (int64_t)starts[region], s1 = (int64_t)ends[region];
  if (s0 == s1) {  // uniform
    if (threadIdx.x == 0) counts[region] = 0;
    return;
  }
  if (!region_finish<13, RG_SEG>(in, s0, s1, region, tmp, counts, gbits, jbits) && threadIdx.x == 0) {
    // more distinct pairs than this image holds (a few very popular queries): left to the big-image kernel
    counts[region] = 0;
    biglist[__hip_atomic_fetch_add(nbig, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = (uint64_t)region;
  }
}

// Regions the kernel above could not hold: the same finish with a 16384-slot set and a 12288-value segment
// (one workgroup per CU), from a fixed grid that walks the device-side list -- nothing is read back to size
// the launch.  A region beyond THAT raises the overflow word (general path).
constexpr int RG_BIG_SEG = 12288;
constexpr int RG_BIG_GRID = 256;
__global__ __launch_bounds__(RG_THREADS, 4) void region_unique_big_kernel(const uint64_t *__restrict__ in,
                                                                          const uint64_t *__restrict__ starts,
                                                                          const uint64_t *__restrict__ ends,
                                                                          uint64_t *__restrict__ tmp,
                                                                          uint64_t *__restrict__ counts,
                                                                          const uint64_t *__restrict__ biglist,
                                                                          const unsigned long long *__restrict__ nbig,
                                                                          uint64_t *__restrict__ overflow, int gbits,
                                                                          int jbits) {
  const unsigned long long nb = *nbig;
  for (unsigned long long e = blockIdx.x; e < nb; e += gridDim.x) {
    const int64_t region = (int64_t)biglist[e];
    const int64_t s0 = (int64_t)starts[region], s1 = (int64_t)ends[region];
    if (!region_finish<14, RG_BIG_SEG>(in, s0, s1, region, tmp, counts, gbits, jbits) && threadIdx.x == 0)
      atomicOr((unsigned long long *)overflow, 1ull);
  }
}

// close the gaps: workgroup r copies its counts[r] words from tmp[starts[r] ..) to out[offs[r] ..)
__global__ __launch_bounds__(256) void region_gather_kernel(const uint64_t *__restrict__ tmp,
                                                            const uint64_t *__restrict__ offs,
                                                            const uint64_t *__restrict__ starts,
                                                            uint64_t *__restrict__ out) {
  const size_t r = blockIdx.x;
  const uint64_t o0 = offs[r], cnt = offs[r + 1] - o0;
  const uint64_t *src = tmp + starts[r];
  for (uint32_t k = threadIdx.x; k < cnt; k += 256) out[o0 + k] = src[k];
}

// workspace: starts[nregions + 1] | counts[nregions + 1] | biglist[nregions] | nbig | chunk totals of the scan |
//            ends[nregions + 1] (fixed-region form only)
static int64_t region_count(int64_t nids, int gbits) { return (nids + (1ll << gbits) - 1) >> gbits; }

QRLSH_EXPORT size_t qrlsh_region_unique_workspace_bytes(int64_t nids, int32_t group_bits) {
  if (nids <= 0 || group_bits < 0 || group_bits > 8) return 64;
  const int64_t nr = region_count(nids, group_bits);
  return (size_t)(4 * (nr + 1) + ceil_div64(nr + 1, SCANL_CHUNK) + 2) * sizeof(uint64_t);
}

// spans of the fixed regions qrlsh_pair_regions_scatter fills: region r = words [r * cap, r * cap + counts[r])
__global__ __launch_bounds__(256) void region_spans_kernel(const uint32_t *__restrict__ counts, int64_t nr, uint32_t cap,
                                                           uint64_t *__restrict__ starts, uint64_t *__restrict__ ends) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nr) return;
  starts[r] = (uint64_t)r * cap;
  ends[r] = (uint64_t)r * cap + (r < nr ? min(counts[r], cap) : 0u);
}

// shared body of the two count entry points.  regions_counts == NULL: `grouped` holds n words sorted by region (bounds by
// binary search); else the fixed regions of qrlsh_pair_regions_scatter (region r at r * cap, regions_counts[r] words).
static int region_unique_count_impl(const char *name, const uint64_t *grouped, int64_t n, const uint32_t *regions_counts,
                                    uint32_t cap, int32_t group_bits, int32_t id_bits, int64_t nids, uint64_t *tmp,
                                    void *workspace, size_t workspace_bytes, uint64_t *total_overflow_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && total_overflow_out && nids > 0 && nids <= (1ll << 32), "%s: bad arguments", name);
  // the 32-bit value (i's low bits, j) must never be the empty-slot marker 0xFFFFFFFF: either it has a spare
  // bit, or the largest j (nids - 1) is not all ones
  // (id_bits <= 31: the kernels build the j mask as (1u << id_bits) - 1)
  QR_CHECK_ARG(group_bits >= 0 && group_bits <= 8 && id_bits >= 1 && id_bits <= 31 && nids <= (1ll << id_bits) &&
                   (group_bits + id_bits < 32 || (group_bits + id_bits == 32 && nids < (1ll << id_bits))),
               "%s: group_bits=%d / id_bits=%d (need group_bits <= 8, id_bits <= 31, group_bits + id_bits <= 32)", name,
               group_bits, id_bits);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(total_overflow_out, 0, 2 * sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("%s: hipMemsetAsync failed", name);
    return QRLSH_EHIP;
  }
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(grouped && tmp && workspace, "%s: null pointer", name);
  if (workspace_bytes < qrlsh_region_unique_workspace_bytes(nids, group_bits)) {
    qrlsh_set_error("%s: workspace %zu < %zu bytes", name, workspace_bytes,
                    qrlsh_region_unique_workspace_bytes(nids, group_bits));
    return QRLSH_EWORKSPACE;
  }
  const int64_t nr = region_count(nids, group_bits);
  QR_CHECK_ARG(nr <= 2147483647ll, "%s: too many regions", name);
  uint64_t *starts = static_cast<uint64_t *>(workspace), *counts = starts + (nr + 1), *biglist = counts + (nr + 1);
  uint64_t *nbig = biglist + nr, *sums = nbig + 1;
  uint64_t *ends = sums + ceil_div64(nr + 1, SCANL_CHUNK) + 1;
  if (hipMemsetAsync(counts + nr, 0, sizeof(uint64_t), st) != hipSuccess ||
      hipMemsetAsync(nbig, 0, sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("%s: hipMemsetAsync failed", name);
    return QRLSH_EHIP;
  }
  if (regions_counts) {
    QR_LAUNCH("region_bounds", region_spans_kernel, dim3((unsigned)ceil_div64(nr + 1, 256)), dim3(256), 0, st, regions_counts,
              nr, cap, starts, ends);
  } else {
    QR_LAUNCH("region_bounds", region_bounds_kernel, dim3((unsigned)ceil_div64(nr + 1, 256)), dim3(256), 0, st, grouped, n,
              32 + group_bits, nr, starts);
    ends = starts + 1;
  }
  QR_LAUNCH("region_unique", region_unique_kernel, dim3((unsigned)nr), dim3(RG_THREADS), 0, st, grouped,
            (const uint64_t *)starts, (const uint64_t *)ends, tmp, counts, biglist,
            reinterpret_cast<unsigned long long *>(nbig), group_bits, id_bits);
  QR_LAUNCH("region_unique_big", region_unique_big_kernel, dim3((unsigned)(nr < RG_BIG_GRID ? nr : RG_BIG_GRID)),
            dim3(RG_THREADS), 0, st, grouped, (const uint64_t *)starts, (const uint64_t *)ends, tmp, counts,
            (const uint64_t *)biglist, (const unsigned long long *)nbig, total_overflow_out + 1, group_bits, id_bits);
  qr_scan_u64(counts, nr + 1, total_overflow_out, sums, st);
  QR_LAUNCH_CHECK(name);
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_region_unique_count(const uint64_t *grouped, int64_t n, int32_t group_bits, int32_t id_bits,
                                           int64_t nids, uint64_t *tmp, void *workspace, size_t workspace_bytes,
                                           uint64_t *total_overflow_out, void *stream) {
  return region_unique_count_impl("qrlsh_region_unique_count", grouped, n, nullptr, 0u, group_bits, id_bits, nids, tmp,
                                  workspace, workspace_bytes, total_overflow_out, stream);
}

// The same from the fixed regions qrlsh_pair_regions_scatter filled (region r = regions[r * cap ..), counts[r] words, any
// order): tmp must hold as many words as the region buffer; n = the number of words scattered (0: nothing to do).
QRLSH_EXPORT int qrlsh_region_unique_count_regions(const uint64_t *regions, const uint32_t *counts, int64_t cap, int64_t n,
                                                   int32_t group_bits, int32_t id_bits, int64_t nids, uint64_t *tmp,
                                                   void *workspace, size_t workspace_bytes, uint64_t *total_overflow_out,
                                                   void *stream) {
  QR_CHECK_ARG(counts && cap > 0 && cap < (1ll << 32), "qrlsh_region_unique_count_regions: bad arguments");
  return region_unique_count_impl("qrlsh_region_unique_count_regions", regions, n, counts, (uint32_t)cap, group_bits, id_bits,
                                  nids, tmp, workspace, workspace_bytes, total_overflow_out, stream);
}

QRLSH_EXPORT int qrlsh_region_unique_fill(const uint64_t *tmp, int64_t n, int32_t group_bits, int64_t nids,
                                          const void *workspace, uint64_t *out, void *stream) {
  QR_CHECK_ARG(n >= 0 && nids > 0 && group_bits >= 0 && group_bits <= 8, "qrlsh_region_unique_fill: bad arguments");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(tmp && workspace && out, "qrlsh_region_unique_fill: null pointer");
  const int64_t nr = region_count(nids, group_bits);
  const uint64_t *starts = static_cast<const uint64_t *>(workspace), *offs = starts + (nr + 1);
  QR_LAUNCH("region_gather", region_gather_kernel, dim3((unsigned)nr), dim3(256), 0, static_cast<hipStream_t>(stream), tmp,
            offs, starts, out);
  QR_LAUNCH_CHECK("qrlsh_region_unique_fill");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_topk_count(const uint64_t *sorted_edges, int64_t n_edges, int32_t K, int32_t id_bits,
                                  void *workspace, size_t workspace_bytes, uint64_t *total_out, void *stream) {
  // id_bits == 0 selects the wide-id edge format (src << 11 | inv, dst as payload)
  QR_CHECK_ARG(K > 0 && id_bits >= 0 && id_bits <= 32, "qrlsh_topk_count: bad K=%d or id_bits=%d (need <= 32)", K,
               id_bits);
  return compact_count<PRED_TOPK>(sorted_edges, n_edges, K, id_bits + 11, workspace, workspace_bytes, total_out, stream,
                                  "qrlsh_topk_count");
}

QRLSH_EXPORT int qrlsh_topk_fill(const uint64_t *sorted_edges, const uint32_t *sorted_dst, int64_t n_edges, int32_t K,
                                 int32_t id_bits, const void *workspace, int32_t *src_out, int32_t *dst_out,
                                 int32_t *milli_out, void *stream) {
  return qrlsh_topk_fill_based(sorted_edges, sorted_dst, n_edges, K, id_bits, 0, workspace, src_out, dst_out, milli_out,
                               stream);
}

QRLSH_EXPORT int qrlsh_topk_fill_based(const uint64_t *sorted_edges, const uint32_t *sorted_dst, int64_t n_edges,
                                       int32_t K, int32_t id_bits, int64_t src_base, const void *workspace,
                                       int32_t *src_out, int32_t *dst_out, int32_t *milli_out, void *stream) {
  QR_CHECK_ARG(K > 0 && id_bits >= 0 && id_bits <= 32 && n_edges >= 0 && src_base >= 0 && src_base < (1ll << 31),
               "qrlsh_topk_fill: bad arguments");
  QR_CHECK_ARG((id_bits == 0) == (sorted_dst != nullptr), "qrlsh_topk_fill: sorted_dst goes with id_bits == 0");
  if (n_edges == 0) return QRLSH_OK;
  QR_CHECK_ARG(sorted_edges && workspace && src_out && dst_out && milli_out, "qrlsh_topk_fill: null pointer");
  QR_LAUNCH("topk_fill", (compact_fill_kernel<PRED_TOPK>), dim3((unsigned)ceil_div64(n_edges, CMP_TILE)),
                     dim3(CMP_THREADS), 0, static_cast<hipStream_t>(stream), sorted_edges, n_edges, K, id_bits + 11,
                     id_bits, sorted_dst, static_cast<const uint64_t *>(workspace), nullptr, src_out, dst_out, milli_out,
                     (int32_t)src_base);
  QR_LAUNCH_CHECK("qrlsh_topk_fill");
  return QRLSH_OK;
}

// ---- a5 tail, select form: per-query top-K without sorting the directed edges ---------------------------
// The sort form above orders all 2n directed edge keys on (src, 1000 - milli): ceil((id_bits + 11) / 8) radix
// passes over 2n words.  But the forward edges (src = i) ARE the scored pair list, already grouped by src and
// ordered by dst; only the n reverse edges (src = j) have to be brought together, and for that a stable sort on
// j's bits alone is enough (ceil(id_bits / 8) passes over n words -- under a third of the key-passes).  A query's
// neighbours are then two runs, [fstart[q], fstart[q+1]) of the pairs and [rstart[q], rstart[q+1]) of the sorted
// reverse words, and every directed edge finds its rank in its query's list by counting the edges of those two
// runs that order before it (value descending, then neighbour id ascending) -- stopping as soon as K of them
// have been seen.  Edge of rank r < K goes to out[off[q] + r], off = exclusive scan of min(K, list length): the
// output is the same (src, value desc, dst asc) COO the sort form writes, bit for bit.  Three kernels by list
// length: up to 16 neighbours (almost every query) a 16-lane group per query ranks by rotating the keys round
// its DPP row; 17 .. 64 a wave per query; longer lists a wave per query with a histogram of the 2001 possible
// values (O(length), see below).
// Reverse words: packed  j << (id_bits + 11) | inv << id_bits | i  (rdst == NULL), or key + payload
// (j << 11 | inv, i) for ids that do not fit.
__device__ static inline uint32_t rev_src(uint64_t w, int id_bits, bool wide) {
  return (uint32_t)(wide ? w >> 11 : w >> (id_bits + 11));
}

// start[q] = first position of `a` whose src is >= q (q = 0 .. nq); a is ordered by src.  blockIdx.y = 0: a = the
// pairs, src = i -> fstart; 1: a = the sorted reverse words, src = j -> rstart.  The thread at a change of src
// fills the (usually 1 - 2) entries up to its src; a long stretch of queries without any edge (the ids beyond
// the last i, below the first j, ...) is left at SEL_UNSET for edge_bounds_fix_kernel, whose threads find their
// entry by binary search -- one thread walking a million-entry gap was the whole cost of this step.
constexpr uint32_t SEL_UNSET = 0xFFFFFFFFu;  // n < 2^31: never a position
constexpr int SEL_GAP = 32;
__device__ static inline int64_t edge_src(const uint64_t *__restrict__ a, int64_t t, bool fwd, int id_bits, bool wide) {
  return (int64_t)(fwd ? (uint32_t)(a[t] >> 32) : rev_src(a[t], id_bits, wide));
}
__global__ __launch_bounds__(256) void edge_bounds_kernel(const uint64_t *__restrict__ pairs,
                                                          const uint64_t *__restrict__ rev, int64_t n, int64_t nq,
                                                          int id_bits, int wide, uint32_t *__restrict__ fstart,
                                                          uint32_t *__restrict__ rstart, int y0 = 0) {
  const bool fwd = blockIdx.y + y0 == 0;  // (y0 = 1: reverse words only, no forward list)
  const uint64_t *a = fwd ? pairs : rev;
  uint32_t *start = fwd ? fstart : rstart;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n) return;
  const int64_t s = t < n ? edge_src(a, t, fwd, id_bits, wide != 0) : nq;
  const int64_t p = t > 0 ? edge_src(a, t - 1, fwd, id_bits, wide != 0) : -1;
  if (s - p > SEL_GAP) {
    if (s <= nq) start[s] = (uint32_t)t;  // the entry of s itself; the stretch below it stays unset
    return;
  }
  for (int64_t q = p + 1; q <= s && q <= nq; ++q) start[q] = (uint32_t)t;
}

__global__ __launch_bounds__(256) void edge_bounds_fix_kernel(const uint64_t *__restrict__ pairs,
                                                              const uint64_t *__restrict__ rev, int64_t n, int64_t nq,
                                                              int id_bits, int wide, uint32_t *__restrict__ fstart,
                                                              uint32_t *__restrict__ rstart, int y0 = 0) {
  const bool fwd = blockIdx.y + y0 == 0;
  const uint64_t *a = fwd ? pairs : rev;
  uint32_t *start = fwd ? fstart : rstart;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q > nq || start[q] != SEL_UNSET) return;
  int64_t lo = 0, hi = n;  // first position whose src is >= q
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (edge_src(a, mid, fwd, id_bits, wide != 0) >= q) hi = mid;
    else lo = mid + 1;
  }
  start[q] = (uint32_t)lo;
}

constexpr int SEL_SHORT = 16;   // lists up to here: one 16-lane group per query (every query is visited)
constexpr int SEL_LONG = 64;    // lists up to here: one wave per query; beyond: the histogram kernel
constexpr int SEL_MAXK = 256;   // largest K of the select form (the sort form has no limit)
constexpr int SEL_LIST_GRID = 1024;

// list lengths -> output counts (min(K, length)); queries whose list does not fit a 16-lane group are put on
// the medium (17 .. 64) or the long list.  A workgroup classifies LEN_QPB consecutive queries, collects its two
// lists in LDS and reserves their room with ONE global atomic each: a popular counter word takes ~90 atomics per
// microsecond, and at 10 M queries nearly every wave holds a medium query.
constexpr int LEN_QPB = 4096;
__global__ __launch_bounds__(256) void topk_len_kernel(const uint32_t *__restrict__ fstart,
                                                       const uint32_t *__restrict__ rstart, int64_t nq, int K,
                                                       uint64_t *__restrict__ cnt, uint32_t *__restrict__ medlist,
                                                       uint32_t *__restrict__ longlist,
                                                       unsigned long long *__restrict__ nlists) {
  __shared__ uint32_t smed[LEN_QPB], slng[LEN_QPB];
  __shared__ uint32_t nmed, nlng;
  __shared__ unsigned long long bmed, blng;
  const int lane = threadIdx.x & (WAVE - 1);
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  if (threadIdx.x == 0) {
    nmed = 0;
    nlng = 0;
  }
  __syncthreads();
  const int64_t q0 = (int64_t)blockIdx.x * LEN_QPB;
#pragma unroll 4
  for (int it = 0; it < LEN_QPB / 256; ++it) {
    const int64_t q = q0 + it * 256 + threadIdx.x;
    uint64_t c = 0;
    if (q < nq) c = (uint64_t)(fstart[q + 1] - fstart[q]) + (rstart[q + 1] - rstart[q]);
    if (q <= nq) cnt[q] = c > (uint64_t)K ? (uint64_t)K : c;  // one word past the end (0): the scan leaves the total there
    const bool med = c > (uint64_t)SEL_SHORT && c <= (uint64_t)SEL_LONG, lng = c > (uint64_t)SEL_LONG;
    const uint64_t mm = __ballot(med), ml = __ballot(lng);
    uint32_t pm = 0, pl = 0;
    if (lane == 0) {
      if (mm) pm = atomicAdd(&nmed, (uint32_t)__popcll(mm));
      if (ml) pl = atomicAdd(&nlng, (uint32_t)__popcll(ml));
    }
    pm = __shfl(pm, 0, WAVE);
    pl = __shfl(pl, 0, WAVE);
    if (med) smed[pm + (uint32_t)__popcll(mm & lt_mask)] = (uint32_t)q;
    if (lng) slng[pl + (uint32_t)__popcll(ml & lt_mask)] = (uint32_t)q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bmed = nmed ? __hip_atomic_fetch_add(&nlists[0], (unsigned long long)nmed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    blng = nlng ? __hip_atomic_fetch_add(&nlists[1], (unsigned long long)nlng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nmed; k += 256) medlist[bmed + k] = smed[k];
  for (uint32_t k = threadIdx.x; k < nlng; k += 256) longlist[blng + k] = slng[k];
}

// (inv << 32 | dst) of element x of a query's list: x < nr -> reverse run, else forward run
__device__ static inline uint64_t sel_key(uint32_t x, uint32_t rs, uint32_t nr, uint32_t fs,
                                          const uint64_t *__restrict__ pairs, const int32_t *__restrict__ milli,
                                          const uint64_t *__restrict__ rev, const uint32_t *__restrict__ rdst,
                                          int id_bits, uint64_t idm) {
  if (x < nr) {
    const uint64_t w = rev[rs + x];
    return rdst ? (w & 0x7FFull) << 32 | rdst[rs + x] : ((w >> id_bits) & 0x7FFull) << 32 | (w & idm);
  }
  const uint32_t y = fs + (x - nr);
  return (uint64_t)(uint32_t)(1000 - milli[y]) << 32 | (uint32_t)pairs[y];
}

// 64-bit value of the lane S positions further round this lane's 16-lane row (DPP row_ror)
template <int S> __device__ static inline uint64_t row_ror64(uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x120 + S, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x120 + S, 0xF, 0xF, false);
  return (uint64_t)(uint32_t)hi << 32 | (uint32_t)lo;
}
template <int S> __device__ static inline uint32_t row_rank(uint64_t mine) {
  uint32_t r = row_ror64<S>(mine) < mine;
  if constexpr (S > 1) r += row_rank<S - 1>(mine);
  return r;
}

// Lists of up to 16 neighbours (almost every query): one 16-lane group per query, lane l holds element l, and a
// lane's rank is the number of smaller keys met while the row rotates past it (15 DPP steps, no memory traffic).
// Absent elements carry the key ~0: never smaller than a real one.
constexpr int SEL_QPG = 4;  // queries per 16-lane group: their loads are issued together (latency-bound otherwise)
__global__ __launch_bounds__(256) void topk_select_short_kernel(const uint64_t *__restrict__ pairs,
                                                                const int32_t *__restrict__ milli,
                                                                const uint64_t *__restrict__ rev,
                                                                const uint32_t *__restrict__ rdst,
                                                                const uint32_t *__restrict__ fstart,
                                                                const uint32_t *__restrict__ rstart,
                                                                const uint64_t *__restrict__ off, int64_t nq, int K,
                                                                int id_bits, int32_t *__restrict__ src_out,
                                                                int32_t *__restrict__ dst_out,
                                                                int32_t *__restrict__ milli_out) {
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) >> 4;
  const uint32_t l = threadIdx.x & 15;
  const uint64_t idm = id_bits >= 32 ? 0xFFFFFFFFull : (1ull << id_bits) - 1ull;
  // query c of this group: group + c * ngroups (consecutive groups -> consecutive queries: the start arrays are read in runs)
  uint32_t fs[SEL_QPG], rs[SEL_QPG], nr[SEL_QPG], len[SEL_QPG];
  uint64_t o0[SEL_QPG], mine[SEL_QPG];
#pragma unroll
  for (int c = 0; c < SEL_QPG; ++c) {
    const int64_t q = group + (int64_t)c * ngroups;
    uint32_t nf = 0;
    fs[c] = rs[c] = nr[c] = 0;
    o0[c] = 0;
    if (q < nq) {
      fs[c] = fstart[q];
      nf = fstart[q + 1] - fs[c];
      rs[c] = rstart[q];
      nr[c] = rstart[q + 1] - rs[c];
      o0[c] = off[q];
    }
    len[c] = nf + nr[c];
    if (len[c] > (uint32_t)SEL_SHORT) len[c] = 0;  // another kernel's query
  }
#pragma unroll
  for (int c = 0; c < SEL_QPG; ++c)
    mine[c] = l < len[c] ? sel_key(l, rs[c], nr[c], fs[c], pairs, milli, rev, rdst, id_bits, idm) : ~0ull;
#pragma unroll
  for (int c = 0; c < SEL_QPG; ++c) {
    const uint32_t rank = row_rank<15>(mine[c]);  // executed by every lane (all lanes of the wave are active here)
    if (l < len[c] && rank < (uint32_t)K) {
      const uint64_t o = o0[c] + rank;
      src_out[o] = (int32_t)(group + (int64_t)c * ngroups);
      dst_out[o] = (int32_t)(uint32_t)mine[c];
      milli_out[o] = 1000 - (int32_t)(uint32_t)(mine[c] >> 32);
    }
  }
}

// Lists of 17 .. 64 neighbours: one wave per query, from a fixed grid that walks the medium list.
__global__ __launch_bounds__(256) void topk_select_medium_kernel(const uint64_t *__restrict__ pairs,
                                                                 const int32_t *__restrict__ milli,
                                                                 const uint64_t *__restrict__ rev,
                                                                 const uint32_t *__restrict__ rdst,
                                                                 const uint32_t *__restrict__ fstart,
                                                                 const uint32_t *__restrict__ rstart,
                                                                 const uint64_t *__restrict__ off,
                                                                 const uint32_t *__restrict__ medlist,
                                                                 const unsigned long long *__restrict__ nlists, int K,
                                                                 int id_bits, int32_t *__restrict__ src_out,
                                                                 int32_t *__restrict__ dst_out,
                                                                 int32_t *__restrict__ milli_out) {
  const int lane = threadIdx.x & (WAVE - 1);
  const uint64_t idm = id_bits >= 32 ? 0xFFFFFFFFull : (1ull << id_bits) - 1ull;
  const unsigned long long nm = nlists[0], nwaves = (unsigned long long)gridDim.x * (blockDim.x / WAVE);
  for (unsigned long long e = (unsigned long long)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6); e < nm;
       e += nwaves) {
    const uint32_t q = medlist[e];
    const uint32_t fs = fstart[q], nf = fstart[q + 1] - fs, rs = rstart[q], nr = rstart[q + 1] - rs;
    const uint32_t len = nf + nr;  // 17 .. 64
    const uint64_t mine = (uint32_t)lane < len ? sel_key(lane, rs, nr, fs, pairs, milli, rev, rdst, id_bits, idm) : ~0ull;
    uint32_t rank = 0;
#pragma unroll 9
    for (int s = 1; s < WAVE; ++s) {  // every other lane's key once (absent elements: ~0, never smaller)
      const int from = (lane + s) & (WAVE - 1);
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)mine, from, WAVE), hi = (uint32_t)__shfl((int)(uint32_t)(mine >> 32), from, WAVE);
      rank += ((uint64_t)hi << 32 | lo) < mine;
    }
    if ((uint32_t)lane < len && rank < (uint32_t)K) {
      const uint64_t o = off[q] + rank;
      src_out[o] = (int32_t)q;
      dst_out[o] = (int32_t)(uint32_t)mine;
      milli_out[o] = 1000 - (int32_t)(uint32_t)(mine >> 32);
    }
  }
}

// Popular queries (lists beyond SEL_LONG): one wave per query, O(list length).  A histogram of the 2001
// possible values (inv = 1000 - milli) locates the value v* at which the K-th neighbour sits; the neighbours
// with inv < v* are all kept, and of those with inv == v* the first K - (number below) in list order -- the
// list order (reverse run, then forward run) IS ascending neighbour id, the tie-break.  The <= K survivors then
// rank themselves among each other.
__global__ __launch_bounds__(256) void topk_select_long_kernel(const uint64_t *__restrict__ pairs,
                                                               const int32_t *__restrict__ milli,
                                                               const uint64_t *__restrict__ rev,
                                                               const uint32_t *__restrict__ rdst,
                                                               const uint32_t *__restrict__ fstart,
                                                               const uint32_t *__restrict__ rstart,
                                                               const uint64_t *__restrict__ off,
                                                               const uint32_t *__restrict__ longlist,
                                                               const unsigned long long *__restrict__ nlists, int K,
                                                               int id_bits, int32_t *__restrict__ src_out,
                                                               int32_t *__restrict__ dst_out,
                                                               int32_t *__restrict__ milli_out, int by_id) {
  constexpr int NV = 2048;  // inv in [0, 2000]
  __shared__ uint32_t hist_all[4][NV];
  __shared__ uint64_t keep_all[4][SEL_MAXK];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
  uint32_t *hist = hist_all[wv];
  uint64_t *keep = keep_all[wv];
  const uint64_t idm = id_bits >= 32 ? 0xFFFFFFFFull : (1ull << id_bits) - 1ull;
  const unsigned long long nl = nlists[1];
  const unsigned long long nwaves = (unsigned long long)gridDim.x * 4;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (unsigned long long e = (unsigned long long)blockIdx.x * 4 + wv; e < nl; e += nwaves) {
    const uint32_t q = longlist[e];
    const uint32_t fs = fstart[q], nf = fstart[q + 1] - fs, rs = rstart[q], nr = rstart[q + 1] - rs;
    const uint32_t len = nf + nr;
    for (int v = lane; v < NV; v += WAVE) hist[v] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t x = lane; x < len; x += WAVE)
      atomicAdd(&hist[(uint32_t)(sel_key(x, rs, nr, fs, pairs, milli, rev, rdst, id_bits, idm) >> 32)], 1u);
    __builtin_amdgcn_wave_barrier();
    // v* = smallest v with count(inv <= v) >= K; below = count(inv < v*).  Lane l owns values [32 l, 32 l + 32).
    uint32_t mysum = 0;
    for (int v = 0; v < NV / WAVE; ++v) mysum += hist[lane * (NV / WAVE) + v];
    uint32_t inc = mysum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc += o;
    }
    const uint64_t reach = __ballot(inc >= (uint32_t)K);  // len > SEL_LONG >= ... may still be < K: then keep all
    uint32_t vstar = NV, below = 0;
    if (reach) {
      const int owner = __ffsll((long long)reach) - 1;
      uint32_t run = __shfl(inc - mysum, owner, WAVE);   // count below the owner's first value
      uint32_t vs = NV, bl = 0;
      if (lane == owner) {
        for (int v = 0; v < NV / WAVE; ++v) {
          const uint32_t h = hist[lane * (NV / WAVE) + v];
          if (run + h >= (uint32_t)K) {
            vs = lane * (NV / WAVE) + v;
            bl = run;
            break;
          }
          run += h;
        }
      }
      vstar = __shfl(vs, owner, WAVE);
      below = __shfl(bl, owner, WAVE);
    }
    const uint32_t want_eq = reach ? (uint32_t)K - below : 0u;  // ties at v* kept, in list order
    // List order IS ascending neighbour id when the list is the query's two runs (smaller ids in the reverse run,
    // larger ones in the forward run, each ascending).  A list made of reverse words alone (by_id: edges that
    // arrived from several scoring ranks) has no such order: the ties to keep are then the want_eq SMALLEST ids
    // among the elements at v* -- found by a radix select on the id, 11 bits per round (ids are distinct in a list).
    uint32_t id_cut = 0xFFFFFFFFu;  // ties with id <= id_cut are kept
    if (by_id && reach && hist[vstar] > want_eq) {  // uniform
      uint32_t prefix = 0, need = want_eq;          // ids whose top bits equal `prefix` are still undecided
      for (int shift = 22; shift >= 0; shift -= 11) {
        __builtin_amdgcn_wave_barrier();
        for (int v = lane; v < NV; v += WAVE) hist[v] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t x = lane; x < len; x += WAVE) {
          const uint64_t k = sel_key(x, rs, nr, fs, pairs, milli, rev, rdst, id_bits, idm);
          const uint32_t idv = (uint32_t)k;
          if ((uint32_t)(k >> 32) == vstar && (shift == 22 || (idv >> (shift + 11)) == prefix))
            atomicAdd(&hist[(idv >> shift) & (NV - 1)], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t ms = 0;
        for (int v = 0; v < NV / WAVE; ++v) ms += hist[lane * (NV / WAVE) + v];
        uint32_t ic = ms;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
          const uint32_t o = __shfl_up(ic, d, WAVE);
          if (lane >= d) ic += o;
        }
        const int owner = __ffsll((long long)__ballot(ic >= need)) - 1;  // (need <= the number of undecided ids)
        uint32_t run = __shfl(ic - ms, owner, WAVE), dg = 0, bl = 0;
        if (lane == owner) {
          for (int v = 0; v < NV / WAVE; ++v) {
            const uint32_t h = hist[lane * (NV / WAVE) + v];
            if (run + h >= need) {
              dg = lane * (NV / WAVE) + v;
              bl = run;
              break;
            }
            run += h;
          }
        }
        dg = __shfl(dg, owner, WAVE);
        bl = __shfl(bl, owner, WAVE);
        prefix = shift == 22 ? dg : (prefix << 11 | dg);
        need -= bl;                                  // ids below this digit are all kept
      }
      id_cut = prefix;                               // the need-th smallest undecided id itself (need == 1 by now)
    }
    // second sweep, in list order: collect the survivors
    uint32_t nkeep = 0, neq = 0;
    for (uint32_t x0 = 0; x0 < len; x0 += WAVE) {
      const uint32_t x = x0 + lane;
      const uint64_t k = x < len ? sel_key(x, rs, nr, fs, pairs, milli, rev, rdst, id_bits, idm) : ~0ull;
      const uint32_t inv = (uint32_t)(k >> 32);
      const bool lt = x < len && inv < vstar;
      const bool eq = x < len && inv == vstar;
      const uint64_t meq = __ballot(eq);
      const bool take_eq = eq && (id_cut != 0xFFFFFFFFu ? (uint32_t)k <= id_cut
                                                        : neq + (uint32_t)__popcll(meq & lt_mask) < want_eq);
      const uint64_t mk = __ballot(lt || take_eq);
      if (lt || take_eq) keep[nkeep + (uint32_t)__popcll(mk & lt_mask)] = k;
      nkeep += (uint32_t)__popcll(mk);
      neq += (uint32_t)__popcll(meq);
    }
    __builtin_amdgcn_wave_barrier();
    // nkeep == min(K, len); rank the survivors among themselves
    const uint64_t o0 = off[q];
    for (uint32_t a = lane; a < nkeep; a += WAVE) {
      const uint64_t k = keep[a];
      uint32_t r = 0;
      for (uint32_t c = 0; c < nkeep; ++c) r += keep[c] < k;
      src_out[o0 + r] = (int32_t)q;
      dst_out[o0 + r] = (int32_t)(uint32_t)k;
      milli_out[o0 + r] = 1000 - (int32_t)(uint32_t)(k >> 32);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// workspace: fstart u32[nq + 1] | rstart u32[nq + 1] | medlist u32[nq] | longlist u32[nq] | off u64[nq + 2] |
//            list lengths u64[2] | chunk totals of the scan
struct SelWs {
  uint32_t *fstart, *rstart, *medlist, *longlist;
  uint64_t *off, *nlong, *sums;
  size_t bytes;
};
static SelWs sel_ws(void *workspace, int64_t nq) {
  SelWs w;
  char *p = static_cast<char *>(workspace);
  size_t o = 0;
  w.fstart = reinterpret_cast<uint32_t *>(p + o);
  o += ((size_t)(nq + 1) * 4 + 15) & ~(size_t)15;
  w.rstart = reinterpret_cast<uint32_t *>(p + o);
  o += ((size_t)(nq + 1) * 4 + 15) & ~(size_t)15;
  w.medlist = reinterpret_cast<uint32_t *>(p + o);
  o += ((size_t)(nq + 1) * 4 + 15) & ~(size_t)15;
  w.longlist = reinterpret_cast<uint32_t *>(p + o);
  o += ((size_t)(nq + 1) * 4 + 15) & ~(size_t)15;
  w.off = reinterpret_cast<uint64_t *>(p + o);
  o += (size_t)(nq + 2) * 8;
  w.nlong = reinterpret_cast<uint64_t *>(p + o);
  o += 16;
  w.sums = reinterpret_cast<uint64_t *>(p + o);
  o += (size_t)(ceil_div64(nq + 1, SCANL_CHUNK) + 2) * 8;
  w.bytes = o;
  return w;
}

QRLSH_EXPORT size_t qrlsh_topk_select_workspace_bytes(int64_t nq) {
  if (nq <= 0) return 64;
  return sel_ws(nullptr, nq).bytes;
}

QRLSH_EXPORT int qrlsh_topk_select_count(const uint64_t *pairs, int64_t n, const uint64_t *rev_sorted,
                                         const uint32_t *rev_dst, int64_t nq, int32_t K, int32_t id_bits, void *workspace,
                                         size_t workspace_bytes, uint64_t *total_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && n < (1ll << 31) && nq > 0 && nq <= (1ll << 32) && K > 0 && K <= SEL_MAXK && id_bits >= 1 &&
                   id_bits <= 32,
               "qrlsh_topk_select_count: bad arguments (n=%lld nq=%lld K=%d (<= %d) id_bits=%d)", (long long)n,
               (long long)nq, K, SEL_MAXK, id_bits);
  // packed reverse words: src << (id_bits + 11) | inv << id_bits | neighbour must fit 64 bits
  QR_CHECK_ARG(rev_dst || id_bits <= 26 || (!pairs && ((uint64_t)(nq - 1) >> (53 - id_bits)) == 0),
               "qrlsh_topk_select_count: packed reverse words need id_bits <= 26 (or, without a forward list, src < 2^(53 - id_bits))");
  // pairs == NULL: the lists are made of the n reverse words alone (the sharded driver: every directed edge a rank
  // receives is such a word, src = its own query)
  QR_CHECK_ARG(total_out && workspace && (n == 0 || rev_sorted), "qrlsh_topk_select_count: null pointer");
  if (workspace_bytes < qrlsh_topk_select_workspace_bytes(nq)) {
    qrlsh_set_error("qrlsh_topk_select_count: workspace %zu < %zu bytes", workspace_bytes,
                    qrlsh_topk_select_workspace_bytes(nq));
    return QRLSH_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const SelWs w = sel_ws(workspace, nq);
  if (hipMemsetAsync(w.nlong, 0, 2 * sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_topk_select_count: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  const dim3 blk(256);
  // both start arrays (contiguous in the workspace) to "unset", boundaries, then the long stretches
  if (hipMemsetAsync(w.fstart, 0xFF, (size_t)((char *)w.medlist - (char *)w.fstart), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_topk_select_count: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  const int lists = pairs ? 2 : 1, y0 = pairs ? 0 : 1;
  if (!pairs && hipMemsetAsync(w.fstart, 0, (size_t)((char *)w.rstart - (char *)w.fstart), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_topk_select_count: hipMemsetAsync failed");   // every forward run is empty
    return QRLSH_EHIP;
  }
  QR_LAUNCH("topk_bounds", edge_bounds_kernel, dim3((unsigned)ceil_div64(n + 1, 256), lists), blk, 0, st, pairs, rev_sorted, n,
            nq, id_bits, rev_dst ? 1 : 0, w.fstart, w.rstart, y0);
  QR_LAUNCH("topk_bounds", edge_bounds_fix_kernel, dim3((unsigned)ceil_div64(nq + 1, 256), lists), blk, 0, st, pairs,
            rev_sorted, n, nq, id_bits, rev_dst ? 1 : 0, w.fstart, w.rstart, y0);
  QR_LAUNCH("topk_len", topk_len_kernel, dim3((unsigned)ceil_div64(nq + 1, LEN_QPB)), blk, 0, st, (const uint32_t *)w.fstart,
            (const uint32_t *)w.rstart, nq, K, w.off, w.medlist, w.longlist,
            reinterpret_cast<unsigned long long *>(w.nlong));
  qr_scan_u64(w.off, nq + 1, total_out, w.sums, st);
  QR_LAUNCH_CHECK("qrlsh_topk_select_count");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_topk_select_fill(const uint64_t *pairs, const int32_t *milli, int64_t n,
                                        const uint64_t *rev_sorted, const uint32_t *rev_dst, int64_t nq, int32_t K,
                                        int32_t id_bits, const void *workspace, int32_t *src_out, int32_t *dst_out,
                                        int32_t *milli_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && n < (1ll << 31) && nq > 0 && K > 0 && K <= SEL_MAXK && id_bits >= 1 && id_bits <= 32,
               "qrlsh_topk_select_fill: bad arguments");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG((pairs == nullptr) == (milli == nullptr) && rev_sorted && workspace && src_out && dst_out && milli_out,
               "qrlsh_topk_select_fill: null pointer (pairs and milli: both or neither)");
  const SelWs w = sel_ws(const_cast<void *>(workspace), nq);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint32_t *fsp = w.fstart, *rsp = w.rstart;
  const uint64_t *offp = w.off;
  const unsigned long long *nl = reinterpret_cast<const unsigned long long *>(w.nlong);
  QR_LAUNCH("topk_select", topk_select_short_kernel, dim3((unsigned)ceil_div64(nq, 16 * SEL_QPG)), dim3(256), 0, st, pairs, milli,
            rev_sorted, rev_dst, fsp, rsp, offp, nq, K, id_bits, src_out, dst_out, milli_out);
  QR_LAUNCH("topk_select_medium", topk_select_medium_kernel, dim3(SEL_LIST_GRID), dim3(256), 0, st, pairs, milli,
            rev_sorted, rev_dst, fsp, rsp, offp, (const uint32_t *)w.medlist, nl, K, id_bits, src_out, dst_out, milli_out);
  QR_LAUNCH("topk_select_long", topk_select_long_kernel, dim3(SEL_LIST_GRID), dim3(256), 0, st, pairs, milli, rev_sorted,
            rev_dst, fsp, rsp, offp, (const uint32_t *)w.longlist, nl, K, id_bits, src_out, dst_out, milli_out,
            pairs ? 0 : 1);
  QR_LAUNCH_CHECK("qrlsh_topk_select_fill");
  return QRLSH_OK;
}
