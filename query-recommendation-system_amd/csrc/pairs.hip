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
