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
                                                                   int32_t *__restrict__ milli_out) {
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
        src_out[p] = (int32_t)(v[i] >> sh);
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

QRLSH_EXPORT size_t qrlsh_compact_workspace_bytes(int64_t n) {
  if (n <= 0) return 16;
  return (size_t)ceil_div64(n, CMP_TILE) * sizeof(uint64_t);
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
  QR_LAUNCH("scan_blocks", scan_u64_kernel, dim3(1), dim3(1024), 0, st, blk, nblk, total_out);
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

QRLSH_EXPORT int qrlsh_topk_count(const uint64_t *sorted_edges, int64_t n_edges, int32_t K, int32_t id_bits,
                                  void *workspace, size_t workspace_bytes, uint64_t *total_out, void *stream) {
  // id_bits == 0 selects the wide-id edge format (src << 11 | inv, dst as payload)
  QR_CHECK_ARG(K > 0 && id_bits >= 0 && id_bits <= 26, "qrlsh_topk_count: bad K=%d or id_bits=%d (need <= 26)", K,
               id_bits);
  return compact_count<PRED_TOPK>(sorted_edges, n_edges, K, id_bits + 11, workspace, workspace_bytes, total_out, stream,
                                  "qrlsh_topk_count");
}

QRLSH_EXPORT int qrlsh_topk_fill(const uint64_t *sorted_edges, const uint32_t *sorted_dst, int64_t n_edges, int32_t K,
                                 int32_t id_bits, const void *workspace, int32_t *src_out, int32_t *dst_out,
                                 int32_t *milli_out, void *stream) {
  QR_CHECK_ARG(K > 0 && id_bits >= 0 && id_bits <= 26 && n_edges >= 0, "qrlsh_topk_fill: bad arguments");
  QR_CHECK_ARG((id_bits == 0) == (sorted_dst != nullptr), "qrlsh_topk_fill: sorted_dst goes with id_bits == 0");
  if (n_edges == 0) return QRLSH_OK;
  QR_CHECK_ARG(sorted_edges && workspace && src_out && dst_out && milli_out, "qrlsh_topk_fill: null pointer");
  QR_LAUNCH("topk_fill", (compact_fill_kernel<PRED_TOPK>), dim3((unsigned)ceil_div64(n_edges, CMP_TILE)),
                     dim3(CMP_THREADS), 0, static_cast<hipStream_t>(stream), sorted_edges, n_edges, K, id_bits + 11,
                     id_bits, sorted_dst, static_cast<const uint64_t *>(workspace), nullptr, src_out, dst_out, milli_out);
  QR_LAUNCH_CHECK("qrlsh_topk_fill");
  return QRLSH_OK;
}
