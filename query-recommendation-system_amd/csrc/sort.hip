// sort.hip -- batched stable LSD radix sort of uint64 keys (+ optional uint32 payload).
//
// Replaces the reference's dict-of-lists buckets (lsh.py:9-15, 31-38), its Python set
// de-duplication (lsh.py:41, 53) and its per-query argsort (recommender.py:206) with one
// primitive.  8 bits per pass; per pass: tile histogram -> exclusive scan -> stable scatter.
// Stability inside a tile comes from wavefront ballots: each lane learns which lanes of
// its wave hold the same digit (8 ballots), ranks itself with a popcount below its lane,
// and the wave keeps running per-digit counters in LDS; the four waves of a workgroup are
// then chained by a 256-entry prefix.
//
// With QRLSH_SORT_MIX the digits come from mix64(key) (a bijection), so after 32 bits
// (4 passes instead of 8) equal keys are adjacent up to 32-bit mix collisions, which the
// pair-emission kernel resolves with a full-key compare.
#include "common.h"

constexpr int SORT_THREADS = 256;
constexpr int SORT_IPT = 16;                          // items per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_IPT;    // 4096 keys per workgroup
constexpr int RADIX = 256;

template <bool MIX> __device__ static inline uint32_t digit_of(uint64_t key, int shift) {
  const uint64_t x = MIX ? qr_mix64(key) : key;
  return (uint32_t)(x >> shift) & (RADIX - 1);
}

// ghist layout: [batch][digit][tile]
template <bool MIX>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                                 int ntiles, int shift,
                                                                 uint32_t *__restrict__ ghist) {
  __shared__ uint32_t h[RADIX];
  const int tile = blockIdx.x, batch = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t *k = keys + (size_t)batch * n;
  const int64_t base = (int64_t)tile * SORT_TILE;
#pragma unroll
  for (int i = 0; i < SORT_IPT; ++i) {
    const int64_t idx = base + (int64_t)i * SORT_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[digit_of<MIX>(k[idx], shift)], 1u);
  }
  __syncthreads();
  ghist[((size_t)batch * RADIX + threadIdx.x) * ntiles + tile] = h[threadIdx.x];
}

// One workgroup per (digit, batch) row of ghist: exclusive scan of the row's ntiles tile
// counts in place, and the row total to rtot[batch][digit].  The scatter kernel turns the
// 256 row totals into digit bases itself, so a pass needs no single-workgroup scan.
__global__ __launch_bounds__(256) void sort_rowscan_kernel(uint32_t *__restrict__ ghist, int ntiles,
                                                           uint32_t *__restrict__ rtot) {
  __shared__ uint32_t wsum[4];
  const int d = blockIdx.x, batch = blockIdx.y;
  uint32_t *row = ghist + ((size_t)batch * RADIX + d) * ntiles;
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < ntiles; base += 256) {
    const int i = base + t;
    const uint32_t v = i < ntiles ? row[i] : 0;
    uint32_t inc = v;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t o = __shfl_up(inc, k, WAVE);
      if (lane >= k) inc += o;
    }
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t x = wsum[k];
      if (k < w) wbase += x;
      tot += x;
    }
    if (i < ntiles) row[i] = carry + wbase + inc - v;
    carry += tot;
    __syncthreads();
  }
  if (t == 0) rtot[(size_t)batch * RADIX + d] = carry;
}

template <bool MIX, bool HAS_VAL, bool IOTA>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const uint64_t *__restrict__ keys_in,
                                                                    const uint32_t *__restrict__ vals_in,
                                                                    uint64_t *__restrict__ keys_out,
                                                                    uint32_t *__restrict__ vals_out, int64_t n,
                                                                    int ntiles, int shift,
                                                                    const uint32_t *__restrict__ goff,
                                                                    const uint32_t *__restrict__ rtot) {
  __shared__ uint32_t cnt[SORT_THREADS / WAVE][RADIX];
  __shared__ uint32_t dsum[SORT_THREADS / WAVE];
  const int tile = blockIdx.x, batch = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < SORT_THREADS / WAVE; ++i) cnt[i][threadIdx.x] = 0;
  __syncthreads();

  const size_t boff = (size_t)batch * n;
  const int64_t wbase = (int64_t)tile * SORT_TILE + (int64_t)w * (WAVE * SORT_IPT);
  uint64_t key[SORT_IPT];
  uint32_t val[SORT_IPT];
  uint32_t dr[SORT_IPT];  // digit << 16 | rank within this wave's part of the tile
  const uint64_t lt_mask = (1ull << lane) - 1ull;

#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n;
    key[k] = valid ? keys_in[boff + idx] : 0;
    if (HAS_VAL) val[k] = IOTA ? (uint32_t)idx : (valid ? vals_in[boff + idx] : 0);
  }
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = digit_of<MIX>(key[k], shift);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const uint64_t bal = __ballot(one);
      m &= one ? bal : ~bal;
    }
    const uint32_t below = (uint32_t)__popcll(m & lt_mask);
    uint32_t prev = 0;
    if (valid) {
      prev = cnt[w][d];
      if (below == 0) cnt[w][d] = prev + (uint32_t)__popcll(m);
    }
    dr[k] = (d << 16) | (prev + below);
  }
  __syncthreads();
  {
    // chain the waves: cnt[w][d] becomes the global position of wave w's first key with digit d
    const int d = threadIdx.x;
    // digit base = exclusive prefix of the 256 row totals of this batch
    const uint32_t tot = rtot[(size_t)batch * RADIX + d];
    uint32_t inc = tot;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t o = __shfl_up(inc, k, WAVE);
      if (lane >= k) inc += o;
    }
    if (lane == WAVE - 1) dsum[w] = inc;
    __syncthreads();
    uint32_t dbase = inc - tot;
#pragma unroll
    for (int k = 0; k < SORT_THREADS / WAVE; ++k)
      if (k < w) dbase += dsum[k];
    uint32_t run = dbase + goff[((size_t)batch * RADIX + d) * ntiles + tile];
#pragma unroll
    for (int i = 0; i < SORT_THREADS / WAVE; ++i) {
      const uint32_t c = cnt[i][d];
      cnt[i][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    if (idx < n) {
      const uint32_t d = dr[k] >> 16;
      const size_t dst = boff + cnt[w][d] + (dr[k] & 0xFFFFu);
      keys_out[dst] = key[k];
      if (HAS_VAL) vals_out[dst] = val[k];
    }
  }
}

QRLSH_EXPORT size_t qrlsh_sort_workspace_bytes(int64_t n, int32_t nbatch) {
  if (n <= 0 || nbatch <= 0) return 16;
  const int64_t ntiles = ceil_div64(n, SORT_TILE);
  return (size_t)nbatch * RADIX * (ntiles + 1) * sizeof(uint32_t);
}

template <bool MIX>
static int sort_passes(uint64_t *ka, uint64_t *kb, uint32_t *va, uint32_t *vb, int64_t n, int nbatch, int bit_lo,
                       int bit_hi, bool iota, uint32_t *ghist, hipStream_t st) {
  const int ntiles = (int)ceil_div64(n, SORT_TILE);
  const dim3 grid(ntiles, nbatch), block(SORT_THREADS);
  const bool has_val = va != nullptr;
  uint32_t *rtot = ghist + (size_t)nbatch * RADIX * ntiles;
  int cur = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    uint64_t *kin = cur ? kb : ka, *kout = cur ? ka : kb;
    uint32_t *vin = cur ? vb : va, *vout = cur ? va : vb;
    QR_LAUNCH("sort_hist", (sort_hist_kernel<MIX>), grid, block, 0, st, kin, n, ntiles, shift, ghist);
    QR_LAUNCH("sort_rowscan", sort_rowscan_kernel, dim3(RADIX, nbatch), dim3(256), 0, st, ghist, ntiles, rtot);
    if (!has_val)
      QR_LAUNCH("sort_scatter_k", (sort_scatter_kernel<MIX, false, false>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot);
    else if (iota && shift == bit_lo)
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<MIX, true, true>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot);
    else
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<MIX, true, false>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot);
    cur ^= 1;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    qrlsh_set_error("qrlsh_sort_u64: launch failed: %s", hipGetErrorString(e));
    return QRLSH_EHIP;
  }
  return cur;
}

QRLSH_EXPORT int qrlsh_sort_u64(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n,
                                int32_t nbatch, int32_t bit_lo, int32_t bit_hi, uint32_t flags, void *workspace,
                                size_t workspace_bytes, void *stream) {
  QR_CHECK_ARG(n >= 0 && nbatch > 0, "qrlsh_sort_u64: bad sizes n=%lld nbatch=%d", (long long)n, nbatch);
  QR_CHECK_ARG(n < (1ll << 32), "qrlsh_sort_u64: n=%lld per batch exceeds 2^32-1", (long long)n);
  QR_CHECK_ARG(bit_lo >= 0 && bit_hi <= 64 && bit_lo <= bit_hi, "qrlsh_sort_u64: bad bit range [%d,%d)", bit_lo,
               bit_hi);
  QR_CHECK_ARG((vals_a == nullptr) == (vals_b == nullptr), "qrlsh_sort_u64: vals_a/vals_b must both be set or NULL");
  if (n == 0 || bit_lo == bit_hi) return 0;
  QR_CHECK_ARG(keys_a && keys_b && workspace, "qrlsh_sort_u64: null pointer");
  if (workspace_bytes < qrlsh_sort_workspace_bytes(n, nbatch)) {
    qrlsh_set_error("qrlsh_sort_u64: workspace %zu < %zu bytes", workspace_bytes,
                    qrlsh_sort_workspace_bytes(n, nbatch));
    return QRLSH_EWORKSPACE;
  }
  QR_CHECK_ARG(ceil_div64(n, SORT_TILE) <= 2147483647ll && nbatch <= 65535, "qrlsh_sort_u64: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t *ghist = static_cast<uint32_t *>(workspace);
  const bool iota = (flags & QRLSH_SORT_IOTA) != 0;
  if (flags & QRLSH_SORT_MIX)
    return sort_passes<true>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_hi, iota, ghist, st);
  return sort_passes<false>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_hi, iota, ghist, st);
}
