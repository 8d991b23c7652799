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

// Workgroups b and b+8 share an XCD (round-robin dispatch; speed only, never correctness).
// Remap so each XCD works on a contiguous range of tiles: the runs that neighbouring tiles
// write for one digit are adjacent in memory, and their shared 64-B sectors then merge in
// ONE L2 instead of being written back partially by two.
__device__ static inline int xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, y = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
}

// digit source: MODE 0 = the key itself, 1 = mix64(key) (grouping sort), 2 = the key with its
// two 32-bit halves packed next to each other, hi << fold | lo (pairs i << 32 | j sort in
// ceil(2*id_bits / 8) passes instead of 2 * ceil(id_bits / 8)).
// 3 = owner: (key >> shift) / aux, one pass that groups words by the rank owning the id field
// (contiguous shards of aux ids each; at most 256 ranks).
// 4 = host: the word is a pair i << 32 | j; digit = the rank that scores it (qr_pair_host: the owner of i
// or of j, chosen by one bit of mix64(pair) so that every rank gets an equal share whatever the data).
enum { SM_PLAIN = 0, SM_MIX = 1, SM_FOLD = 2, SM_OWNER = 3, SM_HOST = 4 };
template <int MODE> __device__ static inline uint32_t digit_of(uint64_t key, int shift, uint32_t fold = 0) {
  if (MODE == SM_OWNER || MODE == SM_HOST) {
    const uint64_t o = MODE == SM_HOST ? qr_pair_host(key, fold) : (key >> shift) / fold;
    return o < RADIX ? (uint32_t)o : RADIX - 1;
  }
  uint64_t x = key;
  if (MODE == SM_MIX) x = qr_mix64(key);
  if (MODE == SM_FOLD) x = ((key >> 32) << fold) | (key & ((1ull << fold) - 1ull));
  return (uint32_t)(x >> shift) & (RADIX - 1);
}

// Partition digit of the fast bucket path: records holding the "empty" key (all -1 band,
// never a candidate: lsh.py:47) are dealt round the parts by their index instead of all
// landing in one part, so a data set with many empty answer sets cannot overflow a part.
template <bool SPREAD>
__device__ static inline uint32_t part_digit(uint64_t key, int64_t idx, int shift, uint64_t ek, uint32_t dmask) {
  const uint64_t x = (SPREAD && key == ek) ? qr_mix64((uint64_t)idx) : qr_mix64(key);
  return (uint32_t)(x >> shift) & dmask;
}

// ghist layout: [batch][digit][tile]
template <int MIX, bool SPREAD = false>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                                 int ntiles, int shift,
                                                                 uint32_t *__restrict__ ghist, uint64_t ek = 0,
                                                                 uint32_t fold = 0, uint32_t dmask = RADIX - 1,
                                                                 const uint32_t *__restrict__ vals = nullptr) {
  __shared__ uint32_t h[RADIX];
  const int tile = blockIdx.x, batch = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t *k = keys + (size_t)batch * n;
  const uint32_t *v = vals ? vals + (size_t)batch * n : nullptr;
  const int64_t base = (int64_t)tile * SORT_TILE;
  // two keys per lane per step (16-byte loads) when the batch base is 16-byte aligned
  const bool wide = ((((uintptr_t)k) & 15) == 0);
#pragma unroll
  for (int i = 0; i < SORT_IPT / 2; ++i) {
    const int64_t idx0 = base + ((int64_t)i * SORT_THREADS + threadIdx.x) * 2;
    uint64_t kk[2] = {0, 0};
    if (wide && idx0 + 1 < n) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      const u64x2 t = *reinterpret_cast<const u64x2 *>(k + idx0);
      kk[0] = t.x;
      kk[1] = t.y;
    } else {
      if (idx0 < n) kk[0] = k[idx0];
      if (idx0 + 1 < n) kk[1] = k[idx0 + 1];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t idx = idx0 + e;
      uint32_t dd = 0;
      if (idx < n) {
        const uint64_t key = kk[e];
        uint32_t d;
        if (SPREAD)
          d = part_digit<SPREAD>(key, (key == ek && v) ? (int64_t)v[idx] : idx, shift, ek, dmask);
        else
          d = digit_of<MIX>(key, shift, fold);
        if (MIX != SM_OWNER && MIX != SM_HOST) atomicAdd(&h[d], 1u);
        dd = d;
      }
      if (MIX == SM_OWNER || MIX == SM_HOST) {
        // a handful of distinct digits (ranks): one LDS atomic per digit per wave, not per key
        const bool valid = idx < n;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          const bool one = (dd >> bit) & 1u;
          const uint64_t bal = __ballot(one);
          m &= one ? bal : ~bal;
        }
        const int lane = threadIdx.x & (WAVE - 1);
        if (valid && (m & ((1ull << lane) - 1ull)) == 0) atomicAdd(&h[dd], (uint32_t)__popcll(m));
      }
    }
  }
  __syncthreads();
  ghist[((size_t)batch * RADIX + threadIdx.x) * ntiles + tile] = h[threadIdx.x];
}

// One workgroup per (digit, batch) row of ghist: exclusive scan of the row's ntiles tile
// counts in place, and the row total to rtot[batch][digit].  The scatter kernel turns the
// 256 row totals into digit bases itself, so a pass needs no single-workgroup scan.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void sort_rowscan_kernel(uint32_t *__restrict__ ghist, int ntiles,
                                                               uint32_t *__restrict__ rtot) {
  constexpr int NW = THREADS / WAVE;
  __shared__ uint32_t wsum[NW];
  const int d = blockIdx.x, batch = blockIdx.y;
  uint32_t *row = ghist + ((size_t)batch * RADIX + d) * ntiles;
  const int t = threadIdx.x, lane = t & (WAVE - 1), w = t >> 6;
  // rows are scanned from registers: 256 threads x 32 for up to 8192 tiles (33 M keys per batch),
  // 1024 threads x 64 for up to 65536 tiles (268 M keys); longer rows take the chunked loop below
  constexpr int ROW_REG = THREADS == 256 ? 32 : 64;
  const int per = (ntiles + THREADS - 1) / THREADS;
  if (per <= ROW_REG) {
    // blocked layout: thread t owns `per` consecutive tiles; every load is issued before the first
    // add, and the whole row needs one workgroup scan instead of one per THREADS tiles
    const int lo = t * per;
    uint32_t held[ROW_REG], s = 0;
#pragma unroll
    for (int k = 0; k < ROW_REG; ++k) held[k] = (k < per && lo + k < ntiles) ? row[lo + k] : 0u;
#pragma unroll
    for (int k = 0; k < ROW_REG; ++k) s += held[k];
    uint32_t inc = s;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t o = __shfl_up(inc, k, WAVE);
      if (lane >= k) inc += o;
    }
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - s, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const uint32_t x = wsum[k];
      if (k < w) run += x;
      tot += x;
    }
#pragma unroll
    for (int k = 0; k < ROW_REG; ++k)
      if (k < per && lo + k < ntiles) {
        row[lo + k] = run;
        run += held[k];
      }
    if (t == 0) rtot[(size_t)batch * RADIX + d] = tot;
    return;
  }
  uint32_t carry = 0;
  for (int base = 0; base < ntiles; base += THREADS) {
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
    for (int k = 0; k < NW; ++k) {
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

// one row-scan launch: 256-thread workgroups while a row fits their registers, 1024 above
static void launch_rowscan(uint32_t *ghist, int ntiles, uint32_t *rtot, int nbatch, hipStream_t st) {
  if (ntiles <= 256 * 32)
    QR_LAUNCH("sort_rowscan", sort_rowscan_kernel<256>, dim3(RADIX, nbatch), dim3(256), 0, st, ghist, ntiles, rtot);
  else
    QR_LAUNCH("sort_rowscan", sort_rowscan_kernel<1024>, dim3(RADIX, nbatch), dim3(1024), 0, st, ghist, ntiles, rtot);
}

template <int MIX, bool HAS_VAL, bool IOTA, bool SPREAD = false>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const uint64_t *__restrict__ keys_in,
                                                                    const uint32_t *__restrict__ vals_in,
                                                                    uint64_t *__restrict__ keys_out,
                                                                    uint32_t *__restrict__ vals_out, int64_t n,
                                                                    int ntiles, int shift,
                                                                    const uint32_t *__restrict__ goff,
                                                                    const uint32_t *__restrict__ rtot,
                                                                    uint64_t ek = 0, uint32_t fold = 0,
                                                                    uint32_t dmask = RADIX - 1) {
  __shared__ uint32_t cnt[SORT_THREADS / WAVE][RADIX];
  __shared__ uint32_t dsum[SORT_THREADS / WAVE];
  const int tile = xcd_tile(blockIdx.x, ntiles), batch = blockIdx.y;
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
    const uint32_t d = SPREAD ? part_digit<SPREAD>(key[k], HAS_VAL ? (int64_t)val[k] : idx, shift, ek, dmask)
                              : digit_of<MIX>(key[k], shift, fold);
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

// Keys-only scatter with LDS staging: after ranking, the tile is laid out in LDS in its
// sorted-by-digit order and written out by consecutive lanes, so a store instruction covers a few
// contiguous runs instead of up to 64 unrelated lines.  Same inputs / outputs as
// sort_scatter_kernel<MODE, false, false>; 37 KB of LDS keeps the 4 workgroups per CU that the
// register budget allows anyway.
template <int MODE>
__global__ __launch_bounds__(SORT_THREADS, 4) void sort_scatter_staged_kernel(const uint64_t *__restrict__ keys_in,
                                                                           uint64_t *__restrict__ keys_out, int64_t n,
                                                                           int ntiles, int shift,
                                                                           const uint32_t *__restrict__ goff,
                                                                           const uint32_t *__restrict__ rtot,
                                                                           uint32_t fold) {
  __shared__ uint32_t cnt[SORT_THREADS / WAVE][RADIX];
  __shared__ uint32_t dsum[SORT_THREADS / WAVE];
  __shared__ uint32_t lsum[SORT_THREADS / WAVE];
  __shared__ uint32_t gdelta[RADIX];
  __shared__ uint64_t skey[SORT_TILE];
  const int tile = xcd_tile(blockIdx.x, ntiles), batch = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < SORT_THREADS / WAVE; ++i) cnt[i][threadIdx.x] = 0;
  __syncthreads();
  const size_t boff = (size_t)batch * n;
  const int64_t tbase = (int64_t)tile * SORT_TILE;
  const int64_t wbase = tbase + (int64_t)w * (WAVE * SORT_IPT);
  uint64_t key[SORT_IPT];
  uint32_t dr[SORT_IPT];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    key[k] = idx < n ? keys_in[boff + idx] : 0;
  }
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = digit_of<MODE>(key[k], shift, fold);
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
    const int d = threadIdx.x;
    // digit base (global) = exclusive prefix of the 256 row totals; tile-local start = exclusive
    // prefix of this tile's 256 digit counts
    const uint32_t tot = rtot[(size_t)batch * RADIX + d];
    uint32_t tc = 0;
#pragma unroll
    for (int i = 0; i < SORT_THREADS / WAVE; ++i) tc += cnt[i][d];
    uint32_t inc = tot, linc = tc;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t o = __shfl_up(inc, k, WAVE), lo = __shfl_up(linc, k, WAVE);
      if (lane >= k) {
        inc += o;
        linc += lo;
      }
    }
    if (lane == WAVE - 1) {
      dsum[w] = inc;
      lsum[w] = linc;
    }
    __syncthreads();
    uint32_t dbase = inc - tot, lstart = linc - tc;
#pragma unroll
    for (int k = 0; k < SORT_THREADS / WAVE; ++k)
      if (k < w) {
        dbase += dsum[k];
        lstart += lsum[k];
      }
    gdelta[d] = dbase + goff[((size_t)batch * RADIX + d) * ntiles + tile] - lstart;
    uint32_t run = lstart;
#pragma unroll
    for (int i = 0; i < SORT_THREADS / WAVE; ++i) {
      const uint32_t c = cnt[i][d];
      cnt[i][d] = run;  // tile-local position of wave i's first key with digit d
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    if (idx < n) skey[cnt[w][dr[k] >> 16] + (dr[k] & 0xFFFFu)] = key[k];
  }
  __syncthreads();
  const int ntile = (int)min((int64_t)SORT_TILE, n - tbase);
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int p = k * SORT_THREADS + threadIdx.x;
    if (p < ntile) {
      const uint64_t kk = skey[p];
      keys_out[boff + gdelta[digit_of<MODE>(kk, shift, fold)] + (uint32_t)p] = kk;
    }
  }
}

// The same staging for the key + id partition passes of the fast bucket path (digit = part_digit,
// SPREAD).  Ids are < 2^24 there (the host checks nq), so the digit rides in the top byte of the
// staged id and is not recomputed (mix64 again) when the tile is written out.
template <bool IOTA>
__global__ __launch_bounds__(SORT_THREADS, 3) void part_scatter_staged_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int64_t n, int ntiles, int shift, const uint32_t *__restrict__ goff,
    const uint32_t *__restrict__ rtot, uint64_t ek, uint32_t dmask) {
  __shared__ uint32_t cnt[SORT_THREADS / WAVE][RADIX];
  __shared__ uint32_t dsum[SORT_THREADS / WAVE];
  __shared__ uint32_t lsum[SORT_THREADS / WAVE];
  __shared__ uint32_t gdelta[RADIX];
  __shared__ uint64_t skey[SORT_TILE];
  __shared__ uint32_t sval[SORT_TILE];
  const int tile = xcd_tile(blockIdx.x, ntiles), batch = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < SORT_THREADS / WAVE; ++i) cnt[i][threadIdx.x] = 0;
  __syncthreads();
  const size_t boff = (size_t)batch * n;
  const int64_t tbase = (int64_t)tile * SORT_TILE;
  const int64_t wbase = tbase + (int64_t)w * (WAVE * SORT_IPT);
  uint64_t key[SORT_IPT];
  uint32_t val[SORT_IPT];
  uint32_t dr[SORT_IPT];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n;
    key[k] = valid ? keys_in[boff + idx] : 0;
    val[k] = IOTA ? (uint32_t)idx : (valid ? vals_in[boff + idx] : 0);
  }
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = part_digit<true>(key[k], (int64_t)val[k], shift, ek, dmask);
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
    const int d = threadIdx.x;
    const uint32_t tot = rtot[(size_t)batch * RADIX + d];
    uint32_t tc = 0;
#pragma unroll
    for (int i = 0; i < SORT_THREADS / WAVE; ++i) tc += cnt[i][d];
    uint32_t inc = tot, linc = tc;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t o = __shfl_up(inc, k, WAVE), lo = __shfl_up(linc, k, WAVE);
      if (lane >= k) {
        inc += o;
        linc += lo;
      }
    }
    if (lane == WAVE - 1) {
      dsum[w] = inc;
      lsum[w] = linc;
    }
    __syncthreads();
    uint32_t dbase = inc - tot, lstart = linc - tc;
#pragma unroll
    for (int k = 0; k < SORT_THREADS / WAVE; ++k)
      if (k < w) {
        dbase += dsum[k];
        lstart += lsum[k];
      }
    gdelta[d] = dbase + goff[((size_t)batch * RADIX + d) * ntiles + tile] - lstart;
    uint32_t run = lstart;
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
      const uint32_t d = dr[k] >> 16, lp = cnt[w][d] + (dr[k] & 0xFFFFu);
      skey[lp] = key[k];
      sval[lp] = val[k] | d << 24;
    }
  }
  __syncthreads();
  const int ntile = (int)min((int64_t)SORT_TILE, n - tbase);
#pragma unroll
  for (int k = 0; k < SORT_IPT; ++k) {
    const int p = k * SORT_THREADS + threadIdx.x;
    if (p < ntile) {
      const uint32_t vv = sval[p];
      const size_t dst = boff + gdelta[vv >> 24] + (uint32_t)p;
      keys_out[dst] = skey[p];
      vals_out[dst] = vv & 0xFFFFFFu;
    }
  }
}

// One-kernel partition for the one-pass bucket path (256 parts): every part owns a fixed region of
// `cap` records, a tile reserves room in each part with one atomicAdd per (tile, part) on the part's
// cursor, and writes its records there through the same LDS staging as above.  No histogram pass, no
// row scan, no bounds search; the order of the records inside a part is whatever the atomics gave
// (the finish does not care).  A part that would exceed `cap` raises the overflow word and its
// records are dropped -- the caller then takes the general path.
// LEVEL2 = false: the input is the band-major key matrix ([batch][n], ids = positions), 256 parts per band.
// LEVEL2 = true : finer partitions (T > 8 bits) take a second step -- the input is the OUTPUT of a first
// step, one batch per (band, coarse part): its in_counts[batch] records sit at batch * in_cap and are
// dealt to nd = 2^(T-c1) fine parts by the next bits of the same hash; ids come with the records.
// OVERFLOW POOL (round 4).  A part's region holds ONE LDS image of the finish (reserved memory ~ 1.7 - 2 x the records
// instead of the 7.6 x of three-image regions), and a part swollen by a popular key -- at 100 M queries over 32768
// table rows the luckiest (row, band) makes one band key common to ~20 000 queries -- spills into a pool shared by all
// parts: a (tile, part) run that does not fit the part's region takes its room from a device bump cursor instead and
// leaves a descriptor {part slot, records, pool position}; the first such run also records how many records the
// region really holds (`fill`: reservations are handed out in cursor order, so the region holds a prefix of them).
// The finish lists such parts like every part beyond its image; bucket_big_gather_kernel then puts each one's records
// (region prefix + its runs) next to each other in the pool, where the block kernel works them.  No pool (keys ==
// nullptr: the coarse step of a two-step partition): an overflowing part raises the flag, as before.
struct PartPool {
  uint64_t *keys;               // pool records (the x words) ...
  uint32_t *vals;               // ... and their ids
  unsigned long long *cursor;   // records handed out so far
  uint32_t cap;                 // records the pool holds (< 2^32)
  uint4 *runs;                  // {part slot, records, pool position, 0} per spilled run
  unsigned long long *nruns;
  uint32_t runs_max;
  uint32_t *fill;               // per part slot: records that sit in its region (0xFFFFFFFF: all of them)
  uint32_t slot_base;           // part slot of this launch's (batch 0, part 0)
};

#ifndef QR_PS_IPT
#define QR_PS_IPT 16
#endif
constexpr int PS_IPT = QR_PS_IPT;  // records per thread of the atomic partition (8, six workgroups per CU: 3.9 ms against 3.4)
constexpr int PS_TILE = SORT_THREADS * PS_IPT;
constexpr int PS_WGS = PS_IPT <= 16 ? 3 : 2;   // workgroups per CU the LDS image (12 B per record) allows
template <bool LEVEL2>
__global__ __launch_bounds__(SORT_THREADS, PS_WGS) void part_scatter_atomic_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int64_t n_in, int ntiles, int shift, uint32_t dmask,
    uint32_t *__restrict__ cursors, uint32_t cap, uint32_t *__restrict__ overflow, uint64_t ek,
    const uint32_t *__restrict__ in_counts, uint32_t in_cap, int64_t chunk_len, int64_t chunk_stride,
    int64_t band_stride, PartPool pool) {
  // What travels through the partition is x = mix64(key), not the key: mix64 is a bijection, so equal x <=> equal
  // keys and the finish can pair on x; the part number of either step is then a shift of the staged word (no
  // second hash in the second step, none at the write-out, nothing to carry in the id word -- ids keep all 32
  // bits for any number of queries).  Records of empty bands (key == ek) never pair (lsh.py:47) and are dropped
  // here; mix64(ek), which no other key maps to, is the finish's free-slot marker.
  // The order of the records inside a part is free, so a record's place in its tile's share of a part is the old
  // value of an LDS counter (one returning ds_add per record), not the eight ballots + popcount a stable rank
  // costs: the kernel was VALU-bound on those (2200 vector instructions per wave, 66 % VALU-busy).
  __shared__ uint32_t cnt[RADIX];
  __shared__ uint32_t lsum[SORT_THREADS / WAVE];
  __shared__ uint32_t gdelta[RADIX];
  __shared__ uint8_t gok[RADIX];
  __shared__ uint64_t skey[PS_TILE];
  __shared__ uint32_t sval[PS_TILE];
  const int tile = LEVEL2 ? (int)blockIdx.x : xcd_tile(blockIdx.x, ntiles), batch = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int64_t n = LEVEL2 ? (int64_t)min(in_counts[batch], in_cap) : n_in;
  const int64_t tbase = (int64_t)tile * PS_TILE;
  if (tbase >= n) return;  // LEVEL2: the grid covers a full region, this one holds fewer records (uniform)
  cnt[threadIdx.x] = 0;
  __syncthreads();
  // first step: band `batch` of the key matrix, either plain ([b][n]: band_stride = n) or in chunks of
  // chunk_len queries chunk_stride words apart (what a band-partitioned all-to-all delivers: [rank][band][nql])
  const size_t boff = LEVEL2 ? (size_t)batch * in_cap : (size_t)batch * (size_t)(band_stride ? band_stride : n);
  const bool chunked = !LEVEL2 && chunk_len > 0 && chunk_len < n;
  const int64_t wbase = tbase + (int64_t)w * (WAVE * PS_IPT);
  const uint32_t nd = dmask + 1u;  // parts per batch
  uint64_t key[PS_IPT];
  uint32_t val[PS_IPT];
  uint32_t dr[PS_IPT];  // part << 16 | place among the tile's records of that part; 0xFFFFFFFF = no record
#pragma unroll
  for (int k = 0; k < PS_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const size_t at = chunked ? boff + (size_t)(idx / chunk_len) * chunk_stride + (size_t)(idx % chunk_len) : boff + idx;
    key[k] = idx < n ? keys_in[at] : ek;
    val[k] = LEVEL2 ? (idx < n ? vals_in[boff + idx] : 0u) : (uint32_t)idx;
  }
#pragma unroll
  for (int k = 0; k < PS_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const bool valid = idx < n && (LEVEL2 || key[k] != ek);
    if (!LEVEL2) key[k] = qr_mix64(key[k]);
    const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
    dr[k] = valid ? (d << 16) | atomicAdd(&cnt[d], 1u) : 0xFFFFFFFFu;
  }
  __syncthreads();
  // thread d speaks for part d.  The reservation goes out first and its result is not touched until the
  // tile has been laid out in LDS (which needs local positions only): the atomic's round trip to memory
  // runs beside the scan and the staging.
  const uint32_t tc = cnt[threadIdx.x];
  const uint32_t gb = tc ? atomicAdd(&cursors[(size_t)batch * nd + threadIdx.x], tc) : 0u;
  uint32_t lstart;
  {
    uint32_t linc = tc;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t lo = __shfl_up(linc, k, WAVE);
      if (lane >= k) linc += lo;
    }
    if (lane == WAVE - 1) lsum[w] = linc;
    __syncthreads();
    lstart = linc - tc;
#pragma unroll
    for (int k = 0; k < SORT_THREADS / WAVE; ++k)
      if (k < w) lstart += lsum[k];
    cnt[threadIdx.x] = lstart;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PS_IPT; ++k) {
    if (dr[k] != 0xFFFFFFFFu) {
      const uint32_t lp = cnt[dr[k] >> 16] + (dr[k] & 0xFFFFu);
      skey[lp] = key[k];
      sval[lp] = val[k];
    }
  }
  {
    const int d = threadIdx.x;
    uint32_t where = gb + tc <= cap ? 1u : 0u;    // 1: the part's region, 2: the pool, 0: nowhere (overflow flag)
    uint32_t delta = (uint32_t)d * cap + gb - lstart;  // mod 2^32; + the staged position gives the place in the batch
    if (!where) {
      if (pool.keys) {  // the run spills: room from the pool's cursor, a descriptor, the region's fill mark
        const unsigned long long pb =
            __hip_atomic_fetch_add(pool.cursor, (unsigned long long)tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pb + tc <= (unsigned long long)pool.cap) {
          const unsigned long long ri = __hip_atomic_fetch_add(pool.nruns, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ri < (unsigned long long)pool.runs_max) {
            const uint32_t slot = pool.slot_base + (uint32_t)batch * nd + (uint32_t)d;
            pool.runs[ri] = make_uint4(slot, tc, (uint32_t)pb, 0u);
            atomicMin(&pool.fill[slot], gb);
            where = 2u;
            delta = (uint32_t)pb - lstart;
          }
        }
      }
      if (!where) atomicOr(overflow, 1u);
    }
    gok[d] = (uint8_t)where;
    gdelta[d] = delta;
  }
  __syncthreads();
  uint32_t nstaged = 0;  // records of the tile that are not of an empty band
#pragma unroll
  for (int k = 0; k < SORT_THREADS / WAVE; ++k) nstaged += lsum[k];
  const size_t obase = (size_t)batch * nd * cap;
#pragma unroll
  for (int k = 0; k < PS_IPT; ++k) {
    const uint32_t p = k * SORT_THREADS + threadIdx.x;
    if (p < nstaged) {
      const uint64_t x = skey[p];
      const uint32_t d = (uint32_t)(x >> shift) & dmask;
      const uint32_t where = gok[d];
      if (where == 1u) {
        const size_t dst = obase + (uint32_t)(gdelta[d] + p);
        keys_out[dst] = x;
        vals_out[dst] = sval[p];
      } else if (where == 2u) {
        const uint32_t dst = gdelta[d] + p;
        pool.keys[dst] = x;
        pool.vals[dst] = sval[p];
      }
    }
  }
}

// ---- pair words grouped by REGION without histogram passes (round 4) ------------------------------------------------------
// The region form of the de-duplication (pairs.hip) needs the emitted words grouped by their region id (i >> g, up to 16
// bits) and NOTHING about the order inside a group.  The stable LSD sort pays for an order nobody reads: per 8-bit pass a
// histogram pass over the words, a scan, and the scatter.  Here the words are dealt most-significant digit first by the
// partition kernel of the bucket path, words only: every digit owns a fixed region of `cap` words, a tile counts its
// digits in LDS, reserves room with ONE atomic per (tile, digit) and writes its staged words in runs.  Level 1 deals by the
// high digit of the region id into tmp regions, level 2 deals every tmp region by the low digit into the final regions
// (region r at r * cap, counts[r] words).  One read + one write of the words per level -- 2.1 -> 1.4 ms for the 190 M words
// of the 10 M-query workload.  A region that outgrows its cap raises the flag (the caller groups by sorting instead).
#ifndef QR_PG_IPT
#define QR_PG_IPT 32   // 8192-word tiles: runs of 32 - 54 words per (tile, digit); 16: 1.77 ms for the two levels at 10 M, 32: 1.46
#endif
constexpr int PG_IPT = QR_PG_IPT;                 // words per thread of the pair-grouping partition
constexpr int PG_TILE = SORT_THREADS * PG_IPT;
constexpr int PG_WGS = PG_IPT <= 16 ? 4 : PG_IPT <= 40 ? 2 : 1;      // workgroups per CU the staged tile allows
template <bool LEVEL2>
__global__ __launch_bounds__(SORT_THREADS, PG_WGS) void pair_group_scatter_kernel(
    const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int64_t n_in, int ntiles, int shift, uint32_t dmask,
    uint32_t *__restrict__ cursors, uint32_t cap, uint32_t *__restrict__ overflow, const uint32_t *__restrict__ in_counts,
    uint32_t in_cap) {
  __shared__ uint32_t cnt[RADIX];
  __shared__ uint32_t lsum[SORT_THREADS / WAVE];
  __shared__ uint32_t gdelta[RADIX];
  __shared__ uint8_t gok[RADIX];
  __shared__ uint64_t skey[PG_TILE];
  const int tile = LEVEL2 ? (int)blockIdx.x : xcd_tile(blockIdx.x, ntiles), batch = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  const int64_t n = LEVEL2 ? (int64_t)min(in_counts[batch], in_cap) : n_in;
  const int64_t tbase = (int64_t)tile * PG_TILE;
  if (tbase >= n) return;  // LEVEL2: the grid covers a full region, this one holds fewer words (uniform)
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const size_t boff = LEVEL2 ? (size_t)batch * in_cap : 0;
  const int64_t wbase = tbase + (int64_t)w * (WAVE * PG_IPT);
  const uint32_t nd = dmask + 1u;
  uint64_t key[PG_IPT];
  uint32_t dr[PG_IPT];
#pragma unroll
  for (int k = 0; k < PG_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    key[k] = idx < n ? in[boff + idx] : 0ull;
  }
#pragma unroll
  for (int k = 0; k < PG_IPT; ++k) {
    const int64_t idx = wbase + (int64_t)k * WAVE + lane;
    const uint32_t d = (uint32_t)(key[k] >> shift) & dmask;
    dr[k] = idx < n ? (d << 16) | atomicAdd(&cnt[d], 1u) : 0xFFFFFFFFu;
  }
  __syncthreads();
  const uint32_t tc = cnt[threadIdx.x];
  const uint32_t gb = tc ? atomicAdd(&cursors[(size_t)batch * nd + threadIdx.x], tc) : 0u;
  uint32_t lstart;
  {
    uint32_t linc = tc;
#pragma unroll
    for (int k = 1; k < WAVE; k <<= 1) {
      const uint32_t lo = __shfl_up(linc, k, WAVE);
      if (lane >= k) linc += lo;
    }
    if (lane == WAVE - 1) lsum[w] = linc;
    __syncthreads();
    lstart = linc - tc;
#pragma unroll
    for (int k = 0; k < SORT_THREADS / WAVE; ++k)
      if (k < w) lstart += lsum[k];
    cnt[threadIdx.x] = lstart;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PG_IPT; ++k)
    if (dr[k] != 0xFFFFFFFFu) skey[cnt[dr[k] >> 16] + (dr[k] & 0xFFFFu)] = key[k];
  {
    const int d = threadIdx.x;
    const bool ok = (uint64_t)gb + tc <= cap;
    if (!ok) atomicOr(overflow, 1u);
    gok[d] = ok;
    gdelta[d] = (uint32_t)d * cap + gb - lstart;
  }
  __syncthreads();
  uint32_t nstaged = 0;
#pragma unroll
  for (int k = 0; k < SORT_THREADS / WAVE; ++k) nstaged += lsum[k];
  const size_t obase = (size_t)batch * nd * cap;
#pragma unroll
  for (int k = 0; k < PG_IPT; ++k) {
    const uint32_t p = k * SORT_THREADS + threadIdx.x;
    if (p < nstaged) {
      const uint64_t x = skey[p];
      const uint32_t d = (uint32_t)(x >> shift) & dmask;
      if (gok[d]) out[obase + (uint32_t)(gdelta[d] + p)] = x;
    }
  }
}

// split of the region-id bits over the two levels, and the region capacities
struct PairRegions {
  int rbits, ra, rb;      // bits of the region id; high digit (level 1), low digit (level 2); ra == 0: one level
  int64_t nregions;       // region slots = na << rb
  int64_t na;             // level-1 digits that can occur
  uint32_t cap_a, cap_b;  // words per tmp region / per final region
};
static PairRegions pair_regions(int64_t n, int64_t nids, int group_bits, double words_per_query) {
  PairRegions r;
  const int64_t nr = (nids + (1ll << group_bits) - 1) >> group_bits;
  r.rbits = 1;
  while ((1ll << r.rbits) < nr) ++r.rbits;
  r.rb = r.rbits <= 8 ? r.rbits : (r.rbits + 1) / 2;
  r.ra = r.rbits - r.rb;
  r.na = (nr + (1ll << r.rb) - 1) >> r.rb;
  r.nregions = r.na << r.rb;
  // words a region holds on average: n / regions, or -- the words of a shard sit in a slice of the id space --
  // what the caller says a query emits
  double per = (double)n / (double)(nr > 0 ? nr : 1);
  const double hint = words_per_query * (double)(1ll << group_bits);
  if (hint > per) per = hint;
  // i is the SMALLER id of a pair: with partners anywhere in the id space the low ids carry up to twice the mean (the
  // density of the minimum of two ids falls linearly to zero at the top), popular queries come on top of that
  const double cb = 3.0 * per + 4096.0;
  const double ca = r.ra ? 2.5 * per * (double)(1ll << r.rb) + 65536.0 : 0.0;
  r.cap_b = (uint32_t)(cb > 4.0e9 ? 4.0e9 : cb);
  r.cap_b = (r.cap_b + 63u) / 64u * 64u;
  r.cap_a = (uint32_t)(ca > 4.0e9 ? 4.0e9 : ca);
  r.cap_a = (r.cap_a + 63u) / 64u * 64u;
  return r;
}

// words of the region buffer (and of the tmp buffer of level 1; 0 when one level is enough), the region capacity and count
QRLSH_EXPORT size_t qrlsh_pair_regions_words(int64_t n, int64_t nids, int32_t group_bits, double words_per_query) {
  if (n <= 0 || nids <= 0 || group_bits < 0 || group_bits > 8) return 0;
  const PairRegions r = pair_regions(n, nids, group_bits, words_per_query);
  if (r.rbits > 16) return 0;   // more than 65536 regions: not served (two levels of at most 256 digits)
  return (size_t)r.nregions * r.cap_b;
}
QRLSH_EXPORT size_t qrlsh_pair_regions_tmp_words(int64_t n, int64_t nids, int32_t group_bits, double words_per_query) {
  if (n <= 0 || nids <= 0 || group_bits < 0 || group_bits > 8) return 0;
  const PairRegions r = pair_regions(n, nids, group_bits, words_per_query);
  return r.ra ? (size_t)r.na * r.cap_a : 0;
}
QRLSH_EXPORT int64_t qrlsh_pair_regions_cap(int64_t n, int64_t nids, int32_t group_bits, double words_per_query) {
  if (n <= 0 || nids <= 0 || group_bits < 0 || group_bits > 8) return 0;
  return (int64_t)pair_regions(n, nids, group_bits, words_per_query).cap_b;
}
QRLSH_EXPORT int64_t qrlsh_pair_regions_count(int64_t n, int64_t nids, int32_t group_bits, double words_per_query) {
  if (n <= 0 || nids <= 0 || group_bits < 0 || group_bits > 8) return 0;
  return pair_regions(n, nids, group_bits, words_per_query).nregions;
}

// words (n pair words i << 32 | j, any order) -> regions[r * cap + k], k < counts[r], r = i >> group_bits; counts:
// uint32 [qrlsh_pair_regions_count + 256] (the tail is level 1's cursors); overflow_out: uint32, != 0 when a region
// outgrew its capacity (nothing usable then).  tmp_regions may be NULL when qrlsh_pair_regions_tmp_words is 0.
QRLSH_EXPORT int qrlsh_pair_regions_scatter(const uint64_t *words, int64_t n, int32_t group_bits, int64_t nids,
                                            double words_per_query, uint64_t *tmp_regions, uint64_t *regions,
                                            uint32_t *counts, uint32_t *overflow_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && n < (1ll << 32) && nids > 0 && group_bits >= 0 && group_bits <= 8 && counts && overflow_out,
               "qrlsh_pair_regions_scatter: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PairRegions r = pair_regions(n > 0 ? n : 1, nids, group_bits, words_per_query);
  QR_CHECK_ARG(r.rbits <= 16 && r.na <= RADIX, "qrlsh_pair_regions_scatter: %d region bits", r.rbits);
  if (hipMemsetAsync(counts, 0, ((size_t)r.nregions + RADIX) * sizeof(uint32_t), st) != hipSuccess ||
      hipMemsetAsync(overflow_out, 0, sizeof(uint32_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_pair_regions_scatter: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(words && regions && (r.ra == 0 || tmp_regions), "qrlsh_pair_regions_scatter: null pointer");
  const int ntiles = (int)ceil_div64(n, PG_TILE);
  const int sh = 32 + group_bits;
  if (r.ra == 0) {
    QR_LAUNCH("pair_group", (pair_group_scatter_kernel<false>), dim3(ntiles, 1), dim3(SORT_THREADS), 0, st, words, regions, n,
              ntiles, sh, (1u << r.rb) - 1u, counts, r.cap_b, overflow_out, (const uint32_t *)nullptr, 0u);
  } else {
    uint32_t *cur_a = counts + r.nregions;
    QR_LAUNCH("pair_group", (pair_group_scatter_kernel<false>), dim3(ntiles, 1), dim3(SORT_THREADS), 0, st, words, tmp_regions,
              n, ntiles, sh + r.rb, (1u << r.ra) - 1u, cur_a, r.cap_a, overflow_out, (const uint32_t *)nullptr, 0u);
    QR_LAUNCH("pair_group", (pair_group_scatter_kernel<true>), dim3((unsigned)ceil_div64(r.cap_a, PG_TILE), (unsigned)r.na),
              dim3(SORT_THREADS), 0, st, (const uint64_t *)tmp_regions, regions, (int64_t)0, 0, sh, (1u << r.rb) - 1u, counts,
              r.cap_b, overflow_out, (const uint32_t *)cur_a, r.cap_a);
  }
  QR_LAUNCH_CHECK("qrlsh_pair_regions_scatter");
  return QRLSH_OK;
}

QRLSH_EXPORT size_t qrlsh_sort_workspace_bytes(int64_t n, int32_t nbatch) {
  if (n <= 0 || nbatch <= 0) return 16;
  const int64_t ntiles = ceil_div64(n, SORT_TILE);
  return (size_t)nbatch * RADIX * (ntiles + 1) * sizeof(uint32_t);
}

template <int MIX>
static int sort_passes(uint64_t *ka, uint64_t *kb, uint32_t *va, uint32_t *vb, int64_t n, int nbatch, int bit_lo,
                       int bit_hi, bool iota, uint32_t fold, uint32_t *ghist, hipStream_t st) {
  const int ntiles = (int)ceil_div64(n, SORT_TILE);
  const dim3 grid(ntiles, nbatch), block(SORT_THREADS);
  const bool has_val = va != nullptr;
  uint32_t *rtot = ghist + (size_t)nbatch * RADIX * ntiles;
  int cur = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    uint64_t *kin = cur ? kb : ka, *kout = cur ? ka : kb;
    uint32_t *vin = cur ? vb : va, *vout = cur ? va : vb;
    QR_LAUNCH("sort_hist", (sort_hist_kernel<MIX>), grid, block, 0, st, kin, n, ntiles, shift, ghist, (uint64_t)0, fold);
    launch_rowscan(ghist, ntiles, rtot, nbatch, st);
    if (!has_val && (MIX == SM_PLAIN || MIX == SM_FOLD))
      QR_LAUNCH("sort_scatter_k", (sort_scatter_staged_kernel<MIX>), grid, block, 0, st, kin, kout, n, ntiles, shift,
                ghist, rtot, fold);
    else if (!has_val)
      QR_LAUNCH("sort_scatter_k", (sort_scatter_kernel<MIX, false, false>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot, (uint64_t)0, fold);
    else if (iota && shift == bit_lo)
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<MIX, true, true>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot, (uint64_t)0, fold);
    else
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<MIX, true, false>), grid, block, 0, st, kin, vin, kout, vout, n,
                         ntiles, shift, ghist, rtot, (uint64_t)0, fold);
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
                                int32_t nbatch, int32_t bit_lo, int32_t bit_hi, uint32_t flags, uint64_t aux,
                                void *workspace, size_t workspace_bytes, void *stream) {
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
    return sort_passes<SM_MIX>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_hi, iota, 0, ghist, st);
  if (flags & QRLSH_SORT_FOLD) {
    QR_CHECK_ARG(aux >= 1 && aux <= 32, "qrlsh_sort_u64: fold width %llu not in [1,32]", (unsigned long long)aux);
    return sort_passes<SM_FOLD>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_hi, iota, (uint32_t)aux, ghist, st);
  }
  if (flags & QRLSH_SORT_OWNER) {
    QR_CHECK_ARG(aux >= 1 && aux < (1ull << 32), "qrlsh_sort_u64: owner shard size %llu not in [1, 2^32)",
                 (unsigned long long)aux);
    // one pass: the digit is the owner rank of the id field that starts at bit_lo
    return sort_passes<SM_OWNER>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_lo + 1, iota, (uint32_t)aux, ghist,
                                 st);
  }
  if (flags & QRLSH_SORT_HOST) {
    QR_CHECK_ARG(aux >= 1 && aux < (1ull << 32), "qrlsh_sort_u64: host shard size %llu not in [1, 2^32)",
                 (unsigned long long)aux);
    return sort_passes<SM_HOST>(keys_a, keys_b, vals_a, vals_b, n, nbatch, 0, 1, iota, (uint32_t)aux, ghist, st);
  }
  return sort_passes<SM_PLAIN>(keys_a, keys_b, vals_a, vals_b, n, nbatch, bit_lo, bit_hi, iota, 0, ghist, st);
}

// bounds_out[g] = first position whose owner (word >> lo) / shard is >= g, g = 0 .. world, for
// words already grouped by owner (QRLSH_SORT_OWNER): the split points of the variable all-to-all.
// lo < 0: the words are pairs grouped with QRLSH_SORT_HOST, owner = qr_pair_host
__global__ void owner_bounds_kernel(const uint64_t *__restrict__ w, int64_t n, int lo, uint64_t shard, int world,
                                    int64_t *__restrict__ bounds_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > world) return;
  int64_t a = 0, b = n;
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    const uint64_t o = lo < 0 ? qr_pair_host(w[mid], (uint32_t)shard) : (w[mid] >> lo) / shard;
    if (o >= (uint64_t)g) b = mid;
    else a = mid + 1;
  }
  bounds_out[g] = a;
}

QRLSH_EXPORT int qrlsh_owner_bounds(const uint64_t *words, int64_t n, int32_t bit_lo, uint64_t shard, int32_t world,
                                    int64_t *bounds_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && bit_lo >= -1 && bit_lo < 64 && shard >= 1 && shard < (1ull << 32) && world >= 1 &&
                   world <= RADIX && bounds_out,
               "qrlsh_owner_bounds: bad arguments");
  QR_CHECK_ARG(n == 0 || words, "qrlsh_owner_bounds: null pointer");
  QR_LAUNCH("owner_bounds", owner_bounds_kernel, dim3((world + 1 + 63) / 64), dim3(64), 0,
            static_cast<hipStream_t>(stream), words, n, bit_lo, shard, world, bounds_out);
  QR_LAUNCH_CHECK("qrlsh_owner_bounds");
  return QRLSH_OK;
}

// ==========================================================================================
// Fast bucket path (a2/a3): a T-bit hash PARTITION + an LDS finish, instead of a full sort.
//
//   partition : one or two of the radix passes above on the top T bits of mix64(key)
//               (T = 8 .. 16, chosen by the host so that a part holds ~2-4 K records)
//               -> per band 2^T parts, contiguous in HBM, ids ascending inside a part;
//   bounds    : one thread per (band, part) binary-searches the part's first record;
//   finish    : one 1024-thread workgroup per (part, band) stages the part's keys in LDS, links
//               equal keys through an LDS hash table (atomicExch chains) and, per record,
//               pairs it with every EARLIER record of the part that has the same FULL key --
//               exactly the (i < j) pairs of that bucket (lsh.py:47-49).
//
// Count-then-fill like the general path; a part larger than FIN_CAP records (heavily skewed
// data) raises the overflow word and the host falls back to the general sort path.
// ==========================================================================================
#ifndef QR_FIN_THREADS
#define QR_FIN_THREADS 1024
#endif
#ifndef QR_FIN_CAP
#define QR_FIN_CAP 6144
#endif
constexpr int FIN_THREADS = QR_FIN_THREADS;
constexpr int FIN_CAP = QR_FIN_CAP;   // records per part that fit the LDS image
constexpr int FIN_IPT = FIN_CAP / FIN_THREADS;
constexpr int FIN_SMALL_THREADS = 512, FIN_SMALL_CAP = 4096;  // the small-part form of the one-pass finish
constexpr int FIN_SMALL_MEAN = 2800;                          // mean records per part up to which it is used

// starts[band][f] = first position of band `band` whose T-bit part number is >= f (f = 0 .. 2^T)
__global__ __launch_bounds__(256) void bucket_bounds_kernel(const uint64_t *__restrict__ keys,
                                                            const uint32_t *__restrict__ ids, int64_t nq, int T,
                                                            uint64_t ek, uint32_t *__restrict__ starts) {
  const int nparts = 1 << T;
  const int f = blockIdx.x * blockDim.x + threadIdx.x, band = blockIdx.y;
  if (f > nparts) return;
  const uint64_t *k = keys + (size_t)band * nq;
  const uint32_t *id = ids + (size_t)band * nq;
  int64_t lo = 0, hi = nq;  // first t with part(t) >= f
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const uint64_t key = k[mid];
    const uint32_t part = part_digit<true>(key, key == ek ? (int64_t)id[mid] : 0, 64 - T, ek, (uint32_t)nparts - 1u);
    if ((int)part >= f) hi = mid;
    else lo = mid + 1;
  }
  starts[(size_t)band * (nparts + 1) + f] = (uint32_t)lo;
}

// Finish of one (part, band): an open-addressing hash table in LDS keyed by the FULL 64-bit key
// (ds_cmpst_b64 claims a slot or finds the key present; the empty-band key, which never enters a
// bucket, doubles as the "free slot" marker).  A record's arrival number o in its slot's counter
// says how many records of its bucket came before it, so
//     pairs of the part = sum of o          (count kernel: that is all it needs)
// and, for the fill, the ids of every bucket are laid out next to each other in LDS (run start =
// exclusive scan of the slot counters, place = o) and a record pairs with the o ids in front of
// it in its run, ordered (smaller id, larger id).  No chains, no walks, no key re-compares.
// LDS: table 48 KB + counters 24 KB = 72 KB -> two 1024-thread workgroups per CU (which also
// needs <= 64 VGPRs: __launch_bounds__(1024, 8)).  Arrival order varies from run to run, so the
// pairs of a part come out in varying order -- as a set they are exact, and the next step sorts.
template <int CAP = FIN_CAP> __device__ static inline uint32_t fin_home(uint64_t key) {
  uint32_t h = (uint32_t)key * 0x9E3779B1u;
  h ^= h >> 15;
  h += (uint32_t)(key >> 32) * 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  return __umulhi(h, (uint32_t)CAP);
}

// The pairs of one record with the `c` ids at run[0 .. c) go to dst[pos .. pos + c).  A short run is written by the
// record's own lane; a long one (a popular key: hundreds to thousands of bucket-mates) by the whole wave, lane t
// writing pair t, t + 64, ... -- consecutive lanes write consecutive words instead of each lane walking thousands
// of words a long stride apart.  Called by every lane of the wave (c = 0 for lanes without a record).
// Threshold, 10 M queries x 32 bands (same box): never cooperative 2.88 ms for the finish, 48 -> 2.56, 12 -> 2.52.
#ifndef QR_FIN_COOP
#define QR_FIN_COOP 16
#endif
constexpr uint32_t FIN_COOP = QR_FIN_COOP;
__device__ static inline void emit_run(uint64_t *__restrict__ dst, uint32_t pos, uint32_t me, const uint32_t *run,
                                       uint32_t run_off, uint32_t c) {
  if (c <= FIN_COOP) {
    for (uint32_t t = 0; t < c; ++t) {
      const uint32_t other = run[run_off + t];
      dst[pos + t] = (uint64_t)min(me, other) << 32 | max(me, other);
    }
  }
  uint64_t big = __ballot(c > FIN_COOP);
  const int lane = threadIdx.x & (WAVE - 1);
  while (big) {  // uniform
    const int L = __ffsll((long long)big) - 1;
    big &= big - 1;
    const uint32_t me_b = __shfl(me, L, WAVE), c_b = __shfl(c, L, WAVE), off_b = __shfl(run_off, L, WAVE),
                   pos_b = __shfl(pos, L, WAVE);
    for (uint32_t t = lane; t < c_b; t += WAVE) {
      const uint32_t other = run[off_b + t];
      dst[pos_b + t] = (uint64_t)min(me_b, other) << 32 | max(me_b, other);
    }
  }
}

// MODE: FIN_COUNT leaves the part's pair count in blk; FIN_FILL writes the pairs at the offset the
// scanned blk holds; FIN_EMIT does both in one go -- the workgroup reserves its output range with
// one atomicAdd on a global cursor (blk[0]) and writes only if the range fits `capacity`; the cursor
// ends up holding the exact total either way, so a caller whose guess was too small retries once.
enum { FIN_COUNT = 0, FIN_FILL = 1, FIN_EMIT = 2 };
// THREADS / CAP: the workgroup and its LDS image.  1024 / 6144 (72 KB: two workgroups per CU) is the general form; when
// the parts are small (mean <= FIN_SMALL_MEAN records: 10 M queries and beyond), 512 / 4096 (48 KB) puts THREE
// independent chains of phases on a CU instead of two (2.51 -> 2.12 ms at 10 M), and a part between 4096 and 6144
// records joins the ones the big kernel works in blocks.
template <int MODE, int THREADS = FIN_THREADS, int CAP = FIN_CAP>
__global__ __launch_bounds__(THREADS, THREADS >= 1024 ? 8 : 6) void bucket_finish_kernel(const uint64_t *__restrict__ keys,
                                                                       const uint32_t *__restrict__ ids, int64_t nq,
                                                                       const uint32_t *__restrict__ starts,
                                                                       int nparts, uint64_t ek,
                                                                       uint64_t *__restrict__ blk,
                                                                       uint32_t *__restrict__ overflow,
                                                                       uint64_t *__restrict__ out,
                                                                       uint64_t capacity,
                                                                       const uint32_t *__restrict__ counts = nullptr,
                                                                       uint32_t cap = 0,
                                                                       uint64_t *__restrict__ biglist = nullptr,
                                                                       unsigned long long *__restrict__ nbig = nullptr,
                                                                       uint32_t big_max = 0, uint32_t big_base = 0) {
  constexpr bool FILL = MODE != FIN_COUNT;
  constexpr int IPT = CAP / THREADS;
  static_assert(CAP % THREADS == 0 && CAP <= 65535, "image = whole records per thread, slots fit 16 bits");
  __shared__ unsigned long long gbase;
  __shared__ __attribute__((aligned(16))) unsigned long long tab[CAP];
  __shared__ uint32_t cnt[CAP];
  __shared__ uint32_t wsum[THREADS / WAVE];
  const int part = blockIdx.x, band = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  const size_t bslot = (size_t)band * nparts + part;
  // records of the part: [start, start + m) of the band's nq sorted records, or (counts given) the first
  // counts[part] records of the part's own region of cap records
  const uint32_t start = counts ? 0u : starts[(size_t)band * (nparts + 1) + part];
  const uint32_t m = counts ? counts[bslot] : starts[(size_t)band * (nparts + 1) + part + 1] - start;
  const size_t first = counts ? bslot * cap : (size_t)band * nq + start;
  if (m > (uint32_t)CAP || (counts && m > cap)) {  // uniform over the workgroup
    if (tid == 0) {
      // a part that holds more records than the LDS image (a popular key with thousands of copies, mostly), in its
      // region or spilled into the pool: left to bucket_finish_big_kernel, which works it in blocks -- one-pass form only
      bool listed = false;
      if (MODE == FIN_EMIT && biglist && counts) {
        const unsigned long long at =
            __hip_atomic_fetch_add(nbig, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at < (unsigned long long)big_max) {
          biglist[at] = (uint64_t)(big_base + bslot);
          listed = true;
        }
      }
      if (!listed) atomicOr(overflow, 1u);
      if (MODE == FIN_COUNT) blk[bslot] = 0;
    }
    return;
  }
  if (m == 0) {
    if (MODE == FIN_COUNT && tid == 0) blk[bslot] = 0;
    return;
  }
  const uint64_t *k = keys + first;
  const uint32_t *id = ids + first;
  uint64_t kreg[IPT];
  uint32_t ireg[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const uint32_t i = tid + j * THREADS;
    kreg[j] = i < m ? k[i] : ek;
    if (FILL) ireg[j] = i < m ? id[i] : 0u;  // coalesced, in flight together with the keys
  }
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const uint32_t i = tid + j * THREADS;
    tab[i] = ek;
    cnt[i] = 0;
  }
  __syncthreads();
  uint32_t so[IPT];  // arrival number << 16 | slot (both < FIN_CAP <= 65535); 0xFFFFFFFF = empty band
  uint32_t mine = 0;
  // first probes of all IPT records go out together (independent LDS atomics in flight), the
  // occasional second and later probes follow per record, then all the counter increments together
  uint32_t slot[IPT];
  unsigned long long seen[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    slot[j] = fin_home<CAP>(kreg[j]);
    seen[j] = kreg[j] != ek
                  ? atomicCAS(&tab[slot[j]], (unsigned long long)ek, (unsigned long long)kreg[j])
                  : (unsigned long long)ek;
  }
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    if (kreg[j] != ek) {
      unsigned long long old = seen[j];
      while (old != ek && old != kreg[j]) {  // FIN_CAP slots for at most FIN_CAP records: a free one always turns up
        slot[j] = slot[j] + 1 == (uint32_t)CAP ? 0u : slot[j] + 1;
        old = atomicCAS(&tab[slot[j]], (unsigned long long)ek, (unsigned long long)kreg[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    so[j] = 0xFFFFFFFFu;
    if (kreg[j] != ek) {
      const uint32_t o = atomicAdd(&cnt[slot[j]], 1u);
      so[j] = o << 16 | slot[j];
      mine += o;
    }
  }
  // block exclusive scan over 1024 threads (a part emits < 2^32 pairs: FIN_CAP^2 / 2)
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += o;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();  // also: every insert is over
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / WAVE; ++i) {
    const uint32_t x = wsum[i];
    if (i < w) base += x;
    tot += x;
  }
  if (MODE == FIN_COUNT) {
    if (tid == 0) blk[bslot] = (uint64_t)tot;
    return;
  }
  if (MODE == FIN_EMIT) {
    if (tot == 0) return;  // uniform
    if (tid == 0)
      gbase = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(blk), (unsigned long long)tot,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint32_t pos0 = base + inc - mine;
  {
    // run starts: exclusive scan of the slot counters, blocked layout (IPT consecutive slots per thread)
    const uint32_t b0 = tid * IPT;
    uint32_t v[IPT], sum = 0;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      v[q] = cnt[b0 + q];
      sum += v[q];
    }
    uint32_t sinc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(sinc, d, WAVE);
      if (lane >= d) sinc += o;
    }
    __syncthreads();  // wsum is read by everyone above
    if (lane == WAVE - 1) wsum[w] = sinc;
    __syncthreads();
    uint32_t run = sinc - sum;
#pragma unroll
    for (int i = 0; i < THREADS / WAVE; ++i)
      if (i < w) run += wsum[i];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      cnt[b0 + q] = run;
      run += v[q];
    }
  }
  __syncthreads();
  uint32_t *grp = reinterpret_cast<uint32_t *>(tab);  // the table is dead: ids, bucket by bucket
#pragma unroll
  for (int j = 0; j < IPT; ++j)
    if (so[j] != 0xFFFFFFFFu) grp[cnt[so[j] & 0xFFFFu] + (so[j] >> 16)] = ireg[j];
  __syncthreads();
  {
    const uint64_t obase = MODE == FIN_EMIT ? (uint64_t)gbase : blk[bslot];
    if (MODE == FIN_EMIT && obase + tot > capacity) return;  // uniform: counted, not written
    uint64_t *dst = out + obase;
    uint32_t pos = pos0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const bool rec = so[j] != 0xFFFFFFFFu;
      const uint32_t o = rec ? so[j] >> 16 : 0u;
      emit_run(dst, pos, ireg[j], grp, rec ? cnt[so[j] & 0xFFFFu] : 0u, o);
      pos += o;
    }
  }
}

// The small-part finish with the arrival counters INSIDE the table words (round 4; partitions of 12 bits and more: 10 M
// queries and up).  Inside a part all words share their top T >= 12 bits (the part number), so a slot needs only the
// low 52 bits of the word to tell keys apart, and the 12 bits above hold the number of records that found it: a slot is
// claimed with one CAS (0 -> key52 | 1 << 52: arrival number 0), joined with one returning 64-bit add of 1 << 52 (the old
// count is the arrival number).  No counter array: the image is 32 KB instead of 48 -- FOUR workgroups per CU instead of
// three (the kernel is a chain of barrier-separated phases; what hides one workgroup's latency is another workgroup).
// After the inserts the table is read once (counts -> run starts) and its space re-used: run starts in the lower half,
// the ids laid out by bucket in the upper half.  Parts of more than 4095 records (the 12-bit count) go to the block
// kernel like every part beyond the image.
constexpr int FIN_PK_THREADS = 512, FIN_PK_CAP = 4096, FIN_PK_MAX = 4095;
__global__ __launch_bounds__(FIN_PK_THREADS, 8) void bucket_finish_packed_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ ids, int nparts, uint64_t *__restrict__ blk,
    uint32_t *__restrict__ overflow, uint64_t *__restrict__ out, uint64_t capacity, const uint32_t *__restrict__ counts,
    uint32_t cap, uint64_t *__restrict__ biglist, unsigned long long *__restrict__ nbig, uint32_t big_max,
    uint32_t big_base) {
  constexpr int THREADS = FIN_PK_THREADS, CAP = FIN_PK_CAP, IPT = CAP / THREADS;
  constexpr unsigned long long KMASK = (1ull << 52) - 1ull, ONE = 1ull << 52;
  __shared__ unsigned long long gbase;
  __shared__ __attribute__((aligned(16))) unsigned long long tab[CAP];
  __shared__ uint32_t wsum[THREADS / WAVE];
  const int part = blockIdx.x, band = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  const size_t bslot = (size_t)band * nparts + part;
  const uint32_t m = counts[bslot];
  if (m > (uint32_t)FIN_PK_MAX) {  // uniform: left to the block kernel (in its region, or spilled into the pool)
    if (tid == 0) {
      bool listed = false;
      const unsigned long long at = __hip_atomic_fetch_add(nbig, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (at < (unsigned long long)big_max) {
        biglist[at] = (uint64_t)(big_base + bslot);
        listed = true;
      }
      if (!listed) atomicOr(overflow, 1u);
    }
    return;
  }
  if (m == 0) return;
  const size_t first = bslot * cap;
  const uint64_t *k = keys + first;
  const uint32_t *id = ids + first;
  uint64_t kreg[IPT];
  uint32_t ireg[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const uint32_t i = tid + j * THREADS;
    kreg[j] = i < m ? k[i] : 0ull;
    ireg[j] = i < m ? id[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < IPT; ++j) tab[tid + j * THREADS] = 0ull;
  __syncthreads();
  uint32_t so[IPT];  // arrival number << 16 | slot; 0xFFFFFFFF = no record
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    so[j] = 0xFFFFFFFFu;
    if (tid + j * THREADS < (int)m) {
      const unsigned long long k52 = kreg[j] & KMASK;
      uint32_t slot = fin_home<CAP>(kreg[j]);
      uint32_t o;
      for (;;) {  // at most 4095 records for 4096 slots: a free one always turns up
        const unsigned long long old = atomicCAS(&tab[slot], 0ull, k52 | ONE);
        if (old == 0ull) {
          o = 0;
          break;
        }
        if ((old & KMASK) == k52) {
          o = (uint32_t)(atomicAdd(&tab[slot], ONE) >> 52);
          break;
        }
        slot = slot + 1 == (uint32_t)CAP ? 0u : slot + 1;
      }
      so[j] = o << 16 | slot;
      mine += o;
    }
  }
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += o;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();  // also: every insert is over
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < THREADS / WAVE; ++i) {
    const uint32_t x = wsum[i];
    if (i < w) base += x;
    tot += x;
  }
  if (tot == 0) return;  // uniform
  if (tid == 0)
    gbase = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(blk), (unsigned long long)tot, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t pos0 = base + inc - mine;
  uint32_t *rs = reinterpret_cast<uint32_t *>(tab);        // run starts: lower half of the table's space ...
  uint32_t *grp = rs + CAP;                                // ... ids by bucket: upper half
  {
    const uint32_t b0 = tid * IPT;
    uint32_t v[IPT], sum = 0;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      v[q] = (uint32_t)(tab[b0 + q] >> 52);
      sum += v[q];
    }
    uint32_t sinc = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(sinc, d, WAVE);
      if (lane >= d) sinc += o;
    }
    __syncthreads();  // wsum was read by everyone above; every table word has been read
    if (lane == WAVE - 1) wsum[w] = sinc;
    __syncthreads();
    uint32_t run = sinc - sum;
#pragma unroll
    for (int i = 0; i < THREADS / WAVE; ++i)
      if (i < w) run += wsum[i];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      rs[b0 + q] = run;
      run += v[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IPT; ++j)
    if (so[j] != 0xFFFFFFFFu) grp[rs[so[j] & 0xFFFFu] + (so[j] >> 16)] = ireg[j];
  __syncthreads();
  {
    const uint64_t obase = (uint64_t)gbase;
    if (obase + tot > capacity) return;  // uniform: counted, not written
    uint64_t *dst = out + obase;
    uint32_t pos = pos0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const bool rec = so[j] != 0xFFFFFFFFu;
      const uint32_t o = rec ? so[j] >> 16 : 0u;
      emit_run(dst, pos, ireg[j], grp, rec ? rs[so[j] & 0xFFFFu] : 0u, o);
      pos += o;
    }
  }
}

// Parts the kernel above listed (more records than its LDS image): worked in BLOCKS of
// FIN_CAP records by workgroups that walk the device-side list (fixed grid; nothing is read back to size the
// launch).  Block bi is finished exactly like a small part (hash table on the full word, arrival numbers, bucket
// runs laid out in LDS -> its own pairs); then every EARLIER block's records are streamed past bi's table: a
// record whose word is in the table pairs with every id of that bucket's run.  Together: every pair of equal words
// of the part, once -- a key with any number of copies up to FIN_BIG_BLOCKS images is no special case any more.
// Output ranges are reserved on the same device cursor as the small parts'.
constexpr int FIN_BIG_GRID = 256;
constexpr int FIN_BIG_BLOCKS = 16;        // blocks of FIN_CAP records a listed part may hold (98 304); beyond: overflow flag
constexpr uint32_t FIN_BIG_LIST = 4096;   // listed parts per band group and call; beyond: overflow flag (general path)
constexpr int FIN_BIG_SLICES = 8;         // workgroups that share a block pair's pairs (a power of two)
constexpr uint32_t POOL_RUNS = 1u << 20;  // spilled-run descriptors per call; beyond: overflow flag

// where the records of a listed part sit: `pool` = 0: in the part buffers at `where` (its own region), 1: in the
// pool at `where` (gathered there by the kernel below); m = 0: nothing to do (the overflow flag is up)
struct BigDesc {
  uint64_t where;
  uint32_t m, pool;
};

// One workgroup per listed part: a part that spilled (fill mark set) gets m records of room at the pool's cursor and
// its records -- the prefix its region holds and every run of the descriptor list that names it -- are copied there,
// next to each other in any order (the finish does not care); a part that fits its region is described in place.
__global__ __launch_bounds__(256) void bucket_big_gather_kernel(const uint64_t *__restrict__ biglist,
                                                                const unsigned long long *__restrict__ nbig, uint32_t big_max,
                                                                BigDesc *__restrict__ desc, const uint64_t *__restrict__ part_keys,
                                                                const uint32_t *__restrict__ part_ids,
                                                                const uint32_t *__restrict__ counts, uint32_t cap, PartPool pool,
                                                                uint32_t max_records, uint32_t *__restrict__ overflow) {
  __shared__ unsigned long long base_s;
  __shared__ uint32_t match[256];
  __shared__ uint32_t nmatch;
  const int tid = threadIdx.x;
  unsigned long long nb = *nbig;
  if (nb > big_max) nb = big_max;
  for (unsigned long long e = blockIdx.x; e < nb; e += gridDim.x) {
    const uint32_t slot = (uint32_t)biglist[e];
    const uint32_t m = counts[slot];
    const uint32_t f = pool.fill ? pool.fill[slot] : 0xFFFFFFFFu;
    if (m > max_records || (f == 0xFFFFFFFFu && m > cap)) {  // (uniform) too large / records were dropped
      if (tid == 0) {
        atomicOr(overflow, 1u);
        desc[e] = BigDesc{0ull, 0u, 0u};
      }
      continue;
    }
    if (f == 0xFFFFFFFFu) {  // all in its region
      if (tid == 0) desc[e] = BigDesc{(uint64_t)slot * cap, m, 0u};
      continue;
    }
    __syncthreads();  // (the previous part's base_s has been read by everyone)
    if (tid == 0)
      base_s = __hip_atomic_fetch_add(pool.cursor, (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long base = base_s;
    if (base + m > (unsigned long long)pool.cap) {  // uniform
      if (tid == 0) {
        atomicOr(overflow, 1u);
        desc[e] = BigDesc{0ull, 0u, 0u};
      }
      continue;
    }
    for (uint32_t i = tid; i < f; i += blockDim.x) {
      pool.keys[base + i] = part_keys[(size_t)slot * cap + i];
      pool.vals[base + i] = part_ids[(size_t)slot * cap + i];
    }
    uint32_t at = f;
    unsigned long long nr = *pool.nruns;
    if (nr > pool.runs_max) nr = pool.runs_max;
    for (unsigned long long r0 = 0; r0 < nr; r0 += blockDim.x) {
      __syncthreads();
      if (tid == 0) nmatch = 0;
      __syncthreads();
      const unsigned long long r = r0 + tid;
      if (r < nr && pool.runs[r].x == slot) match[atomicAdd(&nmatch, 1u)] = (uint32_t)r;
      __syncthreads();
      const uint32_t nm = nmatch;
      for (uint32_t q = 0; q < nm; ++q) {
        const uint4 run = pool.runs[match[q]];
        if (at + run.y <= m)
          for (uint32_t i = tid; i < run.y; i += blockDim.x) {
            pool.keys[base + at + i] = pool.keys[(size_t)run.z + i];
            pool.vals[base + at + i] = pool.vals[(size_t)run.z + i];
          }
        at += run.y;
      }
    }
    if (tid == 0) {
      if (at != m) atomicOr(overflow, 1u);  // (cannot happen: every record of the part is in its region or in a run)
      desc[e] = BigDesc{(uint64_t)base, at == m ? m : 0u, 1u};
    }
  }
}

__global__ __launch_bounds__(FIN_THREADS) void bucket_finish_big_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ ids, const uint64_t *pool_keys, const uint32_t *pool_ids,
    uint64_t ek, const BigDesc *__restrict__ desc, const unsigned long long *__restrict__ nbig, uint32_t big_max,
    uint64_t *__restrict__ blk, uint64_t *__restrict__ out, uint64_t capacity) {
  __shared__ unsigned long long gbase;
  __shared__ __attribute__((aligned(16))) unsigned long long tab[FIN_CAP];
  __shared__ uint32_t cnt[FIN_CAP + 1];
  __shared__ uint32_t grp[FIN_CAP];
  __shared__ uint32_t srt[FIN_CAP];   // the runs in id order (own pairs of a block)
  __shared__ uint32_t wsum[FIN_THREADS / WAVE];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  unsigned long long nb = *nbig;
  if (nb > big_max) nb = big_max;
  // block exclusive scan of one u32 per thread -> (exclusive prefix, total); two barriers
  auto scan = [&](uint32_t v, uint32_t &total) -> uint32_t {
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, WAVE);
      if (lane >= d) inc += o;
    }
    __syncthreads();  // wsum may still be read from the previous scan
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < FIN_THREADS / WAVE; ++i) {
      const uint32_t x = wsum[i];
      if (i < w) base += x;
      tot += x;
    }
    total = tot;
    return base + inc - v;
  };
  // work items: (listed part, block bi, block bj <= bi, slice sl) -- the table of bi is built, then an eighth of bi's
  // own pairs (bj = bi) or of the pairs of bj's records with it is written; a part's items run on different
  // workgroups, so that a key with thousands of copies (1.4 M pairs for 1 700 copies: 0.35 ms when one workgroup
  // writes them all) is the work of eight.  Every workgroup builds the table itself and the arrival numbers differ
  // from build to build, so a slice is defined on something the builds share: bi's own pairs go by the RANK of a
  // record's query id among its bucket-mates (the runs are sorted by id; a record pairs with the mates of smaller
  // id, and belongs to slice rank mod 8), bj's records by their position in bj (mod 8).
  constexpr int COMBOS = FIN_BIG_BLOCKS * (FIN_BIG_BLOCKS + 1) / 2;
  static_assert(FIN_THREADS % FIN_BIG_SLICES == 0, "a thread's records share their position mod the slice count");
  for (unsigned long long item = blockIdx.x; item < nb * COMBOS * FIN_BIG_SLICES; item += gridDim.x) {
    const uint32_t sl = (uint32_t)(item % FIN_BIG_SLICES);
    const unsigned long long e = item / ((unsigned long long)COMBOS * FIN_BIG_SLICES);
    int combo = (int)((item / FIN_BIG_SLICES) % COMBOS);
    uint32_t bi = 0;
    while (combo > (int)bi) {  // combos in the order (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) ...
      combo -= (int)bi + 1;
      ++bi;
    }
    const uint32_t bj_only = (uint32_t)combo;
    const BigDesc de = desc[e];
    const uint32_t m = de.m;
    const uint64_t *k = (de.pool ? pool_keys : keys) + de.where;
    const uint32_t *id = (de.pool ? pool_ids : ids) + de.where;
    const uint32_t nblk = (m + FIN_CAP - 1) / FIN_CAP;
    if (bi >= nblk) continue;  // uniform
    {
      const uint32_t lo = bi * FIN_CAP, mb = min((uint32_t)FIN_CAP, m - lo);
      uint64_t kreg[FIN_IPT];
      uint32_t ireg[FIN_IPT], so[FIN_IPT];
      __syncthreads();  // the previous block / part is done with the arrays
#pragma unroll
      for (int j = 0; j < FIN_IPT; ++j) {
        const uint32_t i = tid + j * FIN_THREADS;
        kreg[j] = i < mb ? k[lo + i] : ek;
        ireg[j] = i < mb ? id[lo + i] : 0u;
        tab[i] = ek;
        cnt[i] = 0;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < FIN_IPT; ++j) {
        so[j] = 0xFFFFFFFFu;
        if (kreg[j] != ek) {
          uint32_t slot = fin_home(kreg[j]);
          for (;;) {  // at most FIN_CAP records for FIN_CAP slots: a free one always turns up
            const unsigned long long old = atomicCAS(&tab[slot], (unsigned long long)ek, (unsigned long long)kreg[j]);
            if (old == ek || old == kreg[j]) break;
            slot = slot + 1 == (uint32_t)FIN_CAP ? 0u : slot + 1;
          }
          const uint32_t o = atomicAdd(&cnt[slot], 1u);
          so[j] = o << 16 | slot;
        }
      }
      __syncthreads();  // every insert is over
      {
        // run starts: exclusive scan of the slot counters, blocked layout; cnt[FIN_CAP] = the block's record count
        const uint32_t b0 = tid * FIN_IPT;
        uint32_t v[FIN_IPT], sum = 0;
#pragma unroll
        for (int q = 0; q < FIN_IPT; ++q) {
          v[q] = cnt[b0 + q];
          sum += v[q];
        }
        uint32_t all;
        uint32_t run = scan(sum, all);
#pragma unroll
        for (int q = 0; q < FIN_IPT; ++q) {
          cnt[b0 + q] = run;
          run += v[q];
        }
        if (tid == 0) cnt[FIN_CAP] = all;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < FIN_IPT; ++j)
        if (so[j] != 0xFFFFFFFFu) grp[cnt[so[j] & 0xFFFFu] + (so[j] >> 16)] = ireg[j];
      __syncthreads();  // the runs are laid out (in this build's arrival order)
      if (bj_only == bi) {
        // this block's own pairs.  Rank of every record's id among its run (ids are distinct inside a bucket), the runs
        // re-laid in id order, then a record of rank r pairs with the r mates in front of it -- if r mod 8 is this slice
        uint32_t rk[FIN_IPT];
#pragma unroll
        for (int j = 0; j < FIN_IPT; ++j) {
          rk[j] = 0;
          if (so[j] != 0xFFFFFFFFu) {
            // (a lane per record: the copies of a popular key fill whole waves, which then walk the run in step --
            // counting a long run with the whole wave, one record after the other, was 2.5x slower)
            const uint32_t sl0 = so[j] & 0xFFFFu, s0 = cnt[sl0], c = cnt[sl0 + 1] - s0, me = ireg[j];
            uint32_t r = 0;
            for (uint32_t u = 0; u < c; ++u) r += grp[s0 + u] < me;
            rk[j] = r;
          }
        }
        __syncthreads();  // every rank is known: the runs may move
#pragma unroll
        for (int j = 0; j < FIN_IPT; ++j)
          if (so[j] != 0xFFFFFFFFu) srt[cnt[so[j] & 0xFFFFu] + rk[j]] = ireg[j];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < FIN_IPT; ++j) {
          if (so[j] == 0xFFFFFFFFu || (rk[j] & (FIN_BIG_SLICES - 1)) != sl) rk[j] = 0;  // not this slice's: no pairs
          mine += rk[j];
        }
        uint32_t tot;
        const uint32_t pos0 = scan(mine, tot);  // (its barriers also end the re-layout)
        if (tid == 0 && tot)
          gbase = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(blk), (unsigned long long)tot,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tot && gbase + tot <= capacity) {  // uniform; counted either way
          uint64_t *dst = out + gbase;
          uint32_t pos = pos0;
#pragma unroll
          for (int j = 0; j < FIN_IPT; ++j) {
            emit_run(dst, pos, ireg[j], srt, so[j] != 0xFFFFFFFFu ? cnt[so[j] & 0xFFFFu] : 0u, rk[j]);
            pos += rk[j];
          }
        }
      }
      // the records of an earlier block against this block's table
      if (bj_only < bi) {
        const uint32_t bj = bj_only;
        const uint32_t lo2 = bj * FIN_CAP;  // earlier blocks are full
        uint32_t hit[FIN_IPT];              // slot of the record's word in the table, 0xFFFFFFFF = absent
        uint32_t mine2 = 0;
#pragma unroll
        for (int j = 0; j < FIN_IPT; ++j) {
          const uint32_t i = tid + j * FIN_THREADS;
          const uint64_t x = k[lo2 + i];
          ireg[j] = id[lo2 + i];
          uint32_t slot = fin_home(x), found = 0xFFFFFFFFu;
          const bool my_slice = (uint32_t)(tid & (FIN_BIG_SLICES - 1)) == sl;
          for (int step = 0; my_slice && step < FIN_CAP; ++step) {  // (bounded: a full table has no free slot to stop at)
            const unsigned long long tv = tab[slot];
            if (tv == x) {
              found = slot;
              break;
            }
            if (tv == ek) break;
            slot = slot + 1 == (uint32_t)FIN_CAP ? 0u : slot + 1;
          }
          if ((uint32_t)(tid & (FIN_BIG_SLICES - 1)) != sl) found = 0xFFFFFFFFu;  // another slice's record of bj
          hit[j] = found;
          if (found != 0xFFFFFFFFu) mine2 += cnt[found + 1] - cnt[found];
        }
        uint32_t tot2;
        const uint32_t p0 = scan(mine2, tot2);
        if (tid == 0 && tot2)
          gbase = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(blk), (unsigned long long)tot2,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tot2 && gbase + tot2 <= capacity) {  // uniform
          uint64_t *dst = out + gbase;
          uint32_t pos = p0;
#pragma unroll
          for (int j = 0; j < FIN_IPT; ++j) {
            const bool h = hit[j] != 0xFFFFFFFFu;
            const uint32_t s0 = h ? cnt[hit[j]] : 0u, c = h ? cnt[hit[j] + 1] - s0 : 0u;
            emit_run(dst, pos, ireg[j], grp, s0, c);
            pos += c;
          }
        }
        __syncthreads();  // gbase is rewritten by the next reservation
      }
    }
  }
}

// the packed-counter form of the small-part finish (QRLSH_FIN_PACKED=0: the separate-counter form; an A/B knob)
static int g_fin_packed = [] {
  const char *e = getenv("QRLSH_FIN_PACKED");
  return !(e && e[0] == '0') ? 1 : 0;
}();

// records a listed part may hold (qrlsh_set_big_part_limit; default and maximum: FIN_BIG_BLOCKS images)
static uint32_t g_big_limit = (uint32_t)FIN_BIG_BLOCKS * FIN_CAP;
QRLSH_EXPORT int64_t qrlsh_set_big_part_limit(int64_t records) {
  const int64_t max = (int64_t)FIN_BIG_BLOCKS * FIN_CAP, old = g_big_limit;
  g_big_limit = (uint32_t)(records <= 0 || records > max ? max : records);
  return old;
}

// workspace: [ghist + rtot of one sort pass][starts: b*(2^T+1) u32][blk: b*2^T u64][16 B tail]
//            [fill: b*2^T u32][pool cursor, run count: 2 u64][runs: POOL_RUNS x 16 B][desc: 16 x FIN_BIG_LIST x 16 B]
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
constexpr int EMIT_MAX_GROUPS = 16;  // band groups of one qrlsh_bucket_pairs_emit call
struct BucketWs {
  uint32_t *ghist, *rtot, *starts;
  uint64_t *blk;
  uint32_t *tail;
  uint32_t *fill;
  unsigned long long *poolctl;
  uint4 *runs;
  BigDesc *desc;
  size_t bytes;
};
static BucketWs bucket_ws(void *workspace, int64_t nq, int32_t b, int32_t T) {
  BucketWs w;
  const size_t nparts = (size_t)1 << T;
  const int64_t ntiles = ceil_div64(nq, SORT_TILE);
  char *p = static_cast<char *>(workspace);
  size_t off = 0;
  w.ghist = reinterpret_cast<uint32_t *>(p + off);
  w.rtot = w.ghist + (size_t)b * RADIX * ntiles;
  off += align16(qrlsh_sort_workspace_bytes(nq, b));
  w.starts = reinterpret_cast<uint32_t *>(p + off);
  off += align16((size_t)b * (nparts + 1) * sizeof(uint32_t));
  w.blk = reinterpret_cast<uint64_t *>(p + off);
  off += (size_t)b * nparts * sizeof(uint64_t);
  w.tail = reinterpret_cast<uint32_t *>(p + off);
  off += 16;
  w.fill = reinterpret_cast<uint32_t *>(p + off);
  off += align16((size_t)b * nparts * sizeof(uint32_t));
  w.poolctl = reinterpret_cast<unsigned long long *>(p + off);
  off += 16;
  w.runs = reinterpret_cast<uint4 *>(p + off);
  off += (size_t)POOL_RUNS * sizeof(uint4);
  w.desc = reinterpret_cast<BigDesc *>(p + off);
  off += (size_t)EMIT_MAX_GROUPS * FIN_BIG_LIST * sizeof(BigDesc);
  w.bytes = off;
  return w;
}

QRLSH_EXPORT size_t qrlsh_bucket_workspace_bytes(int64_t nq, int32_t b, int32_t part_bits) {
  if (nq <= 0 || b <= 0 || part_bits < 8 || part_bits > 16) return 64;
  return bucket_ws(nullptr, nq, b, part_bits).bytes;
}

// partition (one or two radix passes on the T-bit part number) + part bounds
static void bucket_partition(const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids, uint64_t *tmp_keys,
                             uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r, int32_t part_bits, const BucketWs &w,
                             hipStream_t st) {
  const int T = part_bits, nparts = 1 << T;
  const int ntiles = (int)ceil_div64(nq, SORT_TILE);
  const dim3 grid(ntiles, b), block(SORT_THREADS);
  const uint64_t ek = qr_empty_key(r);
  const uint32_t *no_vals = nullptr;
  const bool staged = nq <= (1ll << 24);  // ids fit 24 bits: the LDS-staged scatter carries the digit beside them
  if (T == 8) {
    QR_LAUNCH("sort_hist", (sort_hist_kernel<SM_MIX, true>), grid, block, 0, st, keys, nq, ntiles, 56, w.ghist, ek, 0,
              (uint32_t)RADIX - 1u, no_vals);
    launch_rowscan(w.ghist, ntiles, w.rtot, b, st);
    if (staged)
      QR_LAUNCH("sort_scatter_kv", (part_scatter_staged_kernel<true>), grid, block, 0, st, keys, no_vals, part_keys,
                part_ids, nq, ntiles, 56, w.ghist, w.rtot, ek, (uint32_t)RADIX - 1u);
    else
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<SM_MIX, true, true, true>), grid, block, 0, st, keys, no_vals,
                part_keys, part_ids, nq, ntiles, 56, w.ghist, w.rtot, ek, 0, (uint32_t)RADIX - 1u);
  } else {
    // LSD over the T-bit part number: low T-8 bits first, then the top 8
    const uint32_t lowmask = (1u << (T - 8)) - 1u;
    QR_LAUNCH("sort_hist", (sort_hist_kernel<SM_MIX, true>), grid, block, 0, st, keys, nq, ntiles, 64 - T, w.ghist, ek, 0,
              lowmask, no_vals);
    launch_rowscan(w.ghist, ntiles, w.rtot, b, st);
    if (staged)
      QR_LAUNCH("sort_scatter_kv", (part_scatter_staged_kernel<true>), grid, block, 0, st, keys, no_vals, tmp_keys,
                tmp_ids, nq, ntiles, 64 - T, w.ghist, w.rtot, ek, lowmask);
    else
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<SM_MIX, true, true, true>), grid, block, 0, st, keys, no_vals,
                tmp_keys, tmp_ids, nq, ntiles, 64 - T, w.ghist, w.rtot, ek, 0, lowmask);
    QR_LAUNCH("sort_hist", (sort_hist_kernel<SM_MIX, true>), grid, block, 0, st, (const uint64_t *)tmp_keys, nq, ntiles,
              56, w.ghist, ek, 0, (uint32_t)RADIX - 1u, (const uint32_t *)tmp_ids);
    launch_rowscan(w.ghist, ntiles, w.rtot, b, st);
    if (staged)
      QR_LAUNCH("sort_scatter_kv", (part_scatter_staged_kernel<false>), grid, block, 0, st, (const uint64_t *)tmp_keys,
                (const uint32_t *)tmp_ids, part_keys, part_ids, nq, ntiles, 56, w.ghist, w.rtot, ek,
                (uint32_t)RADIX - 1u);
    else
      QR_LAUNCH("sort_scatter_kv", (sort_scatter_kernel<SM_MIX, true, false, true>), grid, block, 0, st,
                (const uint64_t *)tmp_keys, (const uint32_t *)tmp_ids, part_keys, part_ids, nq, ntiles, 56, w.ghist,
                w.rtot, ek, 0, (uint32_t)RADIX - 1u);
  }
  QR_LAUNCH("bucket_bounds", bucket_bounds_kernel, dim3((nparts + 1 + 255) / 256, b), dim3(256), 0, st,
            (const uint64_t *)part_keys, (const uint32_t *)part_ids, nq, T, ek, w.starts);
}

static int bucket_check(const char *name, const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids,
                        uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r, int32_t part_bits,
                        void *workspace, size_t workspace_bytes, uint64_t *total_overflow_out, hipStream_t st) {
  QR_CHECK_ARG(nq >= 0 && b > 0 && b <= 65535 && r > 0, "%s: bad sizes nq=%lld b=%d r=%d", name, (long long)nq, b, r);
  QR_CHECK_ARG(part_bits >= 8 && part_bits <= 16, "%s: part_bits=%d not in [8,16]", name, part_bits);
  QR_CHECK_ARG(nq < (1ll << 32), "%s: nq too large", name);
  QR_CHECK_ARG(total_overflow_out && workspace, "%s: null pointer", name);
  if (hipMemsetAsync(total_overflow_out, 0, 2 * sizeof(uint64_t), st) != hipSuccess) {
    qrlsh_set_error("%s: hipMemsetAsync failed", name);
    return QRLSH_EHIP;
  }
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(keys && part_keys && part_ids, "%s: null pointer", name);
  QR_CHECK_ARG(part_bits == 8 || (tmp_keys && tmp_ids), "%s: part_bits > 8 needs tmp buffers", name);
  if (workspace_bytes < qrlsh_bucket_workspace_bytes(nq, b, part_bits)) {
    qrlsh_set_error("%s: workspace %zu < %zu bytes", name, workspace_bytes,
                    qrlsh_bucket_workspace_bytes(nq, b, part_bits));
    return QRLSH_EWORKSPACE;
  }
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_bucket_pairs_count(const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids,
                                          uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r,
                                          int32_t part_bits, void *workspace, size_t workspace_bytes,
                                          uint64_t *total_overflow_out, void *stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = bucket_check("qrlsh_bucket_pairs_count", keys, part_keys, part_ids, tmp_keys, tmp_ids, nq, b, r,
                              part_bits, workspace, workspace_bytes, total_overflow_out, st);
  if (rc != QRLSH_OK || nq == 0) return rc;
  const int nparts = 1 << part_bits;
  const uint64_t ek = qr_empty_key(r);
  const BucketWs w = bucket_ws(workspace, nq, b, part_bits);
  bucket_partition(keys, part_keys, part_ids, tmp_keys, tmp_ids, nq, b, r, part_bits, w, st);
  QR_LAUNCH("bucket_count", (bucket_finish_kernel<FIN_COUNT>), dim3(nparts, b), dim3(FIN_THREADS), 0, st,
            (const uint64_t *)part_keys, (const uint32_t *)part_ids, nq, (const uint32_t *)w.starts, nparts, ek, w.blk,
            reinterpret_cast<uint32_t *>(total_overflow_out + 1), (uint64_t *)nullptr, (uint64_t)0);
  QR_LAUNCH("scan_blocks", scan_u64_kernel, dim3(1), dim3(1024), 0, st, w.blk, (int64_t)b * nparts, total_overflow_out);
  QR_LAUNCH_CHECK("qrlsh_bucket_pairs_count");
  return QRLSH_OK;
}

// records every part's region holds in the one-kernel partition (256 parts): the LDS image of the
// finish for full-size inputs, mean + 50 % + 512 for small ones
static uint32_t part_region(int64_t nq) {
  const int64_t c = ((nq / RADIX) * 3 / 2 + 512 + 63) / 64 * 64;
  // full-size inputs: ONE image of the finish; a part swollen by a popular key spills into the overflow pool and goes
  // to bucket_finish_big_kernel instead of sending the whole step to the general path
  return (uint32_t)(c < FIN_CAP ? c : FIN_CAP);
}

// finer partitions (T > 8) go through two such kernels: 2^c1 coarse regions per band, then 2^(T-c1) fine
// regions inside each (2 x mean + 128, at most the LDS image)
// the T bits of a fine partition are split evenly over the two steps (6 + 6 at T = 12: runs of 64 records per
// (tile, part) in both, 64 reservations per tile; 8 + 4 measured 3.71 ms against 3.45 at 10 M queries)
static int coarse_bits(int T) { return (T + 1) / 2; }
static uint32_t coarse_region(int64_t nq, int c1) {
  // equal keys share a region: beside the hash-uniform spread there must be room for popular keys (one
  // with more copies than the LDS image overflows the fine step anyway)
  const double a = (double)nq / (double)(1 << c1);
  const double slack = a >= 4096.0 ? 0.25 * a + 8192.0 : 6.0 * sqrt(a) + 2.0 * a + 64.0;
  return (uint32_t)(((int64_t)(a + slack) + 63) / 64 * 64);
}
// small parts: the 512-thread / 4096-slot form of the finish (three workgroups per CU)
static bool small_form(int64_t nq, int T) { return (nq >> T) >= 1024 && (nq >> T) <= FIN_SMALL_MEAN; }
static uint32_t fine_region(int64_t nq, int T) {
  // real sizes (mean >= 1024 records per part): the LDS image of the finish form that will run (4096 records for
  // means up to 2800 -- 10 M queries: 2441 --, else 6144): reserved = 1.4 - 2 x the records; a part swollen by a
  // popular key spills into the pool.  Tiny inputs: 2 x mean + 128
  if ((nq >> T) >= 1024) return small_form(nq, T) ? FIN_SMALL_CAP : FIN_CAP;
  const int64_t c = ((nq >> T) * 2 + 128 + 63) / 64 * 64;
  return (uint32_t)(c < FIN_CAP ? c : FIN_CAP);
}
static bool one_kernel_partition(int64_t nq, int part_bits) { return nq < (1ll << 32) && part_bits >= 8; }
// records of the overflow pool behind the regions: 1/16 of the records of the call (what popular keys spill, plus the
// gathered copies of the spilled parts), at least 1 M, below 2^32
static size_t pool_records(int64_t nq, int32_t b) {
  size_t n = (size_t)b * (size_t)nq / 16;
  if (n < ((size_t)1 << 20)) n = (size_t)1 << 20;
  if (n > 0xFFFFFF00ull) n = 0xFFFFFF00ull;
  return n;
}
static size_t region_words(int64_t nq, int32_t b, int32_t part_bits) {
  return part_bits == 8 ? (size_t)b * RADIX * part_region(nq) : ((size_t)b << part_bits) * fine_region(nq, part_bits);
}

// words part_keys / part_ids (and, for part_bits > 8, tmp_keys / tmp_ids) must hold for qrlsh_bucket_pairs_emit
QRLSH_EXPORT size_t qrlsh_bucket_part_words(int64_t nq, int32_t b, int32_t part_bits) {
  if (nq <= 0 || b <= 0) return 0;
  const size_t plain = (size_t)b * nq;
  if (!one_kernel_partition(nq, part_bits)) return plain;
  const size_t regions = region_words(nq, b, part_bits) + pool_records(nq, b);   // [regions][overflow pool]
  return regions > plain ? regions : plain;
}
QRLSH_EXPORT size_t qrlsh_bucket_tmp_words(int64_t nq, int32_t b, int32_t part_bits) {
  if (nq <= 0 || b <= 0 || part_bits <= 8) return 0;
  const size_t plain = (size_t)b * nq;
  if (!one_kernel_partition(nq, part_bits)) return plain;
  const int c1 = coarse_bits(part_bits);
  const size_t regions = ((size_t)b << c1) * coarse_region(nq, c1);
  return regions > plain ? regions : plain;
}

// One-pass form: partition + finish with the output range of every part reserved on a device cursor.
// total_overflow_out[0] ends up holding the exact number of pairs whether or not they fitted
// `capacity` words of pairs_out (nothing is written past it); [1] != 0 flags an oversized part.
QRLSH_EXPORT int qrlsh_bucket_pairs_emit(const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids,
                                         uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r,
                                         int32_t part_bits, void *workspace, size_t workspace_bytes,
                                         uint64_t *pairs_out, uint64_t capacity, uint64_t *total_overflow_out,
                                         void *stream) {
  return qrlsh_bucket_pairs_emit_chunked(keys, 0, 0, 0, part_keys, part_ids, tmp_keys, tmp_ids, nq, b, r, part_bits,
                                         workspace, workspace_bytes, pairs_out, capacity, total_overflow_out, stream);
}

// The same with the keys of band t, query q at keys[(q / key_chunk) * key_chunk_stride + t * key_band_stride +
// q % key_chunk] -- the layout a band-partitioned all-to-all delivers ([rank][band][queries of that rank]:
// key_chunk = queries per rank, key_band_stride = key_chunk, key_chunk_stride = bands * key_chunk) -- so the
// multi-GPU driver needs no transposing copy.  key_chunk = 0: plain [b][nq].
QRLSH_EXPORT int qrlsh_bucket_pairs_emit_chunked(const uint64_t *keys, int64_t key_chunk, int64_t key_chunk_stride,
                                                 int64_t key_band_stride, uint64_t *part_keys, uint32_t *part_ids,
                                                 uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b,
                                                 int32_t r, int32_t part_bits, void *workspace,
                                                 size_t workspace_bytes, uint64_t *pairs_out, uint64_t capacity,
                                                 uint64_t *total_overflow_out, void *stream) {
  QR_CHECK_ARG(key_chunk >= 0 && key_chunk_stride >= 0 && key_band_stride >= 0,
               "qrlsh_bucket_pairs_emit_chunked: bad key layout");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = bucket_check("qrlsh_bucket_pairs_emit", keys, part_keys, part_ids, tmp_keys, tmp_ids, nq, b, r,
                              part_bits, workspace, workspace_bytes, total_overflow_out, st);
  if (rc != QRLSH_OK || nq == 0) return rc;
  QR_CHECK_ARG(pairs_out || capacity == 0, "qrlsh_bucket_pairs_emit: null output with capacity %llu",
               (unsigned long long)capacity);
  const int nparts = 1 << part_bits;
  const BucketWs w = bucket_ws(workspace, nq, b, part_bits);
  // the atomic partition hands the finish x = mix64(key): its free-slot marker is mix64(empty key)
  const uint64_t ekx = qr_mix64(qr_empty_key(r));
#define QR_PART_SCATTER(LEVEL2_, ...) QR_LAUNCH("part_scatter", (part_scatter_atomic_kernel<LEVEL2_>), __VA_ARGS__)
  if (one_kernel_partition(nq, part_bits)) {
    // One-kernel partition(s) into fixed regions + the LDS finish.  The bands are independent of each other all the
    // way to the pair cursor, so they are worked in GROUPS that alternate between the caller's stream and an
    // auxiliary one (api.hip: qr_aux_fork): while one group sits in the finish -- a chain of LDS phases that
    // leaves most of the memory system idle -- the next group's partition, which is nothing but memory traffic,
    // shares the device with it.  Same kernels, same buffers (every group touches only its own bands' regions,
    // cursors and counts), results as unordered as before; QRLSH_OVERLAP=0 (or an active profiler) runs the
    // groups one after the other on the caller's stream.
    const int T = part_bits;
    const bool two = T > 8;
    const int c1 = two ? coarse_bits(T) : 8;
    const uint32_t cap1 = two ? coarse_region(nq, c1) : part_region(nq), cap2 = two ? fine_region(nq, T) : cap1;
    const uint32_t lowmask = (1u << (T - c1)) - 1u;
    // step-1 cursors: the histogram area (two steps) or `starts` (one step, read by the finish); step-2: `starts`
    uint32_t *cur1 = two ? w.ghist : w.starts, *cur2 = w.starts;
    if (hipMemsetAsync(cur1, 0, ((size_t)b << c1) * sizeof(uint32_t), st) != hipSuccess ||
        (two && hipMemsetAsync(cur2, 0, ((size_t)b << T) * sizeof(uint32_t), st) != hipSuccess)) {
      qrlsh_set_error("qrlsh_bucket_pairs_emit: hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    uint32_t *ovf = reinterpret_cast<uint32_t *>(total_overflow_out + 1);
    // the overflow pool behind the regions of the part buffers (PartPool above); the step that fills the parts the
    // finish reads spills into it (the coarse step of a two-step partition does not: its regions have their own slack)
    const size_t reg_words = region_words(nq, b, T);
    PartPool pool;
    pool.keys = part_keys + reg_words;
    pool.vals = part_ids + reg_words;
    pool.cursor = w.poolctl;
    pool.nruns = w.poolctl + 1;
    pool.cap = (uint32_t)pool_records(nq, b);
    pool.runs = w.runs;
    pool.runs_max = POOL_RUNS;
    pool.fill = w.fill;
    pool.slot_base = 0;
    PartPool no_pool = pool;
    no_pool.keys = nullptr;
    no_pool.vals = nullptr;
    if (hipMemsetAsync(w.fill, 0xFF, ((size_t)b << T) * sizeof(uint32_t), st) != hipSuccess ||
        hipMemsetAsync(w.poolctl, 0, 16, st) != hipSuccess) {
      qrlsh_set_error("qrlsh_bucket_pairs_emit: hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    // lists of the parts that outgrow the LDS image (bucket_finish_big_kernel), one per band group, in the count /
    // fill form's block area: [16 counters][group 0's list][group 1's list] ...
    constexpr int MAX_GROUPS = EMIT_MAX_GROUPS;
    unsigned long long *nbig0 = reinterpret_cast<unsigned long long *>(w.blk);
    uint64_t *biglist0 = w.blk + MAX_GROUPS;
    const uint64_t slots = (uint64_t)b << T;  // >= 256 words in the block area
    if (hipMemsetAsync(nbig0, 0, MAX_GROUPS * sizeof(unsigned long long), st) != hipSuccess) {
      qrlsh_set_error("qrlsh_bucket_pairs_emit: hipMemsetAsync failed");
      return QRLSH_EHIP;
    }
    // small parts: the 512-thread / 4096-slot finish
    const bool small_parts = two && small_form(nq, T);
    const int ntiles = (int)ceil_div64(nq, PS_TILE);
    const int64_t band_words = key_band_stride ? key_band_stride : nq;  // words between two bands of the key matrix
    static int groups_env = -1;
    if (groups_env < 0) {
      const char *e = getenv("QRLSH_EMIT_GROUPS");
      groups_env = e ? atoi(e) : 2;  // 10 M queries x 32 bands: 18.84 ms per step with 1 group, 18.42 with 2, 18.9 with 4, 19.2 with 8
      if (groups_env < 1) groups_env = 1;
    }
    // small inputs: one group (the second stream's fork / join and the extra launches cost more than the overlap
    // gives: 1.71 against 1.67 ms per step at 1 M queries x 32 bands)
    int GROUPS = (int64_t)b * nq >= (64ll << 20) ? groups_env : 1;
    if (GROUPS > MAX_GROUPS) GROUPS = MAX_GROUPS;
    const int per = (b + GROUPS - 1) / GROUPS;
    const int ngroups = (b + per - 1) / per;
    const uint64_t list_room = (slots - MAX_GROUPS) / (uint64_t)ngroups;
    const uint32_t big_max = (uint32_t)(list_room < FIN_BIG_LIST ? list_room : FIN_BIG_LIST);
    hipStream_t aux = nullptr;
    int gi = 0;
    for (int g0 = 0; g0 < b; g0 += per, ++gi) {
      const int nb = (b - g0 < per) ? b - g0 : per;
      hipStream_t s = (aux && (gi & 1)) ? aux : st;
      uint64_t *biglist = biglist0 + (size_t)gi * big_max;
      unsigned long long *nbig = nbig0 + gi;
      uint64_t *k1 = two ? tmp_keys : part_keys;
      uint32_t *v1 = two ? tmp_ids : part_ids;
      pool.slot_base = (uint32_t)((size_t)g0 << T);
      BigDesc *desc = w.desc + (size_t)gi * FIN_BIG_LIST;
      QR_PART_SCATTER(false, dim3(ntiles, nb), dim3(SORT_THREADS), 0, s, keys + (size_t)g0 * band_words,
                      (const uint32_t *)nullptr, k1 + ((size_t)g0 << c1) * cap1, v1 + ((size_t)g0 << c1) * cap1, nq,
                      ntiles, 64 - c1, (1u << c1) - 1u, cur1 + ((size_t)g0 << c1), cap1, ovf, qr_empty_key(r),
                      (const uint32_t *)nullptr, 0u, key_chunk, key_chunk_stride, key_band_stride, two ? no_pool : pool);
      if (two)
        QR_PART_SCATTER(true, dim3((unsigned)ceil_div64(cap1, PS_TILE), nb << c1), dim3(SORT_THREADS), 0, s,
                        (const uint64_t *)tmp_keys + ((size_t)g0 << c1) * cap1,
                        (const uint32_t *)tmp_ids + ((size_t)g0 << c1) * cap1, part_keys + ((size_t)g0 << T) * cap2,
                        part_ids + ((size_t)g0 << T) * cap2, (int64_t)0, 0, 64 - T, lowmask, cur2 + ((size_t)g0 << T),
                        cap2, ovf, qr_empty_key(r), (const uint32_t *)cur1 + ((size_t)g0 << c1), cap1, (int64_t)0,
                        (int64_t)0, (int64_t)0, pool);
      // the auxiliary stream is forked once the first group's partition is queued and before its finish is: the
      // second group's partition then starts beside the first group's finish, and the two streams stay half a
      // group out of step
      if (gi == 0 && b > per) aux = qr_aux_fork(st);
      if (small_parts && T >= 12 && g_fin_packed)
        QR_LAUNCH("bucket_emit", bucket_finish_packed_kernel, dim3(nparts, nb), dim3(FIN_PK_THREADS), 0, s,
                  (const uint64_t *)part_keys + ((size_t)g0 << T) * cap2, (const uint32_t *)part_ids + ((size_t)g0 << T) * cap2,
                  nparts, total_overflow_out, ovf, pairs_out, capacity, (const uint32_t *)cur2 + ((size_t)g0 << T), cap2, biglist,
                  nbig, big_max, (uint32_t)((size_t)g0 << T));
      else if (small_parts)
        QR_LAUNCH("bucket_emit", (bucket_finish_kernel<FIN_EMIT, FIN_SMALL_THREADS, FIN_SMALL_CAP>), dim3(nparts, nb),
                  dim3(FIN_SMALL_THREADS), 0, s, (const uint64_t *)part_keys + ((size_t)g0 << T) * cap2,
                  (const uint32_t *)part_ids + ((size_t)g0 << T) * cap2, nq, (const uint32_t *)nullptr, nparts, ekx,
                  total_overflow_out, ovf, pairs_out, capacity, (const uint32_t *)cur2 + ((size_t)g0 << T), cap2, biglist, nbig,
                  big_max, (uint32_t)((size_t)g0 << T));
      else
        QR_LAUNCH("bucket_emit", (bucket_finish_kernel<FIN_EMIT>), dim3(nparts, nb), dim3(FIN_THREADS), 0, s,
                  (const uint64_t *)part_keys + ((size_t)g0 << T) * cap2, (const uint32_t *)part_ids + ((size_t)g0 << T) * cap2,
                  nq, (const uint32_t *)nullptr, nparts, ekx, total_overflow_out, ovf, pairs_out, capacity,
                  (const uint32_t *)cur2 + ((size_t)g0 << T), cap2, biglist, nbig, big_max, (uint32_t)((size_t)g0 << T));
      // the parts of this group the finish listed as larger than its LDS image (usually none: the kernel then finds
      // an empty list), on the group's own stream: they are worked beside the next group
      QR_LAUNCH("bucket_emit_big", bucket_big_gather_kernel, dim3(64), dim3(256), 0, s, (const uint64_t *)biglist,
                (const unsigned long long *)nbig, big_max, desc, (const uint64_t *)part_keys, (const uint32_t *)part_ids,
                (const uint32_t *)cur2, cap2, pool, g_big_limit, ovf);
      QR_LAUNCH("bucket_emit_big", bucket_finish_big_kernel, dim3(FIN_BIG_GRID), dim3(FIN_THREADS), 0, s,
                (const uint64_t *)part_keys, (const uint32_t *)part_ids, (const uint64_t *)pool.keys,
                (const uint32_t *)pool.vals, ekx, (const BigDesc *)desc, (const unsigned long long *)nbig, big_max,
                total_overflow_out, pairs_out, capacity);
    }
    if (aux && qr_aux_join(st) != QRLSH_OK) return QRLSH_EHIP;
    QR_LAUNCH_CHECK("qrlsh_bucket_pairs_emit");
    return QRLSH_OK;
  }
#undef QR_PART_SCATTER
  QR_CHECK_ARG(key_chunk == 0, "qrlsh_bucket_pairs_emit_chunked: chunked keys need nq < 2^32");
  bucket_partition(keys, part_keys, part_ids, tmp_keys, tmp_ids, nq, b, r, part_bits, w, st);
  QR_LAUNCH("bucket_emit", (bucket_finish_kernel<FIN_EMIT>), dim3(nparts, b), dim3(FIN_THREADS), 0, st,
            (const uint64_t *)part_keys, (const uint32_t *)part_ids, nq, (const uint32_t *)w.starts, nparts,
            qr_empty_key(r), total_overflow_out, reinterpret_cast<uint32_t *>(total_overflow_out + 1), pairs_out,
            capacity);
  QR_LAUNCH_CHECK("qrlsh_bucket_pairs_emit");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_bucket_pairs_fill(const uint64_t *part_keys, const uint32_t *part_ids, int64_t nq, int32_t b,
                                         int32_t r, int32_t part_bits, void *workspace, uint64_t *pairs_out,
                                         void *stream) {
  QR_CHECK_ARG(nq >= 0 && b > 0 && r > 0 && part_bits >= 8 && part_bits <= 16,
               "qrlsh_bucket_pairs_fill: bad sizes");
  if (nq == 0) return QRLSH_OK;
  QR_CHECK_ARG(part_keys && part_ids && workspace && pairs_out, "qrlsh_bucket_pairs_fill: null pointer");
  const int nparts = 1 << part_bits;
  const BucketWs w = bucket_ws(workspace, nq, b, part_bits);
  QR_LAUNCH("bucket_fill", (bucket_finish_kernel<FIN_FILL>), dim3(nparts, b), dim3(FIN_THREADS), 0,
            static_cast<hipStream_t>(stream), part_keys, part_ids, nq, (const uint32_t *)w.starts, nparts,
            qr_empty_key(r), w.blk, w.tail, pairs_out, (uint64_t)0);
  QR_LAUNCH_CHECK("qrlsh_bucket_pairs_fill");
  return QRLSH_OK;
}
