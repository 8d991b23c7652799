// common.h -- shared device helpers and host-side error plumbing for libqrlsh (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qrlsh.h"

#define QRLSH_EXPORT extern "C" __attribute__((visibility("default")))

constexpr int WAVE = 64;  // CDNA4 wavefront

void qrlsh_set_error(const char *fmt, ...);

#define QR_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      qrlsh_set_error(__VA_ARGS__);    \
      return QRLSH_EINVAL;             \
    }                                  \
  } while (0)

#define QR_LAUNCH_CHECK(name)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      qrlsh_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return QRLSH_EHIP;                                                             \
    }                                                                                \
  } while (0)

// optional per-kernel timing with HIP events on the launch stream (qrlsh_prof_* in api.hip)
int qr_prof_begin(const char *label, hipStream_t st);
void qr_prof_end(int slot, hipStream_t st);

// second stream for overlapping independent pieces of one call (api.hip); nullptr = run everything on `st`
hipStream_t qr_aux_fork(hipStream_t st);
int qr_aux_join(hipStream_t st);

#define QR_LAUNCH(label, kern, grid, block, smem, st, ...)               \
  do {                                                                   \
    const int ps__ = qr_prof_begin(label, st);                           \
    hipLaunchKernelGGL(kern, grid, block, smem, st, __VA_ARGS__);        \
    qr_prof_end(ps__, st);                                               \
  } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-bit bijective mixer (splitmix64 finaliser).  Bijective => equal mixes <=> equal keys
// on all 64 bits; the grouping sort uses only the top 32 bits and resolves the (rare)
// 32-bit collisions with a full-key compare at pair emission.
__host__ __device__ static inline uint64_t qr_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// The rank that scores candidate pair i << 32 | j in the sharded driver (qrlsh/dist.py): the owner of i or
// the owner of j (ranks own contiguous shards of `shard` ids), picked by the top bit of mix64(pair).  Always
// the smaller id would give rank g a share proportional to the ids ABOVE its shard (twice the mean on rank
// 0, nothing on the last rank); a data-independent coin splits the pairs evenly for any id structure.
__host__ __device__ static inline uint64_t qr_pair_host(uint64_t pair, uint32_t shard) {
  const uint64_t id = (qr_mix64(pair) >> 63) ? (pair & 0xFFFFFFFFull) : (pair >> 32);
  return id / shard;
}

// key of a band whose r int16 values are all -1 (skipped by get_candidates, lsh.py:47)
__host__ __device__ static inline uint64_t qr_empty_key(int r) {
  return r >= 4 ? ~0ull : ((1ull << (16 * r)) - 1ull);
}

// Band key from the r low-16 values of one band (lsh.py:28-34).  r <= 4: the tuple itself,
// packed -- an exact bucket id.  r > 4 ("wide band"): a 64-bit hash of the tuple; buckets can
// then (with probability ~2^-64 per pair) merge distinct tuples, so the host runs
// qrlsh_verify_pairs on the unique pairs afterwards and the result stays exact.  The all -1
// tuple always maps to ~0 and nothing else does.
__device__ static inline uint64_t qr_make_key(const uint16_t *s, int r) {
  if (r <= 4) {
    uint64_t k = 0;
    for (int j = 0; j < r; ++j) k |= (uint64_t)s[j] << (16 * j);
    return k;
  }
  uint64_t h = 0x243F6A8885A308D3ull;
  bool empty = true;
  for (int j = 0; j < r; ++j) {
    h = (h ^ s[j]) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    empty &= s[j] == 0xFFFFu;
  }
  h = qr_mix64(h);
  if (empty) return ~0ull;
  return h == ~0ull ? h - 1 : h;
}

// Streams that are written (or read) once and not touched again by the same kernel go past the caches' normal
// allocation (non-temporal hint), leaving L2 to the data the kernel re-reads.  Per kernel, decided by
// measurement (tools/ab_flags.sh, same box): QR_NT_MINHASH (answer-set ids in, signatures / keys / norms out,
// beside the table gather): 3.89 -> 3.61 ms at 10 M queries.  The same hint on the scoring kernel's pair /
// score / edge streams made it slower (3.80 -> 3.90) and is not used.
#ifndef QR_NT_MINHASH
#define QR_NT_MINHASH 1
#endif
template <bool NT, typename T> __device__ static inline void qr_store(T v, T *p) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT, typename T> __device__ static inline T qr_load(const T *p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

__device__ static inline int lane_id() { return threadIdx.x & (WAVE - 1); }

// wave-level inclusive scan of a u64 via shuffles
__device__ static inline uint64_t wave_incl_scan_u64(uint64_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint64_t o = __shfl_up(v, d, WAVE);
    if (lane >= d) v += o;
  }
  return v;
}

// block-level exclusive scan of one u64 per thread (blockDim.x = 256); returns the
// exclusive prefix for this thread and the block total in *total.  smem: >= 4 u64.
__device__ static inline uint64_t block_excl_scan_u64_256(uint64_t v, uint64_t *smem, uint64_t *total) {
  const int lane = lane_id(), w = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan_u64(v);
  if (lane == WAVE - 1) smem[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint64_t s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---- large arrays: three small launches (chunk scans, scan of the chunk totals, add) ------------
constexpr int SCANL_THREADS = 1024;
constexpr int SCANL_PER = 16;
constexpr int SCANL_CHUNK = SCANL_THREADS * SCANL_PER;  // 16384 elements per workgroup

// in-place exclusive scan of every chunk; chunk total -> sums[chunk]
static __global__ __launch_bounds__(SCANL_THREADS) void scan_chunks_kernel(uint64_t *__restrict__ a, int64_t m,
                                                                           uint64_t *__restrict__ sums) {
  __shared__ uint64_t wsum[SCANL_THREADS / 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t lo = (int64_t)blockIdx.x * SCANL_CHUNK + (int64_t)t * SCANL_PER;
  uint64_t held[SCANL_PER], s = 0;
#pragma unroll
  for (int k = 0; k < SCANL_PER; ++k) held[k] = lo + k < m ? a[lo + k] : 0;
#pragma unroll
  for (int k = 0; k < SCANL_PER; ++k) s += held[k];
  uint64_t inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint64_t run = inc - s, tot = 0;
#pragma unroll
  for (int k = 0; k < SCANL_THREADS / 64; ++k) {
    if (k < w) run += wsum[k];
    tot += wsum[k];
  }
#pragma unroll
  for (int k = 0; k < SCANL_PER; ++k)
    if (lo + k < m) {
      a[lo + k] = run;
      run += held[k];
    }
  if (t == 0) sums[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(SCANL_THREADS) void scan_add_kernel(uint64_t *__restrict__ a, int64_t m,
                                                                        const uint64_t *__restrict__ sums) {
  const uint64_t add = sums[blockIdx.x];
  const int64_t lo = (int64_t)blockIdx.x * SCANL_CHUNK;
#pragma unroll
  for (int k = 0; k < SCANL_PER; ++k) {
    const int64_t i = lo + (int64_t)k * SCANL_THREADS + threadIdx.x;
    if (i < m) a[i] += add;
  }
}

// in-place exclusive scan of m uint64 (one workgroup); the grand total goes to *total_out
static __global__ __launch_bounds__(1024) void scan_u64_kernel(uint64_t *__restrict__ a, int64_t m,
                                                        uint64_t *__restrict__ total_out) {
  __shared__ uint64_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t per = (m + 1023) / 1024;
  const int64_t lo = min((int64_t)t * per, m), hi = min(lo + per, m);
  uint64_t s = 0;
  constexpr int SCAN_REG = 32;  // up to 32 K elements: every load is issued before the first add
  uint64_t held[SCAN_REG];
  const bool small = per <= SCAN_REG;
  if (small) {
#pragma unroll
    for (int k = 0; k < SCAN_REG; ++k) held[k] = lo + k < hi ? a[lo + k] : 0;
#pragma unroll
    for (int k = 0; k < SCAN_REG; ++k) s += held[k];
  } else {
    for (int64_t i = lo; i < hi; ++i) s += a[i];
  }
  const uint64_t inc = wave_incl_scan_u64(s);
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    if (i < w) base += wsum[i];
    tot += wsum[i];
  }
  uint64_t run = base + inc - s;
  if (small) {
#pragma unroll
    for (int k = 0; k < SCAN_REG; ++k)
      if (lo + k < hi) {
        a[lo + k] = run;
        run += held[k];
      }
  } else {
    for (int64_t i = lo; i < hi; ++i) {
      const uint64_t v = a[i];
      a[i] = run;
      run += v;
    }
  }
  if (t == 0) *total_out = tot;
}

// exclusive scan of m uint64 in place, total to *total_out.  `sums` = scratch of
// ceil(m / SCANL_CHUNK) words, used (with two more launches) only when the array is large.
static inline void qr_scan_u64(uint64_t *a, int64_t m, uint64_t *total_out, uint64_t *sums, hipStream_t st) {
  if (m <= 2 * SCANL_CHUNK || sums == nullptr) {
    QR_LAUNCH("scan_blocks", scan_u64_kernel, dim3(1), dim3(1024), 0, st, a, m, total_out);
    return;
  }
  const int64_t nch = (m + SCANL_CHUNK - 1) / SCANL_CHUNK;
  QR_LAUNCH("scan_blocks", scan_chunks_kernel, dim3((unsigned)nch), dim3(SCANL_THREADS), 0, st, a, m, sums);
  QR_LAUNCH("scan_blocks", scan_u64_kernel, dim3(1), dim3(1024), 0, st, sums, nch, total_out);
  QR_LAUNCH("scan_blocks", scan_add_kernel, dim3((unsigned)nch), dim3(SCANL_THREADS), 0, st, a, m,
            (const uint64_t *)sums);
}

