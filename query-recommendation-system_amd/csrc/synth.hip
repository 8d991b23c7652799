// synth.hip -- synthetic answer sets for bench / tests.
//
// Not part of the reference's path: this is the input generator.  Bit-identical twin of
// oracle/qr_oracle.c:qro_synth_* -- a pure function of (seed, query index), so the CPU
// baseline, the GPU and every shard regenerate the same rows without a sequential RNG.
#include "common.h"

constexpr int SYN_MAXS = 48;

__device__ static inline uint64_t syn_rnd(uint64_t seed, uint64_t stream, uint64_t idx, uint64_t ctr) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (stream + 1);
  x = qr_mix64(x ^ (idx * 0xD1342543DE82EF95ull));
  x = qr_mix64(x + ctr * 0xA24BAED4963EE407ull + 0x9E3779B97F4A7C15ull);
  return x;
}
__device__ static inline uint32_t syn_range(uint64_t r, uint32_t D) {
  return (uint32_t)(((r >> 32) * (uint64_t)D) >> 32);
}

__device__ static int syn_one(uint64_t seed, int64_t q, int64_t nb, uint32_t D, const uint32_t *__restrict__ cdf,
                              int ncdf, uint32_t thr, uint32_t *out) {
  const uint64_t bidx = (uint64_t)(q % nb);
  const uint32_t u = (uint32_t)(syn_rnd(seed, 1, bidx, 0) >> 32);
  int size = 0;
  while (size < ncdf && u >= cdf[size]) ++size;
  if (size < 1) size = 1;
  if (size > SYN_MAXS) size = SYN_MAXS;
  int n = 0;
  for (int k = 0; k < size; ++k) {
    uint32_t e = syn_range(syn_rnd(seed, 2, bidx, (uint64_t)k), D);
    const uint64_t rr = syn_rnd(seed, 3, (uint64_t)q, (uint64_t)k);
    if ((uint32_t)(rr & 0xFFFFFFu) < thr) e = syn_range(syn_rnd(seed, 4, (uint64_t)q, (uint64_t)k), D);
    int pos = n;
    while (pos > 0 && out[pos - 1] > e) --pos;
    if (pos > 0 && out[pos - 1] == e) continue;
    for (int m = n; m > pos; --m) out[m] = out[m - 1];
    out[pos] = e;
    ++n;
  }
  return n;
}

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nb, uint32_t D,
                                                    const uint32_t *__restrict__ cdf, int ncdf, uint32_t thr,
                                                    int32_t *__restrict__ sizes, const int64_t *__restrict__ offsets,
                                                    int32_t *__restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq_local) return;
  uint32_t tmp[SYN_MAXS];
  const int n = syn_one(seed, q0 + i, nb, D, cdf, ncdf, thr, tmp);
  if (sizes) sizes[i] = n;
  if (rows) {
    const int64_t o = offsets[i];
    for (int k = 0; k < n; ++k) rows[o + k] = (int32_t)tmp[k];
  }
}

static int synth_launch(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total, int32_t cluster, uint32_t D,
                        const uint32_t *cdf, int32_t ncdf, uint32_t thr, int32_t *sizes, const int64_t *offsets,
                        int32_t *rows, void *stream) {
  QR_CHECK_ARG(nq_local >= 0 && nq_total > 0 && cluster > 0 && D > 0 && cdf && ncdf > 0, "qrlsh_synth: bad arguments");
  if (nq_local == 0) return QRLSH_OK;
  int64_t nb = nq_total / cluster;
  if (nb < 1) nb = 1;
  QR_LAUNCH("synth", synth_kernel, dim3((unsigned)ceil_div64(nq_local, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seed, q0, nq_local, nb, D, cdf, ncdf, thr, sizes, offsets, rows);
  QR_LAUNCH_CHECK("qrlsh_synth");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_synth_sizes(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total, int32_t cluster,
                                   uint32_t D, const uint32_t *cdf, int32_t ncdf, uint32_t rep_thresh24,
                                   int32_t *sizes_out, void *stream) {
  QR_CHECK_ARG(sizes_out, "qrlsh_synth_sizes: null output");
  return synth_launch(seed, q0, nq_local, nq_total, cluster, D, cdf, ncdf, rep_thresh24, sizes_out, nullptr, nullptr,
                      stream);
}

QRLSH_EXPORT int qrlsh_synth_fill(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total, int32_t cluster,
                                  uint32_t D, const uint32_t *cdf, int32_t ncdf, uint32_t rep_thresh24,
                                  const int64_t *offsets, int32_t *rows_out, void *stream) {
  QR_CHECK_ARG(offsets && rows_out, "qrlsh_synth_fill: null pointer");
  return synth_launch(seed, q0, nq_local, nq_total, cluster, D, cdf, ncdf, rep_thresh24, nullptr, offsets, rows_out,
                      stream);
}
