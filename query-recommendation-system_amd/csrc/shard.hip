// shard.hip -- device-side glue of the query-sharded (one process per GPU) driver, qrlsh/dist.py.
//
// The reference (lsh.py, recommender.py:145-214) is single-process; SURVEY.md 8(e) shards its hot path by
// query id.  A rank scores candidate pairs whose two signature rows live on up to two ranks, so between the
// kernels of the path proper the driver needs: the set of remote query ids its pairs touch (for the row
// fetch), pairs re-indexed into its row table [local rows | fetched rows], the rows other ranks ask for,
// and edge keys re-based to its own id range.  All of that is index work on the device -- no host
// round trip beyond the sizes an all-to-all needs.
#include <type_traits>

#include "common.h"

// ---- id set: a bitmap over the (padded) global id space + its rank structure ----------------------------
// workspace: bitmap u32[nw] | prefix u64[nw + 1] | scan scratch u64[..]     (nw = ceil(nids / 32))
struct IdSetWs {
  uint32_t *bm;
  uint64_t *prefix, *scratch;
  int64_t nw;
  size_t bytes;
};
static IdSetWs idset_ws(void *workspace, int64_t nids) {
  IdSetWs w;
  w.nw = (nids + 31) / 32;
  char *p = static_cast<char *>(workspace);
  size_t off = 0;
  w.bm = reinterpret_cast<uint32_t *>(p + off);
  off += ((size_t)w.nw * 4 + 15) & ~(size_t)15;
  w.prefix = reinterpret_cast<uint64_t *>(p + off);
  off += (size_t)(w.nw + 2) * 8;
  w.scratch = reinterpret_cast<uint64_t *>(p + off);
  off += (size_t)(ceil_div64(w.nw + 1, SCANL_CHUNK) + 2) * 8;
  w.bytes = off;
  return w;
}

QRLSH_EXPORT size_t qrlsh_idset_workspace_bytes(int64_t nids) {
  if (nids <= 0) return 64;
  return idset_ws(nullptr, nids).bytes;
}

// both endpoints of every pair that fall outside [q0, q0 + nql) set their bit
__global__ __launch_bounds__(256) void idset_mark_kernel(const uint64_t *__restrict__ pairs, int64_t n, uint64_t q0,
                                                         uint64_t nql, uint32_t *__restrict__ bm) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t pr = pairs[t], i = pr >> 32, j = pr & 0xFFFFFFFFull;
  // test before set: a remote query is touched by ~10 pairs (configs[4], one rank: 501 M pairs, 48.7 M remote ids), and
  // an atomic on a bit that is already set is the expensive way to find that out -- 18.7 ms of returning-nothing
  // atomics against ~3 with the test.  The test reads the word at device scope (past the L1, which the atomics of
  // other CUs never update); a stale "unset" only costs the atomic it would have cost anyway.
  if (i - q0 >= nql) {
    const uint32_t m = 1u << (i & 31);
    if (!(__hip_atomic_load(&bm[i >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m)) atomicOr(&bm[i >> 5], m);
  }
  if (j - q0 >= nql) {
    const uint32_t m = 1u << (j & 31);
    if (!(__hip_atomic_load(&bm[j >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m)) atomicOr(&bm[j >> 5], m);
  }
}

__global__ __launch_bounds__(256) void idset_popc_kernel(const uint32_t *__restrict__ bm, int64_t nw,
                                                         uint64_t *__restrict__ pc) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w <= nw) pc[w] = w < nw ? (uint64_t)__popc(bm[w]) : 0ull;   // one word past the end: the scan leaves the total there
}

// number of set ids below `id` (id <= nids)
__device__ static inline uint64_t idset_rank(const uint32_t *__restrict__ bm, const uint64_t *__restrict__ prefix,
                                             uint64_t id) {
  const uint64_t w = id >> 5;
  const uint32_t bit = (uint32_t)(id & 31);
  return prefix[w] + (bit ? (uint64_t)__popc(bm[w] & ((1u << bit) - 1u)) : 0ull);
}

__global__ void idset_bounds_kernel(const uint32_t *__restrict__ bm, const uint64_t *__restrict__ prefix, int64_t nids,
                                    int64_t shard, int world, int64_t *__restrict__ bounds_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > world) return;
  const int64_t id = min((int64_t)g * shard, nids);
  // the bitmap has ceil(nids / 32) words and prefix one more entry: id == nids with nids % 32 == 0 reads
  // prefix[nw] (the total) and no bitmap word
  bounds_out[g] = (int64_t)idset_rank(bm, prefix, (uint64_t)id);
}

__global__ __launch_bounds__(256) void idset_list_kernel(const uint32_t *__restrict__ bm,
                                                         const uint64_t *__restrict__ prefix, int64_t nw,
                                                         uint64_t *__restrict__ ids_out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  uint32_t m = bm[w];
  uint64_t at = prefix[w];
  while (m) {
    const int bit = __ffs((int)m) - 1;
    ids_out[at++] = (uint64_t)w * 32 + bit;
    m &= m - 1;
  }
}

__global__ __launch_bounds__(256) void idset_remap_kernel(const uint64_t *__restrict__ pairs, int64_t n, uint64_t q0,
                                                          uint64_t nql, const uint32_t *__restrict__ bm,
                                                          const uint64_t *__restrict__ prefix,
                                                          uint64_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t pr = pairs[t], i = pr >> 32, j = pr & 0xFFFFFFFFull;
  const uint64_t si = i - q0 < nql ? i - q0 : nql + idset_rank(bm, prefix, i);
  const uint64_t sj = j - q0 < nql ? j - q0 : nql + idset_rank(bm, prefix, j);
  out[t] = si << 32 | sj;
}

QRLSH_EXPORT int qrlsh_idset_build(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, int64_t nids,
                                   int64_t shard, int32_t world, void *workspace, size_t workspace_bytes,
                                   int64_t *bounds_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && q0 >= 0 && nql >= 0 && nids > 0 && nids <= (1ll << 32) && shard >= 1 && world >= 1,
               "qrlsh_idset_build: bad sizes n=%lld nids=%lld", (long long)n, (long long)nids);
  QR_CHECK_ARG(workspace && bounds_out && (n == 0 || pairs), "qrlsh_idset_build: null pointer");
  if (workspace_bytes < qrlsh_idset_workspace_bytes(nids)) {
    qrlsh_set_error("qrlsh_idset_build: workspace %zu < %zu bytes", workspace_bytes, qrlsh_idset_workspace_bytes(nids));
    return QRLSH_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const IdSetWs w = idset_ws(workspace, nids);
  if (hipMemsetAsync(w.bm, 0, (size_t)w.nw * 4, st) != hipSuccess) {
    qrlsh_set_error("qrlsh_idset_build: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  if (n)
    QR_LAUNCH("idset_mark", idset_mark_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, pairs, n,
              (uint64_t)q0, (uint64_t)nql, w.bm);
  QR_LAUNCH("idset_popc", idset_popc_kernel, dim3((unsigned)ceil_div64(w.nw + 1, 256)), dim3(256), 0, st,
            (const uint32_t *)w.bm, w.nw, w.prefix);
  // exclusive scan over nw + 1 words: prefix[nw] ends up holding the total as well
  qr_scan_u64(w.prefix, w.nw + 1, w.prefix + w.nw + 1, w.scratch, st);
  QR_LAUNCH("idset_bounds", idset_bounds_kernel, dim3((unsigned)((world + 1 + 63) / 64)), dim3(64), 0, st,
            (const uint32_t *)w.bm, (const uint64_t *)w.prefix, nids, shard, world, bounds_out);
  QR_LAUNCH_CHECK("qrlsh_idset_build");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_idset_list(const void *workspace, int64_t nids, uint64_t *ids_out, void *stream) {
  QR_CHECK_ARG(nids > 0 && workspace && ids_out, "qrlsh_idset_list: bad arguments");
  const IdSetWs w = idset_ws(const_cast<void *>(workspace), nids);
  QR_LAUNCH("idset_list", idset_list_kernel, dim3((unsigned)ceil_div64(w.nw, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), (const uint32_t *)w.bm, (const uint64_t *)w.prefix, w.nw, ids_out);
  QR_LAUNCH_CHECK("qrlsh_idset_list");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_idset_remap(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, const void *workspace,
                                   int64_t nids, uint64_t *out, void *stream) {
  QR_CHECK_ARG(n >= 0 && q0 >= 0 && nql >= 0 && nids > 0, "qrlsh_idset_remap: bad sizes");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(pairs && out && workspace, "qrlsh_idset_remap: null pointer");
  const IdSetWs w = idset_ws(const_cast<void *>(workspace), nids);
  QR_LAUNCH("idset_remap", idset_remap_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), pairs, n, (uint64_t)q0, (uint64_t)nql, (const uint32_t *)w.bm,
            (const uint64_t *)w.prefix, out);
  QR_LAUNCH_CHECK("qrlsh_idset_remap");
  return QRLSH_OK;
}

// ---- rows other ranks asked for: rows_out[k] = the signature row of global id ids[k], norms_out[k] its norm ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4 *__restrict__ sig, int vec_per_row,
                                                          const int64_t *__restrict__ norm2,
                                                          const uint64_t *__restrict__ ids, int64_t n, uint64_t q0,
                                                          u32x4 *__restrict__ rows_out,
                                                          int64_t *__restrict__ norms_out) {
  const int64_t total = n * vec_per_row;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e / vec_per_row;
    const int c = (int)(e - k * vec_per_row);
    const uint64_t row = ids[k] - q0;
    rows_out[e] = sig[row * vec_per_row + c];
    if (c == 0) norms_out[k] = norm2[row];
  }
}

QRLSH_EXPORT int qrlsh_gather_rows(const void *sig, int64_t row_bytes, const int64_t *norm2, const uint64_t *ids,
                                   int64_t n, int64_t q0, void *rows_out, int64_t *norms_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && q0 >= 0, "qrlsh_gather_rows: bad sizes (row_bytes=%lld)",
               (long long)row_bytes);
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(sig && norm2 && ids && rows_out && norms_out, "qrlsh_gather_rows: null pointer");
  QR_CHECK_ARG((((uintptr_t)sig | (uintptr_t)rows_out) & 15) == 0, "qrlsh_gather_rows: 16-B alignment");
  const int vpr = (int)(row_bytes / 16);
  int64_t blocks = ceil_div64(n * vpr, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  QR_LAUNCH("gather_rows", gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
            static_cast<const u32x4 *>(sig), vpr, norm2, ids, n, (uint64_t)q0, static_cast<u32x4 *>(rows_out), norms_out);
  QR_LAUNCH_CHECK("qrlsh_gather_rows");
  return QRLSH_OK;
}

// ---- edges that arrived at the owner of their src: re-based to the local id range ----------------------
// in : packed  src << (id_bits + 11) | inv << id_bits | dst          (dst == NULL), or
//      key + payload  (src << 11 | inv, dst)                          (any id width)
// out: (src - q0) << (id_bits + 11) | inv << id_bits | dst   -- fits whenever bits(nql) + 11 + id_bits <= 64,
// so the top-K sort can order (src, inv, dst) in one key even when two GLOBAL ids would not fit.
__global__ __launch_bounds__(256) void edges_localize_kernel(const uint64_t *__restrict__ edges,
                                                             const uint32_t *__restrict__ dst, int64_t n,
                                                             int id_bits, uint64_t q0,
                                                             uint64_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint64_t e = edges[t];
  uint64_t src, inv, d;
  if (dst) {
    src = e >> 11;
    inv = e & 0x7FFull;
    d = dst[t];
  } else {
    src = e >> (id_bits + 11);
    inv = (e >> id_bits) & 0x7FFull;
    d = e & ((1ull << id_bits) - 1ull);
  }
  out[t] = (src - q0) << (id_bits + 11) | inv << id_bits | d;
}

QRLSH_EXPORT int qrlsh_edges_localize(const uint64_t *edges, const uint32_t *edge_dst, int64_t n, int32_t id_bits,
                                      int64_t q0, int64_t nql, uint64_t *out, void *stream) {
  QR_CHECK_ARG(n >= 0 && id_bits >= 1 && id_bits <= 32 && q0 >= 0 && nql >= 1, "qrlsh_edges_localize: bad arguments");
  int lb = 1;
  while (lb < 63 && (1ll << lb) < nql) ++lb;
  QR_CHECK_ARG(lb + 11 + id_bits <= 64, "qrlsh_edges_localize: %d local + 11 + %d id bits do not fit 64", lb, id_bits);
  QR_CHECK_ARG(edge_dst || id_bits <= 26, "qrlsh_edges_localize: packed input needs id_bits <= 26");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(edges && out, "qrlsh_edges_localize: null pointer");
  QR_LAUNCH("edges_localize", edges_localize_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0,
            static_cast<hipStream_t>(stream), edges, edge_dst, n, id_bits, (uint64_t)q0, out);
  QR_LAUNCH_CHECK("qrlsh_edges_localize");
  return QRLSH_OK;
}

// ---- answer sets of chosen queries out of the replicated per-shard CSR arrays ("sets" mode of qrlsh/dist.py) ------
// Every rank holds every shard's answer sets as they came off the wire: offs[world][nql + 1] (32- or 64-bit words)
// and rows[world][max_nnz] (16-bit words when the table has at most 65536 rows, else 32-bit), shard g = queries
// [g * nql, (g + 1) * nql).  A scoring rank needs the sets of the remote queries its pairs touch, as one CSR the
// MinHash kernel can take: count (lengths -> exclusive scan, total read back by the caller for the allocation), fill.
template <typename OffT>
__global__ __launch_bounds__(256) void sets_len_kernel(const uint64_t *__restrict__ ids, int64_t n,
                                                       const OffT *__restrict__ offs, int64_t nql,
                                                       uint64_t *__restrict__ lens) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n) return;
  uint64_t len = 0;
  if (t < n) {
    const uint64_t q = ids[t], g = q / (uint64_t)nql, l = q - g * (uint64_t)nql;
    const OffT *o = offs + g * (uint64_t)(nql + 1) + l;
    len = (uint64_t)(o[1] - o[0]);
  }
  lens[t] = len;  // one word past the end: the scan leaves the total there
}

// 16 lanes per query copy its row ids (consecutive lanes = consecutive words)
template <typename OffT, typename RowT>
__global__ __launch_bounds__(256) void sets_fill_kernel(const uint64_t *__restrict__ ids, int64_t n,
                                                        const OffT *__restrict__ offs, const RowT *__restrict__ rows,
                                                        int64_t nql, int64_t max_nnz,
                                                        const uint64_t *__restrict__ out_off,
                                                        int32_t *__restrict__ rows_out) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lig = threadIdx.x & 15;
  if (t >= n) return;
  const uint64_t q = ids[t], g = q / (uint64_t)nql, l = q - g * (uint64_t)nql;
  const uint64_t s0 = (uint64_t)offs[g * (uint64_t)(nql + 1) + l];
  const uint64_t o0 = out_off[t], len = out_off[t + 1] - o0;
  const RowT *src = rows + g * (uint64_t)max_nnz + s0;
  for (uint64_t k = lig; k < len; k += 16) rows_out[o0 + k] = (int32_t)(uint32_t)(typename std::make_unsigned<RowT>::type)src[k];
}

QRLSH_EXPORT size_t qrlsh_gather_sets_workspace_bytes(int64_t n) {
  return (size_t)(ceil_div64((n > 0 ? n : 0) + 1, SCANL_CHUNK) + 2) * sizeof(uint64_t);
}

// offsets_out[n + 1] (uint64): exclusive scan of the chosen queries' set sizes, [n] = the number of row ids
QRLSH_EXPORT int qrlsh_gather_sets_count(const uint64_t *ids, int64_t n, const void *offs, int32_t off_bytes, int64_t nql,
                                         int64_t world, uint64_t *offsets_out, void *workspace, size_t workspace_bytes,
                                         void *stream) {
  QR_CHECK_ARG(n >= 0 && nql > 0 && world > 0 && (off_bytes == 4 || off_bytes == 8) && offsets_out && workspace,
               "qrlsh_gather_sets_count: bad arguments");
  QR_CHECK_ARG(n == 0 || (ids && offs), "qrlsh_gather_sets_count: null pointer");
  if (workspace_bytes < qrlsh_gather_sets_workspace_bytes(n)) {
    qrlsh_set_error("qrlsh_gather_sets_count: workspace %zu < %zu bytes", workspace_bytes, qrlsh_gather_sets_workspace_bytes(n));
    return QRLSH_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)ceil_div64(n + 1, 256)), block(256);
  if (off_bytes == 4)
    QR_LAUNCH("sets_len", (sets_len_kernel<int32_t>), grid, block, 0, st, ids, n, static_cast<const int32_t *>(offs), nql,
              offsets_out);
  else
    QR_LAUNCH("sets_len", (sets_len_kernel<int64_t>), grid, block, 0, st, ids, n, static_cast<const int64_t *>(offs), nql,
              offsets_out);
  uint64_t *sums = static_cast<uint64_t *>(workspace);
  qr_scan_u64(offsets_out, n + 1, sums, sums + 1, st);   // (offsets_out[n] = the total; sums[0] receives it too)
  QR_LAUNCH_CHECK("qrlsh_gather_sets_count");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_gather_sets_fill(const uint64_t *ids, int64_t n, const void *offs, int32_t off_bytes,
                                        const void *rows, int32_t row_bytes, int64_t nql, int64_t max_nnz,
                                        const uint64_t *offsets, int32_t *rows_out, void *stream) {
  QR_CHECK_ARG(n >= 0 && nql > 0 && max_nnz >= 0 && (off_bytes == 4 || off_bytes == 8) && (row_bytes == 2 || row_bytes == 4),
               "qrlsh_gather_sets_fill: bad arguments");
  if (n == 0) return QRLSH_OK;
  QR_CHECK_ARG(ids && offs && rows && offsets && rows_out, "qrlsh_gather_sets_fill: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)ceil_div64(n * 16, 256)), block(256);
#define QR_SETS_FILL(OT, RT)                                                                                          \
  QR_LAUNCH("sets_fill", (sets_fill_kernel<OT, RT>), grid, block, 0, st, ids, n, static_cast<const OT *>(offs),       \
            static_cast<const RT *>(rows), nql, max_nnz, offsets, rows_out)
  if (off_bytes == 4 && row_bytes == 2) QR_SETS_FILL(int32_t, int16_t);
  else if (off_bytes == 4) QR_SETS_FILL(int32_t, int32_t);
  else if (row_bytes == 2) QR_SETS_FILL(int64_t, int16_t);
  else QR_SETS_FILL(int64_t, int32_t);
#undef QR_SETS_FILL
  QR_LAUNCH_CHECK("qrlsh_gather_sets_fill");
  return QRLSH_OK;
}
