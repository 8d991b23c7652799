// minhash.hip -- a1 (MinHash signatures) with fused a2 (band keys) and row norms.
//
// Reference: Recommender.compute_signatures, recommender.py:105-143, and the key
// construction of LSH.make_subvecs / compute_buckets, lsh.py:17-38.
//
// Design (MI355X): the P permutations are stored TRANSPOSED, perm_t[d][0..P), so that
//     sig[q][:] = elementwise-min over d in A(q) of perm_t[d][:]
// is a row gather + min: every answer-set row is one contiguous 2P- or 4P-byte read
// (16 B per lane, LPR lanes per row, 64/LPR rows in flight per wave-instruction) from a
// table that lives in L2 / Infinity Cache (8 MB for D=32768, P=128, uint16).  One wave
// owns one query at a time; a 256-thread workgroup owns QPB consecutive queries so the
// band keys can be transposed through LDS and written band-major ([b][nq]) in full
// 512-byte runs.
#include "common.h"

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct TabVec;
template <> struct TabVec<uint16_t> {
  using type = u16x8;
  static constexpr int N = 8;
  __device__ static type init() { return (type)(0xFFFF); }
};
template <> struct TabVec<int32_t> {
  using type = i32x4;
  static constexpr int N = 4;
  __device__ static type init() { return (type)(0x7FFFFFFF); }
};

template <typename V> __device__ static inline V vec_shfl_xor(V v, int m) {
  typedef int i4 __attribute__((ext_vector_type(4)));
  i4 x = __builtin_bit_cast(i4, v);
  x.x = __shfl_xor(x.x, m, WAVE);
  x.y = __shfl_xor(x.y, m, WAVE);
  x.z = __shfl_xor(x.z, m, WAVE);
  x.w = __shfl_xor(x.w, m, WAVE);
  return __builtin_bit_cast(V, x);
}

// Write one lane's VEC finished values of query q (columns [col, col+VEC)): the int32 row
// (reference layout), and/or the compact uint16 row (0xFFFF = -1; only when D <= 65535), the
// low-16 image for the band keys, and the lane's share of the squared norm.
template <typename TabT>
__device__ static inline int64_t store_sig(typename TabVec<TabT>::type acc, int n, int64_t q, int col, int P,
                                           bool vec_store, int32_t *__restrict__ sig, uint16_t *__restrict__ sig16,
                                           uint16_t *srow) {
  constexpr int VEC = TabVec<TabT>::N;
  int32_t out[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) out[e] = (n > 0) ? (int32_t)acc[e] : -1;
  int64_t nrm = 0;
  if (vec_store && col + VEC <= P) {
    if (sig) {
      int32_t *dst = sig + (size_t)q * P + col;
#pragma unroll
      for (int e = 0; e < VEC; e += 4)
        *reinterpret_cast<i32x4 *>(dst + e) = (i32x4){out[e], out[e + 1], out[e + 2], out[e + 3]};
    }
    if (sig16) {
      uint16_t *d16 = sig16 + (size_t)q * P + col;
      if (VEC == 8 && (P % 8) == 0) {
        u16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (uint16_t)out[e % VEC];
        qr_store<QR_NT_MINHASH != 0>(v, reinterpret_cast<u16x8 *>(d16));
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d16[e] = (uint16_t)out[e];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) nrm += (int64_t)out[e] * out[e];
    if (srow) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) srow[col + e] = (uint16_t)out[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (col + e < P) {
        if (sig) sig[(size_t)q * P + col + e] = out[e];
        if (sig16) sig16[(size_t)q * P + col + e] = (uint16_t)out[e];
        nrm += (int64_t)out[e] * out[e];
        if (srow) srow[col + e] = (uint16_t)out[e];
      }
  }
  return nrm;
}

// LPR = lanes per table row (power of two); one lane covers VEC = 16 B / sizeof(TabT)
// permutations; G = 64 / LPR rows are fetched per wave-instruction.
template <typename TabT, int LPR>
__global__ __launch_bounds__(256) void minhash_kernel(const int64_t *__restrict__ offsets,
                                                      const int32_t *__restrict__ rows, int64_t nq,
                                                      const TabT *__restrict__ tab, int P, int P_stride,
                                                      int32_t *__restrict__ sig, uint16_t *__restrict__ sig16,
                                                      int64_t *__restrict__ norm2, uint64_t *__restrict__ keys, int b,
                                                      int r, int qpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t *s16 = reinterpret_cast<uint16_t *>(smem_raw);  // [qpb][ldk] low-16 signature image
  using VecT = typename TabVec<TabT>::type;
  constexpr int VEC = TabVec<TabT>::N;
  constexpr int G = WAVE / LPR;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  const int g = lane / LPR, lig = lane % LPR;
  const int64_t q0 = (int64_t)blockIdx.x * qpb;
  const int ldk = P + 2;
  const bool vec_store = (P % 4) == 0;

  for (int ql = wave; ql < qpb; ql += 4) {
    const int64_t q = q0 + ql;
    if (q >= nq) break;  // wave-uniform
    const int64_t lo = offsets[q];
    const int n = (int)(offsets[q + 1] - lo);
    int64_t nrm = 0;
    for (int col0 = 0; col0 < P_stride; col0 += LPR * VEC) {
      const int col = col0 + lig * VEC;
      const bool colok = col < P_stride;
      VecT acc = TabVec<TabT>::init();
      for (int base = 0; base < n; base += WAVE) {
        const int cnt = min(WAVE, n - base);
        const int my = (lane < cnt) ? rows[lo + base + lane] : 0;
        const int iters = (cnt + G - 1) / G;
        for (int it = 0; it < iters; ++it) {
          const int i = it * G + g;
          const int d = __shfl(my, i & (WAVE - 1), WAVE);
          if (i < cnt && colok) {
            const VecT v = *reinterpret_cast<const VecT *>(tab + (size_t)d * P_stride + col);
            acc = __builtin_elementwise_min(acc, v);
          }
        }
      }
#pragma unroll
      for (int m = LPR; m < WAVE; m <<= 1) acc = __builtin_elementwise_min(acc, vec_shfl_xor(acc, m));

      if (g == 0 && colok) nrm += store_sig<TabT>(acc, n, q, col, P, vec_store, sig, sig16, keys ? s16 + ql * ldk : nullptr);
    }
    if (norm2) {
#pragma unroll
      for (int m = 1; m < WAVE; m <<= 1) nrm += __shfl_xor(nrm, m, WAVE);
      if (lane == 0) norm2[q] = nrm;
    }
  }

  if (keys) {
    __syncthreads();
    const int nql = (int)min((int64_t)qpb, nq - q0);
    for (int idx = threadIdx.x; idx < qpb * b; idx += blockDim.x) {
      const int ql = idx % qpb, band = idx / qpb;
      if (ql < nql) {
        keys[(size_t)band * nq + q0 + ql] = qr_make_key(s16 + ql * ldk + band * r, r);
      }
    }
  }
}

// ---- v2: one LPR-lane GROUP per query, G = 64/LPR queries in flight per wave ----------------
// The v1 kernel above walks one query per wave and pays three dependent memory latencies
// (offsets -> row ids -> table rows) per query with nothing else in flight.  Here a wave owns
// 16 consecutive queries: their 17 offsets come from ONE coalesced load, each group gathers
// all rows of ITS query with independent 16-B loads (no cross-group reduction: every lane ends
// up with the final minimum of its 16-B column chunk), and the row ids of the next batch of G
// queries are prefetched while the current batch's gathers are in flight.
constexpr int MH_QPW = 16;  // queries per wave (4 waves -> 64 per workgroup)
#ifndef MH_CH
#define MH_CH 4  // table rows gathered per step (independent loads in flight per lane)
#endif

// value of lane `idx` of every LPR-lane group.  Groups of 16 are DPP rows: one v_mov with
// row_newbcast instead of an LDS-pipe ds_bpermute plus its wait.
template <int I> __device__ static inline int row_bcast(int v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + I, 0xF, 0xF, false);
}
template <int LPR> __device__ static inline int group_bcast(int v, int idx, int g) {
  if constexpr (LPR == 16) {
    switch (idx) {
      case 0: return row_bcast<0>(v);
      case 1: return row_bcast<1>(v);
      case 2: return row_bcast<2>(v);
      case 3: return row_bcast<3>(v);
      case 4: return row_bcast<4>(v);
      case 5: return row_bcast<5>(v);
      case 6: return row_bcast<6>(v);
      case 7: return row_bcast<7>(v);
      case 8: return row_bcast<8>(v);
      case 9: return row_bcast<9>(v);
      case 10: return row_bcast<10>(v);
      case 11: return row_bcast<11>(v);
      case 12: return row_bcast<12>(v);
      case 13: return row_bcast<13>(v);
      case 14: return row_bcast<14>(v);
      default: return row_bcast<15>(v);
    }
  } else {
    return __shfl(v, g * LPR + idx, WAVE);
  }
}

template <typename TabT, int LPR>
__global__ __launch_bounds__(256) void minhash_group_kernel(const int64_t *__restrict__ offsets,
                                                            const int32_t *__restrict__ rows, int64_t nq,
                                                            const TabT *__restrict__ tab, int P, int P_stride,
                                                            int32_t *__restrict__ sig, uint16_t *__restrict__ sig16,
                                                            int64_t *__restrict__ norm2, uint64_t *__restrict__ keys,
                                                            int b, int r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t *s16 = reinterpret_cast<uint16_t *>(smem_raw);  // [64][ldk] low-16 signature image
  using VecT = typename TabVec<TabT>::type;
  constexpr int VEC = TabVec<TabT>::N;
  constexpr int G = WAVE / LPR;
  constexpr int QPB = 4 * MH_QPW;
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  const int g = lane / LPR, lig = lane % LPR;
  const int64_t q0 = (int64_t)blockIdx.x * QPB;
  const int64_t qw0 = q0 + wave * MH_QPW;
  const int ldk = P + 2;
  const int col = lig * VEC;
  const bool colok = col < P_stride;
  const bool vec_store = (P % 4) == 0;
  const char *tbytes = reinterpret_cast<const char *>(tab);
  const uint32_t row_bytes = (uint32_t)P_stride * sizeof(TabT), col_bytes = (colok ? col : 0) * sizeof(TabT);

  // 17 offsets of this wave's 16 queries in one load
  int64_t offs = 0;
  {
    const int64_t qi = qw0 + lane;
    if (lane <= MH_QPW) offs = qr_load<QR_NT_MINHASH != 0>(&offsets[qi < nq ? qi : nq]);
  }
  auto batch_lo = [&](int batch, int &n, int &qlw) -> int64_t {
    qlw = batch * G + g;  // the wave-local query this group works on in this batch
    const int64_t lo = __shfl(offs, qlw, WAVE), hi = __shfl(offs, qlw + 1, WAVE);
    n = (qw0 + qlw < nq) ? (int)(hi - lo) : 0;
    return lo;
  };
  int n_cur, qlw_cur;
  int64_t lo_cur = batch_lo(0, n_cur, qlw_cur);
  int my_cur = (lig < n_cur) ? qr_load<QR_NT_MINHASH != 0>(&rows[lo_cur + lig]) : 0;

#pragma unroll 1
  for (int batch = 0; batch < MH_QPW / G; ++batch) {
    const int ql = wave * MH_QPW + qlw_cur;  // query index inside the workgroup
    const int64_t q = q0 + ql;
    // prefetch the next batch's first LPR row ids
    int n_nxt = 0, my_nxt = 0, qlw_nxt = 0;
    int64_t lo_nxt = 0;
    if (batch + 1 < MH_QPW / G) {
      lo_nxt = batch_lo(batch + 1, n_nxt, qlw_nxt);
      if (lig < n_nxt) my_nxt = qr_load<QR_NT_MINHASH != 0>(&rows[lo_nxt + lig]);
    }
    VecT acc = TabVec<TabT>::init();
    int my = my_cur;
    // the MH_CH gathers of a step sit in their own exec regions with the min after all of them, so
    // they are independent loads in flight together, not MH_CH load -> wait -> min round trips
    for (int base = 0; __any(base < n_cur); base += LPR) {
      const int my_after = (base + LPR + lig < n_cur) ? qr_load<QR_NT_MINHASH != 0>(&rows[lo_cur + base + LPR + lig]) : 0;
#pragma unroll
      for (int sub = 0; sub < LPR; sub += MH_CH) {
        if (sub > 0 && !__any(base + sub < n_cur)) break;  // wave-uniform
        VecT v[MH_CH];
#pragma unroll
        for (int j = 0; j < MH_CH; ++j) {
          const int ds = group_bcast<LPR>(my, sub + j, g);
          // 32-bit byte offset from a uniform base (table < 4 GiB, D <= 2^24: checked by the host; 0 <= d < D is the
          // caller's precondition -- qrlsh_check_csr tests it, ops.minhash(validate=True) by default)
          v[j] = TabVec<TabT>::init();
          if (base + sub + j < n_cur)
            v[j] = *reinterpret_cast<const VecT *>(tbytes + (__umul24((uint32_t)ds, row_bytes) + col_bytes));
        }
#pragma unroll
        for (int j = 0; j < MH_CH; ++j) acc = __builtin_elementwise_min(acc, v[j]);
      }
      my = my_after;
    }
    if (q < nq) {
      int64_t nrm = 0;
      if (colok) nrm = store_sig<TabT>(acc, n_cur, q, col, P, vec_store, sig, sig16, keys ? s16 + ql * ldk : nullptr);
      if (norm2) {
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) nrm += __shfl_xor(nrm, m, WAVE);
        if (lig == 0) qr_store<QR_NT_MINHASH != 0>(nrm, &norm2[q]);
      }
    } else if (norm2) {
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) (void)__shfl_xor((int64_t)0, m, WAVE);
    }
    n_cur = n_nxt;
    lo_cur = lo_nxt;
    my_cur = my_nxt;
    qlw_cur = qlw_nxt;
  }

  if (keys) {
    __syncthreads();
    const int nql = (int)min((int64_t)QPB, nq - q0);
    for (int idx = threadIdx.x; idx < QPB * b; idx += blockDim.x) {
      const int ql = idx % QPB, band = idx / QPB;
      if (ql < nql) {
        qr_store<QR_NT_MINHASH != 0>(qr_make_key(s16 + ql * ldk + band * r, r), &keys[(size_t)band * nq + q0 + ql]);
      }
    }
  }
}

// a2 standalone: band keys (band-major) from an int32 signature matrix.
__global__ __launch_bounds__(256) void band_keys_kernel(const int32_t *__restrict__ sig, int64_t nq, int P,
                                                        int b, int r, uint64_t *__restrict__ keys, int qpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t *s16 = reinterpret_cast<uint16_t *>(smem_raw);
  const int ldk = P + 2;
  const int64_t q0 = (int64_t)blockIdx.x * qpb;
  const int nql = (int)min((int64_t)qpb, nq - q0);
  const int32_t *src = sig + (size_t)q0 * P;
  for (int idx = threadIdx.x; idx < nql * P; idx += blockDim.x) {
    const int ql = idx / P, c = idx - ql * P;
    s16[ql * ldk + c] = (uint16_t)src[idx];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < qpb * b; idx += blockDim.x) {
    const int ql = idx % qpb, band = idx / qpb;
    if (ql < nql) {
      keys[(size_t)band * nq + q0 + ql] = qr_make_key(s16 + ql * ldk + band * r, r);
    }
  }
}

// exact squared L2 norm of every signature row; one wave per row, or (rows of thousands of columns: N4's rating
// rows) one workgroup per row.
template <bool WG_PER_ROW>
__global__ __launch_bounds__(256) void row_norms_kernel(const int32_t *__restrict__ sig, int64_t nq, int P,
                                                        int64_t *__restrict__ norm2) {
  const int lane = threadIdx.x & (WAVE - 1);
  if (WG_PER_ROW) {
    __shared__ int64_t part[256 / WAVE];
    for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
      const int32_t *s = sig + (size_t)q * P;
      int64_t acc = 0;
      for (int c = threadIdx.x; c < P; c += 256) acc += (int64_t)s[c] * s[c];
#pragma unroll
      for (int m = 1; m < WAVE; m <<= 1) acc += __shfl_xor(acc, m, WAVE);
      if (lane == 0) part[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) {
        int64_t t = 0;
#pragma unroll
        for (int i = 0; i < 256 / WAVE; ++i) t += part[i];
        norm2[q] = t;
      }
      __syncthreads();
    }
    return;
  }
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t q = wid; q < nq; q += nw) {
    const int32_t *s = sig + (size_t)q * P;
    int64_t acc = 0;
    for (int c = lane; c < P; c += WAVE) acc += (int64_t)s[c] * s[c];
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) acc += __shfl_xor(acc, m, WAVE);
    if (lane == 0) norm2[q] = acc;
  }
}

// Precondition check of the CSR answer sets (qrlsh_minhash gathers table row rows[k] unchecked): flag bit 0 =
// offsets[0] != 0 or offsets[nq] != nnz, bit 1 = offsets decrease somewhere, bit 2 = a row id outside [0, D).
__global__ __launch_bounds__(256) void check_csr_kernel(const int64_t *__restrict__ offsets,
                                                        const int32_t *__restrict__ rows, int64_t nq, int64_t nnz, int D,
                                                        uint32_t *__restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad = 0;
  if (t == 0 && (offsets[0] != 0 || offsets[nq] != nnz)) bad |= 1u;
  if (t < nq && offsets[t + 1] < offsets[t]) bad |= 2u;
  if (t < nnz && (uint32_t)rows[t] >= (uint32_t)D) bad |= 4u;
  if (bad) atomicOr(flags, bad);
}

QRLSH_EXPORT int qrlsh_check_csr(const int64_t *offsets, const int32_t *rows, int64_t nq, int64_t nnz, int32_t D,
                                 uint32_t *flags_out, void *stream) {
  QR_CHECK_ARG(nq >= 0 && nnz >= 0 && D > 0 && offsets && flags_out && (nnz == 0 || rows), "qrlsh_check_csr: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(flags_out, 0, sizeof(uint32_t), st) != hipSuccess) {
    qrlsh_set_error("qrlsh_check_csr: hipMemsetAsync failed");
    return QRLSH_EHIP;
  }
  const int64_t m = (nq > nnz ? nq : nnz) + 1;
  QR_LAUNCH("check_csr", check_csr_kernel, dim3((unsigned)ceil_div64(m, 256)), dim3(256), 0, st, offsets, rows, nq, nnz, D,
            flags_out);
  QR_LAUNCH_CHECK("qrlsh_check_csr");
  return QRLSH_OK;
}

static int pick_qpb(int P) {
  // LDS image = qpb * (P + 2) * 2 bytes; keep it <= 32 KiB so several workgroups share a CU
  int qpb = 64;
  while (qpb > 4 && (size_t)qpb * (P + 2) * 2 > 32768) qpb >>= 1;
  return qpb;
}

template <typename TabT>
static int launch_minhash(const int64_t *offsets, const int32_t *rows, int64_t nq, const void *perm_t, int P,
                          int P_stride, int D, int32_t *sig, uint16_t *sig16, int64_t *norm2, uint64_t *keys, int b,
                          int r, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(TabT);
  const int lanes = (P_stride + VEC - 1) / VEC;
  const TabT *tab = static_cast<const TabT *>(perm_t);
  const dim3 block(256);
  const size_t smem_group = keys ? (size_t)64 * (P + 2) * 2 : 0;
  if (lanes <= 64 && smem_group <= 65536 && D <= (1 << 24) && (size_t)D * P_stride * sizeof(TabT) < (1ull << 32)) {
    // group-per-query kernel: 64 queries per workgroup
    const dim3 grid((unsigned)ceil_div64(nq, 64));
#define QR_MHG(LPR_)                                                                                            \
  QR_LAUNCH("minhash", (minhash_group_kernel<TabT, LPR_>), grid, block, smem_group, st, offsets, rows, nq, tab, P, \
            P_stride, sig, sig16, norm2, keys, b, r)
    if (lanes <= 4) QR_MHG(4);
    else if (lanes <= 8) QR_MHG(8);
    else if (lanes <= 16) QR_MHG(16);
    else if (lanes <= 32) QR_MHG(32);
    else QR_MHG(64);
#undef QR_MHG
    QR_LAUNCH_CHECK("qrlsh_minhash");
    return QRLSH_OK;
  }
  // very long signatures: wave-per-query kernel with a column loop
  const int qpb = pick_qpb(P);
  const size_t smem = keys ? (size_t)qpb * (P + 2) * 2 : 0;
  const dim3 grid((unsigned)ceil_div64(nq, qpb));
  QR_LAUNCH("minhash", (minhash_kernel<TabT, 64>), grid, block, smem, st, offsets, rows, nq, tab, P, P_stride, sig,
            sig16, norm2, keys, b, r, qpb);
  QR_LAUNCH_CHECK("qrlsh_minhash");
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_minhash(const int64_t *offsets, const int32_t *rows, int64_t nq, const void *perm_t,
                               int32_t perm_dtype, int32_t P, int32_t P_stride, int32_t D, int32_t *sig_out,
                               uint16_t *sig16_out, int64_t *norm2_out, uint64_t *keys_out, int32_t b,
                               void *stream) {
  QR_CHECK_ARG(nq >= 0 && P > 0 && D > 0 && P_stride >= P, "qrlsh_minhash: bad sizes nq=%lld P=%d P_stride=%d D=%d",
               (long long)nq, P, P_stride, D);
  QR_CHECK_ARG(nq == 0 || (offsets && perm_t && (sig_out || sig16_out)), "qrlsh_minhash: null pointer");
  QR_CHECK_ARG(!sig16_out || (D <= 65535 && perm_dtype == QRLSH_PERM_U16),
               "qrlsh_minhash: the compact uint16 signature needs D <= 65535 and a uint16 table (D=%d)", D);
  QR_CHECK_ARG(((uintptr_t)sig16_out & 15) == 0, "qrlsh_minhash: 16-B alignment");
  QR_CHECK_ARG(perm_dtype == QRLSH_PERM_U16 || perm_dtype == QRLSH_PERM_I32, "qrlsh_minhash: bad perm_dtype %d",
               perm_dtype);
  const int esz = perm_dtype == QRLSH_PERM_U16 ? 2 : 4;
  QR_CHECK_ARG(((size_t)P_stride * esz) % 16 == 0, "qrlsh_minhash: P_stride*elem (%d*%d) must be a multiple of 16 B",
               P_stride, esz);
  QR_CHECK_ARG(perm_dtype != QRLSH_PERM_U16 || D <= 65536, "qrlsh_minhash: uint16 table needs D <= 65536 (D=%d)", D);
  QR_CHECK_ARG(((uintptr_t)perm_t & 15) == 0 && ((uintptr_t)sig_out & 15) == 0, "qrlsh_minhash: 16-B alignment");
  int r = 0;
  if (keys_out) {
    QR_CHECK_ARG(b > 0 && P % b == 0, "qrlsh_minhash: signature length %d not divisible by b=%d", P, b);
    r = P / b;
  }
  if (nq == 0) return QRLSH_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (perm_dtype == QRLSH_PERM_U16)
    return launch_minhash<uint16_t>(offsets, rows, nq, perm_t, P, P_stride, D, sig_out, sig16_out, norm2_out, keys_out,
                                    b, r, st);
  return launch_minhash<int32_t>(offsets, rows, nq, perm_t, P, P_stride, D, sig_out, nullptr, norm2_out, keys_out, b,
                                 r, st);
}

QRLSH_EXPORT int qrlsh_band_keys(const int32_t *sig, int64_t nq, int32_t P, int32_t b, uint64_t *keys_out,
                                 int64_t *norm2_out, void *stream) {
  QR_CHECK_ARG(nq >= 0 && P > 0 && (nq == 0 || (sig && keys_out)), "qrlsh_band_keys: bad arguments");
  QR_CHECK_ARG(b > 0 && P % b == 0, "qrlsh_band_keys: signature length %d not divisible by b=%d", P, b);
  const int r = P / b;
  if (nq == 0) return QRLSH_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int qpb = pick_qpb(P);
  QR_LAUNCH("band_keys", band_keys_kernel, dim3((unsigned)ceil_div64(nq, qpb)), dim3(256), (size_t)qpb * (P + 2) * 2, st,
                     sig, nq, P, b, r, keys_out, qpb);
  QR_LAUNCH_CHECK("qrlsh_band_keys");
  if (norm2_out) return qrlsh_row_norms(sig, nq, P, norm2_out, stream);
  return QRLSH_OK;
}

QRLSH_EXPORT int qrlsh_row_norms(const int32_t *sig, int64_t nq, int32_t P, int64_t *norm2_out, void *stream) {
  QR_CHECK_ARG(nq >= 0 && P > 0 && (nq == 0 || (sig && norm2_out)), "qrlsh_row_norms: bad arguments");
  if (nq == 0) return QRLSH_OK;
  if (P >= 2048) {
    QR_LAUNCH("row_norms", row_norms_kernel<true>, dim3((unsigned)(nq < 65536 ? nq : 65536)), dim3(256), 0,
              static_cast<hipStream_t>(stream), sig, nq, P, norm2_out);
  } else {
    const int64_t blocks = ceil_div64(nq, 4);
    QR_LAUNCH("row_norms", row_norms_kernel<false>, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
              static_cast<hipStream_t>(stream), sig, nq, P, norm2_out);
  }
  QR_LAUNCH_CHECK("qrlsh_row_norms");
  return QRLSH_OK;
}
