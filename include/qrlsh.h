/*
 * qrlsh.h -- C ABI of libqrlsh.so, the MI355X (gfx950) implementation of the
 * MinHash-LSH candidate-generation + pair-scoring hot path of
 * wamuumu/query-recommendation-system (lsh.py, recommender.py:105-214).
 *
 * The reference is pure Python and has no FFI of its own; this header is the
 * boundary its Python call surface (lsh.LSH, Recommender.compute_signatures /
 * compute_querySimilarities) binds through ctypes.  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - the caller owns every buffer (torch tensors in the Python host layer); the
 *     library never allocates device memory;
 *   - variable-size outputs are count-then-fill: *_count leaves the size in a
 *     device word the caller reads back, *_fill writes into caller memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls
 *     are asynchronous on it and safe under hipGraph capture (no allocation, no sync);
 *   - return value: 0 (QRLSH_OK) or a negative QRLSH_E* code; qrlsh_last_error()
 *     gives a thread-local message for the last failure on this thread.
 */
#ifndef QRLSH_H
#define QRLSH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QRLSH_OK 0
#define QRLSH_EINVAL (-1)       /* bad argument (incl. P % b != 0: lsh.py:20 asserts) */
#define QRLSH_EHIP (-2)         /* a HIP runtime call failed */
#define QRLSH_EUNSUPPORTED (-3) /* shape outside what the kernels cover */
#define QRLSH_EWORKSPACE (-4)   /* workspace too small */

#define QRLSH_PERM_U16 0 /* permutation table element = uint16 (D <= 65536) */
#define QRLSH_PERM_I32 1 /* permutation table element = int32 */

#define QRLSH_SIG_I32 0 /* signature rows int32 [nq][P] (the reference's values) */
#define QRLSH_SIG_U16 1 /* compact rows uint16 [nq][P], 0xFFFF = -1; only valid when D <= 65535 */

#define QRLSH_SORT_MIX 1u  /* radix digits are taken from mix64(key) (grouping sort) */
#define QRLSH_SORT_IOTA 2u /* first pass synthesises vals = index within the batch */
#define QRLSH_SORT_FOLD 4u /* digits from (key >> 32) << w | (key & (2^w - 1)), w = aux:
                              sorts pairs i << 32 | j over [0, 2w) in ceil(2w / 8) passes */
#define QRLSH_SORT_OWNER 8u /* ONE pass whose digit is (key >> bit_lo) / aux: groups words by the rank
                               that owns the id starting at bit_lo (shards of aux ids, <= 256 ranks) */

#define QRLSH_SORT_HOST 16u /* ONE pass over pair words i << 32 | j whose digit is the rank that scores the pair
                               in the sharded driver: the owner (shards of aux ids) of i or of j, chosen by
                               the top bit of mix64(pair) -- an even split of the pairs for any id structure */

int qrlsh_version(void);
const char *qrlsh_last_error(void);

/* 64-bit bijective mixer used by QRLSH_SORT_MIX (host copy, for tests) */
uint64_t qrlsh_mix64_host(uint64_t x);

/* ---- a1: MinHash signatures ------------------------------------------------
 * Replaces Recommender.compute_signatures, recommender.py:105-143:
 *     sig[q][p] = min_{d in A(q)} perm_p[d],  -1 when A(q) is empty.
 * offsets[nq+1] / rows[nnz] : CSR answer sets (compute_shingles' output, :68-103)
 * perm_t : the P permutations TRANSPOSED, [D][P_stride] (row d = the P permuted
 *          indices of table row d), element type perm_dtype; P_stride >= P, and
 *          P_stride * sizeof(element) a multiple of 16.
 * sig_out   [nq][P] int32 (row-major; the reference returns the same matrix as int64), or NULL
 * sig16_out [nq][P] uint16 compact rows (0xFFFF = -1), or NULL; needs D <= 65535 and a uint16
 *           table.  At least one of sig_out / sig16_out.  Half the bytes for qrlsh_score_pairs.
 * norm2_out [nq] int64 = sum_p sig^2 (exact), or NULL          -- feeds qrlsh_score_pairs
 * keys_out  [b][nq] uint64 band keys (see qrlsh_band_keys), or NULL -- fused a2
 * PRECONDITION (not checked by qrlsh_minhash itself, which gathers table row rows[k] directly): offsets[0] = 0,
 * offsets non-decreasing, offsets[nq] = nnz and 0 <= rows[k] < D.  qrlsh_check_csr tests exactly that on the
 * device: *flags_out (device uint32) = 0 when it holds, else bit 0 = bad first / last offset, bit 1 = offsets
 * decrease, bit 2 = a row id out of range.
 */
int qrlsh_check_csr(const int64_t *offsets, const int32_t *rows, int64_t nq, int64_t nnz, int32_t D,
                    uint32_t *flags_out, void *stream);
int qrlsh_minhash(const int64_t *offsets, const int32_t *rows, int64_t nq, const void *perm_t,
                  int32_t perm_dtype, int32_t P, int32_t P_stride, int32_t D, int32_t *sig_out,
                  uint16_t *sig16_out, int64_t *norm2_out, uint64_t *keys_out, int32_t b, void *stream);

/* ---- a2: band keys -----------------------------------------------------------
 * Replaces LSH.make_subvecs / compute_buckets' key construction, lsh.py:17-38:
 * band i of a signature = its values [i*r, (i+1)*r) cast to int16 (:28) and joined
 * into a string (:33).  Equal strings <=> equal int16 tuples, so for r <= 4
 *     key = sum_k (sig[i*r + k] & 0xFFFF) << (16 * k)
 * is an exact bucket id.  For r > 4 ("wide bands") the key is a 64-bit hash of the tuple (the all
 * -1 tuple still maps to ~0): bucket ids are then exact only up to hash collisions, and the
 * caller must pass the unique pairs through qrlsh_verify_pairs (below) to stay exact.
 * keys_out is band-major [b][nq].  norm2_out optional.  Returns QRLSH_EINVAL if P % b != 0.
 */
int qrlsh_band_keys(const int32_t *sig, int64_t nq, int32_t P, int32_t b, uint64_t *keys_out,
                    int64_t *norm2_out, void *stream);

/* ---- radix sort (replaces the dict-of-lists buckets, lsh.py:9-15,31-38) -------
 * Stable LSD radix sort of nbatch independent arrays of n uint64 keys (+ optional
 * uint32 payload), 8 bits per pass over bits [bit_lo, bit_hi) of the key (or of
 * mix64(key) with QRLSH_SORT_MIX: equal keys still end up adjacent, in payload
 * order, after 32 bits instead of 64).  Buffers a/b ping-pong; the return value
 * (>= 0) says where the result is: 0 = a, 1 = b.  vals_a/vals_b may both be NULL.
 */
size_t qrlsh_sort_workspace_bytes(int64_t n, int32_t nbatch);
int qrlsh_sort_u64(uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, int64_t n,
                   int32_t nbatch, int32_t bit_lo, int32_t bit_hi, uint32_t flags, uint64_t aux,
                   void *workspace, size_t workspace_bytes, void *stream);

/* split points of words grouped with QRLSH_SORT_OWNER: bounds_out[g] (device int64 [world+1]) = first
 * position whose owner (word >> bit_lo) / shard is >= g.  bit_lo = -1: pair words grouped with
 * QRLSH_SORT_HOST (owner = the scoring rank of the pair). */
int qrlsh_owner_bounds(const uint64_t *words, int64_t n, int32_t bit_lo, uint64_t shard, int32_t world,
                       int64_t *bounds_out, void *stream);

/* ---- a3: candidate pairs -------------------------------------------------------
 * Replaces LSH.get_candidates, lsh.py:40-55.  Input: per band, keys sorted with
 * QRLSH_SORT_MIX over the top hash_bits (8..32) bits of mix64(key), i.e. bits
 * [64 - hash_bits, 64), and their query ids ([b][nq] each).  Every run of
 * equal keys that is not the all -1 tuple (:47) yields all its (i < j) pairs (:49).
 * pairs are i << 32 | j; duplicates across bands are still present (the Python
 * set's job, :41) -- sort them and call qrlsh_unique_*.
 *   count: *total_out (device uint64) = number of pairs; workspace keeps per-block offsets
 *   fill : writes exactly that many pairs (capacity checked by the caller)
 */
size_t qrlsh_pairs_workspace_bytes(int64_t nq, int32_t b);
int qrlsh_pairs_count(const uint64_t *sorted_keys, int64_t nq, int32_t b, int32_t r, int32_t hash_bits,
                      void *workspace, size_t workspace_bytes, uint64_t *total_out, void *stream);
int qrlsh_pairs_fill(const uint64_t *sorted_keys, const uint32_t *sorted_ids, int64_t nq, int32_t b,
                     int32_t r, int32_t hash_bits, const void *workspace, uint64_t *pairs_out,
                     void *stream);

/* Fast form of the same step for keys straight from qrlsh_minhash / qrlsh_band_keys (band-major
 * [b][nq], NOT sorted): radix partition on the top part_bits (8..16) bits of mix64(key) -- one
 * pass for 8 bits, two above (tmp_keys / tmp_ids are the intermediate buffers, may be NULL for
 * 8) -- into part_keys / part_ids, then an LDS hash-group finish per (part, band).  Pick
 * part_bits so that nq / 2^part_bits is ~2-4 K.  count leaves {total pairs, overflow flag} in
 * total_overflow_out[2] (device uint64 x2); overflow != 0 means a part exceeded the LDS image
 * (6144 records; heavily skewed data): ignore the total and use the general path
 * (qrlsh_sort_u64 + qrlsh_pairs_count/fill) instead.  fill must follow a count on the same
 * workspace.  Same pairs as the general path, in a different (still duplicate-carrying) order.
 */
size_t qrlsh_bucket_workspace_bytes(int64_t nq, int32_t b, int32_t part_bits);
int qrlsh_bucket_pairs_count(const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids,
                             uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r,
                             int32_t part_bits, void *workspace, size_t workspace_bytes,
                             uint64_t *total_overflow_out, void *stream);
int qrlsh_bucket_pairs_fill(const uint64_t *part_keys, const uint32_t *part_ids, int64_t nq, int32_t b,
                            int32_t r, int32_t part_bits, void *workspace, uint64_t *pairs_out,
                            void *stream);
/* One-pass form of the two calls above: every part reserves its output range on a device cursor, so
 * the count pass (and its scan) disappears -- at the price of sizing pairs_out by a guess.  At most
 * `capacity` words of pairs_out are written; total_overflow_out[0] receives the exact number of pairs
 * whether or not they fitted (if it exceeds capacity: allocate that many and call again), [1] the same
 * oversized-part flag as qrlsh_bucket_pairs_count.  The pairs come out in no particular order.
 * part_keys / part_ids must hold qrlsh_bucket_part_words(nq, b, part_bits) words and, for part_bits > 8,
 * tmp_keys / tmp_ids qrlsh_bucket_tmp_words(...): the partition is ONE kernel per level (one level for
 * part_bits = 8, two of about part_bits / 2 bits each beyond) that gives every part a fixed region of ONE LDS image
 * of the finish and reserves room in it with an atomic per (tile, part) -- no histogram pass, no scan, no bounds
 * search.  Behind the regions the same buffers hold an OVERFLOW POOL (1/16 of the records, at least 1 M): a part
 * swollen by a popular key (lsh.py:42-49 makes a bucket of m queries m(m-1)/2 pairs whatever m is; at 100 M queries
 * over 32768 table rows m reaches ~20 000) spills there and is worked in blocks of one image; only a part beyond
 * qrlsh_set_big_part_limit records (default 16 images = 98 304), more than 4096 such parts per band group, or an
 * exhausted pool raise the overflow flag.  Reserved: ~1.4 - 2 x the b * nq records.  These buffers are scratch:
 * what they hold afterwards is mix64(key) (a bijection of the keys, which is all the pairing needs), not the keys,
 * and the records of empty bands are gone. */
size_t qrlsh_bucket_part_words(int64_t nq, int32_t b, int32_t part_bits);
/* records a part beyond the LDS image may hold and stay on the partition path (<= 0 or beyond the maximum: the
 * default, 98 304); returns the previous limit.  Process-wide; a tuning / test knob. */
int64_t qrlsh_set_big_part_limit(int64_t records);
size_t qrlsh_bucket_tmp_words(int64_t nq, int32_t b, int32_t part_bits);
int qrlsh_bucket_pairs_emit(const uint64_t *keys, uint64_t *part_keys, uint32_t *part_ids,
                            uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r,
                            int32_t part_bits, void *workspace, size_t workspace_bytes,
                            uint64_t *pairs_out, uint64_t capacity, uint64_t *total_overflow_out,
                            void *stream);

/* qrlsh_bucket_pairs_emit with the keys of band t, query q at
 *     keys[(q / key_chunk) * key_chunk_stride + t * key_band_stride + q % key_chunk]
 * -- what a band-partitioned all-to-all delivers ([rank][band][queries of that rank]: key_chunk = queries
 * per rank, key_band_stride = key_chunk, key_chunk_stride = bands * key_chunk), so the multi-GPU driver
 * needs no transposing copy; key_chunk_stride may exceed bands * key_chunk (a band range read out of a buffer
 * that holds more bands per rank).  key_chunk = 0: plain [b][nq].  Chunked keys need nq < 2^32. */
int qrlsh_bucket_pairs_emit_chunked(const uint64_t *keys, int64_t key_chunk, int64_t key_chunk_stride,
                                    int64_t key_band_stride, uint64_t *part_keys, uint32_t *part_ids,
                                    uint64_t *tmp_keys, uint32_t *tmp_ids, int64_t nq, int32_t b, int32_t r,
                                    int32_t part_bits, void *workspace, size_t workspace_bytes,
                                    uint64_t *pairs_out, uint64_t capacity, uint64_t *total_overflow_out,
                                    void *stream);

/* unique of a sorted uint64 array (count-then-fill) */
size_t qrlsh_compact_workspace_bytes(int64_t n);
int qrlsh_unique_count(const uint64_t *sorted, int64_t n, void *workspace, size_t workspace_bytes,
                       uint64_t *total_out, void *stream);
int qrlsh_unique_fill(const uint64_t *sorted, int64_t n, const void *workspace, uint64_t *out,
                      void *stream);

/* Sorted unique pairs from pairs that are only GROUPED BY i (qrlsh_sort_u64 over bits [32, 32 + id_bits)
 * of the emitted i << 32 | j words: about half the passes of the full (i, j) sort): every row -- the
 * pairs of one i, tens of words -- is de-duplicated (hash set) and ordered by j in LDS.  Replaces the Python set of
 * lsh.py:41,53 like qrlsh_unique_*, same result.  group_bits > 0: the words are ordered by
 * i >> group_bits only (a row is then 2^group_bits consecutive i; every id < 2^id_bits and group_bits +
 * id_bits <= 32) -- at 2^20 ids that is two radix passes instead of three.  count: tmp is scratch of n
 * words; leaves {number of
 * unique pairs, overflow flag} in total_overflow_out[2] (device uint64 x2); rows that do not fit a
 * workgroup's chunk image (more than ~1024 pairs past a 2048-word boundary) get a workgroup of their own;
 * overflow != 0 means one i has more than 12288 emitted pairs: ignore the total and use the general
 * path (full qrlsh_sort_u64 + qrlsh_unique_*) on the same words.  fill follows a count on the same
 * tmp / workspace and writes exactly `total` words, ascending.
 */
size_t qrlsh_row_unique_workspace_bytes(int64_t n);
int qrlsh_row_unique_count(const uint64_t *grouped, int64_t n, int32_t group_bits, int32_t id_bits,
                           uint64_t *tmp, void *workspace, size_t workspace_bytes, uint64_t *total_overflow_out,
                           void *stream);
int qrlsh_row_unique_fill(const uint64_t *tmp, int64_t n, const void *workspace, uint64_t *out, void *stream);

/* The same result from words grouped by i >> group_bits with group_bits up to 8 (REGIONS of 2^group_bits
 * consecutive queries, a few thousand words each: at 2^24 ids and group_bits = 8 the grouping sort needs two
 * radix passes, not three): one workgroup per region streams its words through an LDS hash set of the 32-bit
 * values (i's low bits, j), counts the distinct ones per i, and places each by the number of smaller ones of
 * its own i.  Only the DISTINCT pairs of a region are bounded, not its words: about 5 K in the main kernel,
 * about 11 K in the big-image kernel that takes over the regions beyond that (very popular queries).  nids = number of query
 * ids (regions = ceil(nids / 2^group_bits)); needs group_bits + id_bits <= 32, and nids < 2^id_bits when it is
 * exactly 32.  count / fill / overflow as qrlsh_row_unique_*.  This is the default de-duplication of the
 * pipeline; qrlsh_row_unique_* remains for id widths that leave no room for group bits. */
size_t qrlsh_region_unique_workspace_bytes(int64_t nids, int32_t group_bits);
int qrlsh_region_unique_count(const uint64_t *grouped, int64_t n, int32_t group_bits, int32_t id_bits, int64_t nids,
                              uint64_t *tmp, void *workspace, size_t workspace_bytes, uint64_t *total_overflow_out,
                              void *stream);
int qrlsh_region_unique_fill(const uint64_t *tmp, int64_t n, int32_t group_bits, int64_t nids, const void *workspace,
                             uint64_t *out, void *stream);

/* The same de-duplication WITHOUT the grouping sort: the region finish needs the words grouped by region (i >> group_bits)
 * and nothing about the order inside a group, so the stable radix passes (a histogram pass, a scan and a scatter per 8
 * bits) are replaced by a most-significant-digit-first partition into FIXED regions with one atomic reservation per
 * (tile, digit): one read and one write of the words per level, two levels for up to 65536 regions.
 * qrlsh_pair_regions_scatter deals the n emitted words into regions[r * cap ..) (counts[r] words each; cap =
 * qrlsh_pair_regions_cap, buffers of qrlsh_pair_regions_words / _tmp_words words, counts of _count + 256 uint32);
 * words_per_query: what a query of the populated id range emits on average (0: n / nids) -- sizes the regions (3 x the
 * mean + 4096: i is the smaller id of a pair, so low ids carry up to twice the mean).  *overflow_out != 0: a region
 * outgrew its capacity (group with qrlsh_sort_u64 instead).  qrlsh_pair_regions_words returns 0 when the id space has
 * more than 65536 regions (not served).  qrlsh_region_unique_count_regions is qrlsh_region_unique_count on those
 * regions (tmp: as many words as the region buffer); qrlsh_region_unique_fill follows it as usual. */
size_t qrlsh_pair_regions_words(int64_t n, int64_t nids, int32_t group_bits, double words_per_query);
size_t qrlsh_pair_regions_tmp_words(int64_t n, int64_t nids, int32_t group_bits, double words_per_query);
int64_t qrlsh_pair_regions_cap(int64_t n, int64_t nids, int32_t group_bits, double words_per_query);
int64_t qrlsh_pair_regions_count(int64_t n, int64_t nids, int32_t group_bits, double words_per_query);
int qrlsh_pair_regions_scatter(const uint64_t *words, int64_t n, int32_t group_bits, int64_t nids, double words_per_query,
                               uint64_t *tmp_regions, uint64_t *regions, uint32_t *counts, uint32_t *overflow_out,
                               void *stream);
int qrlsh_region_unique_count_regions(const uint64_t *regions, const uint32_t *counts, int64_t cap, int64_t n,
                                      int32_t group_bits, int32_t id_bits, int64_t nids, uint64_t *tmp, void *workspace,
                                      size_t workspace_bytes, uint64_t *total_overflow_out, void *stream);

/* ---- a5: pair scoring ------------------------------------------------------------
 * Replaces the cosine of recommender.py:203-204 for one candidate pair:
 *     np.around(cosine_similarity([sig_i, sig_j])[0][1], 3)
 * computed as exact integer dot / (sqrt(norm2_i) * sqrt(norm2_j)) in float64 (0 if a
 * norm is 0, as sklearn's normalize does), milli = rint(cos * 1000) so that
 * milli / 1000.0 == np.around(cos, 3).  sig is int32 [n][P] (QRLSH_SIG_I32) or the compact
 * uint16 rows written by qrlsh_minhash (QRLSH_SIG_U16).
 * norm2 may be NULL: the two squared norms are then summed from the rows inside the kernel (same exact integers;
 * slower on MI355X -- 5.1 vs 4.4 ms on 45 M pairs: the kernel is short of integer-multiply throughput, not of the
 * 64-byte sector a precomputed norm costs).
 * cos_out (double, unrounded) and edge_out are optional.  edge_out[2n] receives the two
 * directed top-K sort keys of each pair:
 *     src << (id_bits + 11) | (1000 - milli) << id_bits | dst        (needs id_bits <= 26)
 * or, when edge_dst_out[2n] is given (any id width, "wide ids"),
 *     edge_out = src << 11 | (1000 - milli),  edge_dst_out = dst   (a key + payload record).
 */
int qrlsh_row_norms(const int32_t *sig, int64_t nq, int32_t P, int64_t *norm2_out, void *stream);
/* exact candidate test (needed only for r = P / b > 4): flags_out[t] = 1 iff pair t shares a band
 * whose r int16 values are all equal and not all -1 (lsh.py:31-53) */
int qrlsh_verify_pairs(const void *sig, int32_t sig_dtype, int32_t P, int32_t b, const uint64_t *pairs,
                       int64_t n, uint8_t *flags_out, void *stream);
int qrlsh_score_pairs(const void *sig, int32_t sig_dtype, const int64_t *norm2, int32_t P,
                      const uint64_t *pairs, int64_t n, int32_t *milli_out, double *cos_out,
                      uint64_t *edge_out, int32_t id_bits, uint32_t *edge_dst_out, void *stream);

/* ---- a5: per-query top-K -----------------------------------------------------------
 * Replaces argsort(values)[::-1][:K] per query, recommender.py:206-210, on the edge keys
 * sorted ascending (so: src, value descending, dst ascending -- the documented
 * tie-break; the reference's own tie order is arbitrary).  count-then-fill; the output
 * is COO (src, dst, milli), at most K rows per src.  Wide-id edges: id_bits = 0 and sorted_dst =
 * the payload sorted along with the keys (NULL for the packed format).
 */
int qrlsh_topk_count(const uint64_t *sorted_edges, int64_t n_edges, int32_t K, int32_t id_bits,
                     void *workspace, size_t workspace_bytes, uint64_t *total_out, void *stream);
int qrlsh_topk_fill(const uint64_t *sorted_edges, const uint32_t *sorted_dst, int64_t n_edges, int32_t K,
                    int32_t id_bits, const void *workspace, int32_t *src_out, int32_t *dst_out,
                    int32_t *milli_out, void *stream);
/* the same with src_base added to every src written (edge keys whose src field is relative to a rank's
 * first query id: qrlsh_edges_localize) */
int qrlsh_topk_fill_based(const uint64_t *sorted_edges, const uint32_t *sorted_dst, int64_t n_edges, int32_t K,
                          int32_t id_bits, int64_t src_base, const void *workspace, int32_t *src_out,
                          int32_t *dst_out, int32_t *milli_out, void *stream);

/* Select form of the same cut, without sorting the directed edges: the forward edges of a query are its run of
 * the (sorted) scored pair list; only the n reverse words qrlsh_score_pairs_rev writes (rev_out[t] =
 * j << (id_bits + 11) | inv << id_bits | i, or key + payload j << 11 | inv, i for ids beyond 26 bits) are
 * sorted, stably, on j's bits alone -- ceil(id_bits / 8) passes over n words instead of ceil((id_bits + 11) / 8)
 * over 2n.  Every directed edge then counts the edges of its query's two runs that order before it (value
 * descending, neighbour id ascending; at most K of them are looked for) and, if fewer than K do, lands at that
 * rank of its query's output row.  count: *total_out = number of edges kept; fill writes the same (src, dst,
 * milli) COO qrlsh_topk_fill does.  nq = number of query ids; n < 2^31 pairs.
  * pairs == NULL (and milli == NULL in _fill): the lists are made of the n reverse words alone, whatever scored them
 * (the sharded driver: every directed edge a rank receives, re-based to its id range by qrlsh_edges_localize, IS
 * such a word with src = one of its own queries; sorted on the src bits only); packed words then need
 * src < 2^(53 - id_bits) instead of id_bits <= 26.
 */
int qrlsh_score_pairs_rev(const void *sig, int32_t sig_dtype, const int64_t *norm2, int32_t P,
                          const uint64_t *pairs, int64_t n, int32_t *milli_out, uint64_t *rev_out, int32_t id_bits,
                          uint32_t *rev_dst_out, void *stream);
/* 1 (default): compact rows of 128 / 256 values with precomputed norms are scored by the run form (16 consecutive pairs
 * per 16-lane group, first row kept while it does not change); 0: by the generic form.  Same results bit for bit; an
 * A/B and test knob.  Returns the previous setting (-1: not yet decided, i.e. the default). */
int qrlsh_set_score_runs(int on);
size_t qrlsh_topk_select_workspace_bytes(int64_t nq);
int qrlsh_topk_select_count(const uint64_t *pairs, int64_t n, const uint64_t *rev_sorted, const uint32_t *rev_dst,
                            int64_t nq, int32_t K, int32_t id_bits, void *workspace, size_t workspace_bytes,
                            uint64_t *total_out, void *stream);
int qrlsh_topk_select_fill(const uint64_t *pairs, const int32_t *milli, int64_t n, const uint64_t *rev_sorted,
                           const uint32_t *rev_dst, int64_t nq, int32_t K, int32_t id_bits, const void *workspace,
                           int32_t *src_out, int32_t *dst_out, int32_t *milli_out, void *stream);

/* ---- multi-GPU glue (one process per GPU; qrlsh/dist.py) ----------------------------------------------
 * remap_pairs: an owner scores pairs (i local, j anywhere) against a row table [its nql local rows | the
 * fetched remote rows]; out[t] = (i - q0) << 32 | slot(j), slot(j) = j - q0 for a local j, else nql + the
 * position of j in `need` (the ascending global ids of the fetched rows).
 * pair_edges: the directed edge keys of scored pairs, forward (src = i) and reverse (src = j) in separate
 * arrays: packed (id_bits > 0, as qrlsh_score_pairs writes them) or key + payload (id_bits == 0).
 */
int qrlsh_remap_pairs(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, const uint64_t *need,
                      int64_t n_need, uint64_t *out, void *stream);
int qrlsh_pair_edges(const uint64_t *pairs, const int32_t *milli, int64_t n, int32_t id_bits,
                     uint64_t *fwd_out, uint64_t *rev_out, uint32_t *fwd_dst_out, uint32_t *rev_dst_out,
                     void *stream);
/* (rev_out == NULL: both edges of pair t go to fwd_out[2t], fwd_out[2t+1] -- and fwd_dst_out likewise --
 * exactly the layout qrlsh_score_pairs' edge_out has.)
 *
 * Sharded scoring.  A pair is scored on the rank qr_pair_host names (QRLSH_SORT_HOST): the owner of one of
 * its two queries, so at most one of its signature rows is remote.
 *   idset_*: the set of remote ids a rank's pairs touch, as a bitmap over the nids global ids plus its rank
 *     structure, kept in `workspace` (qrlsh_idset_workspace_bytes).  build marks both endpoints of every
 *     pair outside [q0, q0 + nql) and leaves in bounds_out[g] (device int64 [world + 1]) the number of marked
 *     ids below g * shard -- the per-owner request sizes, bounds_out[world] = the total; list writes the ids
 *     ascending (the row-fetch request); remap rewrites each pair as slot(i) << 32 | slot(j) into the row
 *     table [nql local rows | fetched rows in id order]: slot(x) = x - q0 if local, else nql + rank of x.
 *   gather_rows: the answer to such a request on the owning rank: rows_out[k] = signature row ids[k] - q0
 *     (row_bytes bytes, a multiple of 16), norms_out[k] its norm.
 *   score_pairs_split: qrlsh_score_pairs against a row table in two pieces -- row x < split_rows from
 *     sig / norm2, the others (x - split_rows) from sig_b / norm2_b -- so the fetched rows are never copied.
 *   edges_localize: directed edges that arrived at the owner of their src (packed, or key + payload when
 *     edge_dst is given) re-based to its id range: (src - q0) << (id_bits + 11) | inv << id_bits | dst, which
 *     fits one word whenever bits(nql) + 11 + id_bits <= 64; the top-K sort then orders (src, value, dst)
 *     completely (edges from several scoring ranks arrive in no useful order). */
size_t qrlsh_idset_workspace_bytes(int64_t nids);
int qrlsh_idset_build(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, int64_t nids, int64_t shard,
                      int32_t world, void *workspace, size_t workspace_bytes, int64_t *bounds_out, void *stream);
int qrlsh_idset_list(const void *workspace, int64_t nids, uint64_t *ids_out, void *stream);
int qrlsh_idset_remap(const uint64_t *pairs, int64_t n, int64_t q0, int64_t nql, const void *workspace,
                      int64_t nids, uint64_t *out, void *stream);
int qrlsh_gather_rows(const void *sig, int64_t row_bytes, const int64_t *norm2, const uint64_t *ids, int64_t n,
                      int64_t q0, void *rows_out, int64_t *norms_out, void *stream);
int qrlsh_score_pairs_split(const void *sig, const int64_t *norm2, int64_t split_rows, const void *sig_b,
                            const int64_t *norm2_b, int32_t sig_dtype, int32_t P, const uint64_t *pairs,
                            int64_t n, int32_t *milli_out, void *stream);
int qrlsh_edges_localize(const uint64_t *edges, const uint32_t *edge_dst, int64_t n, int32_t id_bits, int64_t q0,
                         int64_t nql, uint64_t *out, void *stream);

/* ---- N2: answer sets (the producer of the hot path's input) ---------------------------------
 * Replaces Recommender.compute_shingles, recommender.py:68-103, for queries that are
 * conjunctions of attribute=value: bitmaps[row][words_per_row] holds one bit per table row for
 * every (feature, value) of the table (bit i of word w = table row 32*w + i; D = table rows);
 * qrows[q][f] names the bitmap row of query q's value for feature f, -1 = unconstrained (a value
 * absent from the table points at an all-zero row).  count: sizes_out[q] = |A(q)|;
 * fill: rows_out[offsets[q] ..) = the matching table rows, ascending (offsets = exclusive scan
 * of the sizes) -- the CSR qrlsh_minhash consumes.  nfeat <= 64.
 */
int qrlsh_answer_sets_count(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                            const int32_t *qrows, int64_t nq, int32_t nfeat, int32_t *sizes_out,
                            void *stream);
int qrlsh_answer_sets_fill(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                           const int32_t *qrows, int64_t nq, int32_t nfeat, const int64_t *offsets,
                           int32_t *rows_out, void *stream);

/* one-sweep form of the two calls above: sweep = sizes + the first 64 row ids of every query
 * parked in slots_out[nq][64]; after the exclusive scan of the sizes, compact moves the slots of the
 * queries with <= 64 rows to their CSR positions.  Queries with more rows (sizes_out > 64) still
 * need qrlsh_answer_sets_fill (pass it a qrows/offsets subset, or call it for all). */
int qrlsh_answer_sets_sweep(const uint32_t *bitmaps, int64_t words_per_row, int64_t D,
                            const int32_t *qrows, int64_t nq, int32_t nfeat, int32_t *sizes_out,
                            int32_t *slots_out, void *stream);
int qrlsh_answer_sets_compact(const int32_t *slots, const int64_t *offsets, int64_t nq,
                              int32_t *rows_out, void *stream);

/* ---- N1: hybrid prediction loop (the consumer of the hot path's output) ---------------------
 * Replaces the per-cell loop of Recommender.compute_scores, recommender.py:301-331, and
 * weighted_average, recommender.py:36-47.  ratings int32 [nu][nq] (0 = missing).  Query
 * neighbours in CSR form (q_off[nq+1], q_idx, q_val = rounded cosine, i.e. milli / 1000.0) as
 * qrlsh_topk_* produce them; user neighbours padded [nu][ku] (u_idx = -1 past the end).
 * out[nu][nq] = the utility matrix with every zero cell replaced by round(blend) (0 when neither
 * side predicts).  float64 arithmetic, no FMA, round half to even.  sum_order picks the order of the two
 * np.sum calls inside weighted_average: QRLSH_SUM_PAIRWISE = numpy's pairwise sum (the reference run as
 * plain Python; the order the committed fixtures pin), QRLSH_SUM_SEQUENTIAL = one accumulator in index
 * order (what numba's nopython np.sum does where numba is installed; unpinned here).
 * Limits: ku <= 64 (QRLSH_EINVAL otherwise); a query with more than 64 neighbours sets *too_long_out
 * (device uint32, required) to 1; its cells are never walked (they get 0) and the result is not to be used -- the
 * caller reads the flag back.
 * kq > 0 with a workspace of qrlsh_predict_workspace_bytes(nu, nq, kq): the caller states the longest query list
 * (kq <= 64; a longer one sets the flag, as above) and the kernel that fits the data runs, same results from all:
 *   tile form (ku <= 32, every rating in 0 .. 255 -- the reference's are 0 .. 100): the matrix is transposed to
 *     bytes in the workspace, a workgroup owns 64 users x 16 queries, every list is read once per tile and the
 *     rating gathers are 64 consecutive bytes (query side) or stay inside one nu-byte row (user side);
 *   row form (nq <= 131072; what runs when a rating does not fit a byte -- decided on the device, no read-back):
 *     one workgroup per user slice, the lists transposed to [kq][nq] in the workspace, the user's row in LDS;
 *   cell form otherwise: one thread per cell over the transposed lists.
 * kq = 0 / workspace = NULL: one thread per cell over the lists in their CSR form.
 */
#define QRLSH_SUM_PAIRWISE 0
#define QRLSH_SUM_SEQUENTIAL 1
size_t qrlsh_predict_workspace_bytes(int64_t nu, int64_t nq, int32_t kq);
int qrlsh_predict(const int32_t *ratings, int64_t nu, int64_t nq, const int64_t *q_off,
                  const int32_t *q_idx, const double *q_val, const int32_t *u_idx, const double *u_val,
                  int32_t ku, double query_weight, double user_weight, double default_mean,
                  int32_t sum_order, int32_t *out, uint32_t *too_long_out, int32_t kq, void *workspace,
                  size_t workspace_bytes, void *stream);

/* ---- N4: user similarity, the part after the clustering ----------------------------------------
 * Recommender.compute_userSimilarities, recommender.py:263-288: inside a cluster every user's row is centred on
 * the mean of its non-zero ratings IN AN INTEGER ARRAY (the centred values are truncated toward zero), then
 * cosine of every pair of rows.  out[u][c] = ratings[u][c] == 0 ? 0 : (int)((double)ratings[u][c] - mean_u),
 * mean_u = (double)sum / (double)count over the non-zero ratings; row stride nq_stride >= nq (padding = 0).
 * The pairs of a cluster are qrlsh_bucket_pairs_emit on the labels (one band), their cosine qrlsh_score_pairs
 * on these rows, the per-user cut qrlsh_topk_select_* (qrlsh/users.py).  The clustering itself (:226-261) is
 * the reference's scikit-learn call and stays on the host.
 */
int qrlsh_center_rows(const int32_t *ratings, int64_t nu, int64_t nq, int64_t nq_stride, int32_t *out,
                      void *stream);

/* The clustering features of the same step (recommender.py:226-234: StandardScaler().fit_transform(ratings), then
 * PCA(min(r, c, 200)).fit(.).transform(.)) for a matrix with far more columns (queries) than rows (users): column
 * statistics as StandardScaler computes them (mean_out[nq]; inv_scale_out[nq] = 1 / scale, scale = sqrt(population
 * variance), 1 for constant columns) and the Gram matrix gram_out[nu][nu] = Z Z^T of the standardized matrix
 * Z = (ratings - mean) * inv_scale, in float64 on the matrix cores (v_mfma_f64_16x16x4_f64), the ratings read as
 * integers and standardized while they are staged (Z is never materialised), summed over column slices in a fixed
 * order (deterministic).  The PCA scores are U_k sqrt(lambda_k) of gram's eigen-decomposition (qrlsh/users.py).
 * workspace: qrlsh_user_gram_workspace_bytes(nu, nq) (the per-slice partial matrices). */
size_t qrlsh_user_gram_workspace_bytes(int64_t nu, int64_t nq);
int qrlsh_user_gram(const int32_t *ratings, int64_t nu, int64_t nq, double *mean_out, double *inv_scale_out,
                    double *gram_out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- multi-GPU, "sets" mode: answer sets of chosen queries out of the replicated per-shard CSR arrays ------------
 * Every rank holds every shard's answer sets as an all-gather delivered them: offs[world][nql + 1] (off_bytes = 4 or 8)
 * and rows[world][max_nnz] (row_bytes = 2: unsigned 16-bit row ids, tables of at most 65536 rows; or 4), shard g =
 * queries [g * nql, (g + 1) * nql).  For the n global query ids in `ids` (the remote queries a rank's pairs touch:
 * qrlsh_idset_list): _count writes offsets_out[n + 1] = exclusive scan of their set sizes ([n] = number of row ids:
 * the caller reads it back to allocate), _fill the row ids as int32 -- the CSR qrlsh_minhash takes.  Replaces a row
 * fetch from the owners: no signature row crosses a link.  Workspace: qrlsh_gather_sets_workspace_bytes(n). */
size_t qrlsh_gather_sets_workspace_bytes(int64_t n);
int qrlsh_gather_sets_count(const uint64_t *ids, int64_t n, const void *offs, int32_t off_bytes, int64_t nql,
                            int64_t world, uint64_t *offsets_out, void *workspace, size_t workspace_bytes,
                            void *stream);
int qrlsh_gather_sets_fill(const uint64_t *ids, int64_t n, const void *offs, int32_t off_bytes, const void *rows,
                           int32_t row_bytes, int64_t nql, int64_t max_nnz, const uint64_t *offsets,
                           int32_t *rows_out, void *stream);

/* ---- synthetic answer sets (bench / test input) ---------------
 * Bit-identical twin of oracle/qr_oracle.c:qro_synth_*: a pure function of (seed, q).
 * sizes: sizes_out[i] = |A(q0 + i)|; fill: rows at offsets[i] (offsets = exclusive scan).
 */
int qrlsh_synth_sizes(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total, int32_t cluster,
                      uint32_t D, const uint32_t *cdf, int32_t ncdf, uint32_t rep_thresh24,
                      int32_t *sizes_out, void *stream);
int qrlsh_synth_fill(uint64_t seed, int64_t q0, int64_t nq_local, int64_t nq_total, int32_t cluster,
                     uint32_t D, const uint32_t *cdf, int32_t ncdf, uint32_t rep_thresh24,
                     const int64_t *offsets, int32_t *rows_out, void *stream);

/* ---- optional per-kernel profiler ------------------------------------------------------
 * When enabled, every kernel launch of the library is bracketed by two HIP events recorded on
 * the launch stream.  qrlsh_prof_report waits for them and writes one "label count total_ms"
 * line per kernel label into buf_host; returns the number of labels.  enable(on) also clears
 * what was recorded so far; pause(1) / pause(0) stops / resumes the bracketing and keeps the
 * records (the events serialise back-to-back launches -- about 3 us per event, 10 % of a
 * config-2 step -- so bench.py brackets every Nth step of its timed region, not all of them).
 * bench.py uses this for the live roofline figures.
 */
int qrlsh_prof_enable(int on);
/* Intra-call overlap: qrlsh_bucket_pairs_emit* works its bands in groups that alternate between the caller's stream
 * and one auxiliary stream of the library (forked from / joined back into the caller's stream inside the call, so
 * the call stays ONE asynchronous operation on `stream`), letting a group's LDS-bound finish share the device
 * with the next group's memory-bound partition.  On by default; 0 (or the environment variable QRLSH_OVERLAP=0,
 * read on first use) runs everything on the caller's stream.  Steps bracketed by the profiler run serially. */
int qrlsh_set_overlap(int on);
int qrlsh_prof_pause(int paused);
int qrlsh_prof_report(char *buf_host, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* QRLSH_H */
