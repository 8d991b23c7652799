"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package must never do so.

Two restatements of the reference's hot path live here:

* the C one (oracle/qr_oracle.c -> libqroracle.so), reached through ctypes -- fast
  enough to check mid-size cases and to serve as the timed CPU baseline;
* `naive_*` pure-Python versions that keep the reference's own data structures
  (dict of string keys, itertools.combinations, set of tuples) for tiny cases --
  a second, independent witness for the C code.

Parity status: PINNED against tests/golden (captured from the reference by
tools/make_golden.py); see tests/test_oracle_golden.py.
"""
import ctypes
import math
import os
import subprocess
from itertools import combinations

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f64p = ctypes.POINTER(ctypes.c_double)


def build():
    """Compile oracle/qr_oracle.c with gcc (idempotent)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libqroracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.qro_candidates.restype = ctypes.c_int64
        L.qro_candidates_from_sig.restype = ctypes.c_int64
        L.qro_emitted_pairs.restype = ctypes.c_int64
        L.qro_topk.restype = ctypes.c_int64
        L.qro_filter_pair_host.restype = ctypes.c_int64
        L.qro_sort_unique_u64.restype = ctypes.c_int64
        L.qro_free.argtypes = [ctypes.c_void_p]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def max_threads():
    return int(lib().qro_max_threads())


def set_threads(n):
    lib().qro_set_threads(ctypes.c_int(int(n)))


# ---------------------------------------------------------------------------
# host-side pieces shared by every restatement
# ---------------------------------------------------------------------------
def legacy_permutations(seed, P, D):
    """P consecutive np.random.permutation(D) draws from the legacy global stream
    (recommender.py:120), made explicit: RandomState(seed) is the same stream as
    np.random.seed(seed).  Returns int32 [P][D]."""
    rs = np.random.RandomState(seed)
    out = np.empty((P, D), dtype=np.int32)
    for p in range(P):
        out[p] = rs.permutation(D)
    return out


def max_candidates(nq):
    """K = round(log_1.5 nq)  (recommender.py:151)."""
    return round(math.log(nq, 1.5))


def select_bands(P, thresh=0.2):
    """Largest b with P % b == 0, b % 10 == 0 and round((1/b)**(1/r), 2) >= thresh
    (recommender.py:153-163).  The reference falls through to an UnboundLocalError
    when nothing qualifies; here that is a ValueError."""
    for b in range(P, 0, -1):
        if P % b == 0 and b % 10 == 0:
            r = P / b
            if round((1 / b) ** (1 / r), 2) >= thresh:
                return b
    raise ValueError("no band count satisfies the reference rule for PERM=%d" % P)


# ---------------------------------------------------------------------------
# C restatement
# ---------------------------------------------------------------------------
def minhash(offsets, rows, perm):
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    P, D = perm.shape
    nq = len(offsets) - 1
    sig = np.empty((nq, P), dtype=np.int32)
    lib().qro_minhash(_p(offsets, c_i64p), _p(rows, c_i32p), ctypes.c_int64(nq), _p(perm, c_i32p),
                      ctypes.c_int32(P), ctypes.c_int32(D), _p(sig, c_i32p))
    return sig


def band_keys(sig, b):
    sig = np.ascontiguousarray(sig, dtype=np.int32)
    nq, P = sig.shape
    keys = np.empty((nq, b), dtype=np.uint64)
    rc = lib().qro_band_keys(_p(sig, c_i32p), ctypes.c_int64(nq), ctypes.c_int32(P), ctypes.c_int32(b),
                             _p(keys, c_u64p))
    if rc != 0:
        if P % b != 0:
            raise AssertionError("signature length %d not divisible by b=%d" % (P, b))
        raise ValueError("band width r=%d > 4 is not supported by the 64-bit key" % (P // b))
    return keys


def candidates(keys, r):
    """-> sorted unique uint64 array of (i << 32 | j), i < j"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    nq, b = keys.shape
    out = c_u64p()
    n = lib().qro_candidates(_p(keys, c_u64p), ctypes.c_int64(nq), ctypes.c_int32(b), ctypes.c_int32(r),
                             ctypes.byref(out))
    if n < 0:
        raise ValueError("bad arguments")
    arr = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
    lib().qro_free(out)
    return arr


def candidates_from_sig(sig, b):
    """candidates for ANY band width (buckets grouped by exact int16-tuple comparison):
    -> sorted unique uint64 array of (i << 32 | j), i < j"""
    sig = np.ascontiguousarray(sig, dtype=np.int32)
    nq, P = sig.shape
    if b <= 0 or P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))
    out = c_u64p()
    n = lib().qro_candidates_from_sig(_p(sig, c_i32p), ctypes.c_int64(nq), ctypes.c_int32(P), ctypes.c_int32(b),
                                      ctypes.byref(out))
    arr = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy()
    lib().qro_free(out)
    return arr


def filter_pair_host(pairs, shard, rank):
    """the words of `pairs` (order kept) that rank `rank` of a sharded run scores: the owner (shards of `shard`
    consecutive ids) of i or of j, picked by the top bit of mix64(pair) -- csrc/common.h: qr_pair_host"""
    pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
    out = np.empty(len(pairs), dtype=np.uint64)
    n = lib().qro_filter_pair_host(_p(pairs, c_u64p), ctypes.c_int64(len(pairs)), ctypes.c_uint32(shard),
                                   ctypes.c_uint32(rank), _p(out, c_u64p))
    return out[:n].copy()


def sort_unique(words):
    """sorted unique copy of a uint64 array (all cores)"""
    a = np.array(words, dtype=np.uint64, copy=True)
    n = lib().qro_sort_unique_u64(_p(a, c_u64p), ctypes.c_int64(len(a)))
    return a[:n].copy()


def emitted_pairs(keys, r):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    nq, b = keys.shape
    return int(lib().qro_emitted_pairs(_p(keys, c_u64p), ctypes.c_int64(nq), ctypes.c_int32(b),
                                       ctypes.c_int32(r)))


def score_pairs(sig, pairs, mode=1, want_cos=False):
    sig = np.ascontiguousarray(sig, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
    n = len(pairs)
    milli = np.empty(n, dtype=np.int32)
    cosv = np.empty(n, dtype=np.float64) if want_cos else None
    lib().qro_score_pairs(_p(sig, c_i32p), ctypes.c_int32(sig.shape[1]), _p(pairs, c_u64p),
                          ctypes.c_int64(n), ctypes.c_int32(mode), _p(milli, c_i32p),
                          _p(cosv, c_f64p) if want_cos else None)
    return (milli, cosv) if want_cos else milli


def topk(pairs, milli, K):
    """-> (src, dst, milli) int32 arrays, sorted by (src, value desc, dst asc), <= K per src"""
    pairs = np.ascontiguousarray(pairs, dtype=np.uint64)
    milli = np.ascontiguousarray(milli, dtype=np.int32)
    n = len(pairs)
    src = np.empty(2 * n, dtype=np.int32)
    dst = np.empty(2 * n, dtype=np.int32)
    val = np.empty(2 * n, dtype=np.int32)
    m = lib().qro_topk(_p(pairs, c_u64p), _p(milli, c_i32p), ctypes.c_int64(n), ctypes.c_int32(K),
                       _p(src, c_i32p), _p(dst, c_i32p), _p(val, c_i32p))
    return src[:m].copy(), dst[:m].copy(), val[:m].copy()


def query_similarities(offsets, rows, D, P, b, K, seed):
    """Whole hot path on the CPU: -> dict with sig, keys, pairs, milli, topk arrays."""
    perm = legacy_permutations(seed, P, D)
    sig = minhash(offsets, rows, perm)
    if P // b <= 4:
        keys = band_keys(sig, b)
        pairs = candidates(keys, P // b)
    else:
        keys = None
        pairs = candidates_from_sig(sig, b)
    milli = score_pairs(sig, pairs, mode=1)
    src, dst, val = topk(pairs, milli, K)
    return dict(sig=sig, keys=keys, pairs=pairs, milli=milli, src=src, dst=dst, val=val)


def sims_to_dict(src, dst, val):
    """(src, dst, milli) COO -> {q: {'indexes': int64[], 'values': float64[]}} as
    recommender.py:206-210 returns it (values = milli / 1000 == np.around(cos, 3))."""
    out = {}
    if len(src) == 0:
        return out
    cut = np.flatnonzero(np.diff(src)) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [len(src)]))
    for s, e in zip(starts, ends):
        out[int(src[s])] = {"indexes": dst[s:e].astype(np.int64),
                            "values": val[s:e].astype(np.float64) / 1000.0}
    return out


def answer_sets(columns, queries):
    """recommender.py:83-95: per query, AND of (column == value) over its constrained features
    ("" = unconstrained), indices of the surviving rows.  columns: list of 1-D str arrays;
    queries: (nq, nfeat) str array.  -> CSR (offsets int64, rows int32)."""
    columns = [np.asarray(c) for c in columns]
    D = len(columns[0])
    sets = []
    for q in range(len(queries)):
        cond = np.ones(D, dtype=bool)
        for ft in range(len(columns)):
            if queries[q][ft] != "":
                cond &= (columns[ft] == queries[q][ft])
        sets.append(np.flatnonzero(cond).astype(np.int32))
    offsets = np.zeros(len(sets) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in sets], out=offsets[1:])
    rows = np.concatenate(sets).astype(np.int32) if sets else np.zeros(0, np.int32)
    return offsets, rows


# ---------------------------------------------------------------------------
# synthetic answer sets (bench / test input; not part of the reference)
# ---------------------------------------------------------------------------
def poisson_cdf_u32(mean, n=64):
    """32-bit fixed-point CDF thresholds of Poisson(mean): size = #{k : u >= cdf[k]}."""
    out = np.empty(n, dtype=np.uint32)
    term = math.exp(-mean)
    acc = 0.0
    for k in range(n):
        acc += term
        out[k] = min(int(acc * 4294967296.0), 4294967295)
        term *= mean / (k + 1)
    return out


def synth_csr(nq, D, seed=0, cluster=8, mean=16.0, p_replace=0.15, q0=0, nq_local=None):
    if nq_local is None:
        nq_local = nq - q0
    cdf = poisson_cdf_u32(mean)
    thr = int(p_replace * (1 << 24))
    sizes = np.empty(nq_local, dtype=np.int32)
    args = (ctypes.c_uint64(seed), ctypes.c_int64(q0), ctypes.c_int64(nq_local), ctypes.c_int64(nq),
            ctypes.c_int32(cluster), ctypes.c_uint32(D), _p(cdf, c_u32p), ctypes.c_int32(len(cdf)),
            ctypes.c_uint32(thr))
    lib().qro_synth_sizes(*args, _p(sizes, c_i32p))
    offsets = np.zeros(nq_local + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    rows = np.empty(int(offsets[-1]), dtype=np.int32)
    lib().qro_synth_fill(*args, _p(offsets, c_i64p), _p(rows, c_i32p))
    return offsets, rows


# ---------------------------------------------------------------------------
# naive restatement with the reference's own data structures (tiny cases only)
# ---------------------------------------------------------------------------
def naive_minhash(offsets, rows, perm):
    """recommender.py:116-139 literally: walk rows in permuted order, first hit wins."""
    P, D = perm.shape
    nq = len(offsets) - 1
    inv = {d: [] for d in range(D)}
    for q in range(nq):
        for d in rows[offsets[q]:offsets[q + 1]]:
            inv[int(d)].append(q)
    sign = np.full((P, nq), -1, dtype=np.int64)
    for i in range(P):
        p = perm[i]
        seen = set()
        for ind in np.argsort(p):
            for q in inv[int(ind)]:
                if q not in seen:
                    seen.add(q)
                    sign[i][q] = p[ind]
    return sign.T


def naive_candidates(sig, b):
    """lsh.py:17-55 literally: int16 cast, comma-joined string keys, dict buckets,
    combinations, set."""
    sig = np.asarray(sig)
    l = sig.shape[1]
    assert l % b == 0
    r = l // b
    buckets = [dict() for _ in range(b)]
    for counter, s in enumerate(sig):
        sub = np.stack([s[i:i + r] for i in range(0, l, r)]).astype("int16").astype(str)
        for i, sv in enumerate(sub):
            buckets[i].setdefault(",".join(sv), []).append(counter)
    cands = set()
    for band in buckets:
        for key, hits in band.items():
            if len(hits) > 1 and set(key.split(",")) != {"-1"}:
                for c in combinations(hits, 2):
                    cands.add(c)
    return cands


def pairs_to_u64(pairs_2col):
    a = np.asarray(pairs_2col, dtype=np.uint64).reshape(-1, 2)
    return (a[:, 0] << np.uint64(32)) | a[:, 1]


def u64_to_pairs(p):
    p = np.asarray(p, dtype=np.uint64)
    return np.stack([(p >> np.uint64(32)).astype(np.int64), (p & np.uint64(0xFFFFFFFF)).astype(np.int64)], axis=1)


# ---------------------------------------------------------------------------
# N4 / N1: user similarity and the hybrid prediction loop (recommender.py:216-343)
# ---------------------------------------------------------------------------
QUERY_WEIGHT, USER_WEIGHT, DEFAULT_MEAN = 0.6, 0.4, 60   # recommender.py:32-34


def user_cluster_labels(ratings):
    """recommender.py:226-261: StandardScaler -> PCA(min(r, c, 200)) -> BIRCH with round(nu ** (1/1.3))
    clusters; clusters of one user are pooled under the label n_clusters.  -> int labels [nu]"""
    from sklearn.cluster import Birch
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    ratings = np.asarray(ratings)
    nu = ratings.shape[0]
    n_clusters = round(nu ** (1 / 1.3))
    x = StandardScaler().fit_transform(ratings)
    x = PCA(n_components=min(x.shape[0], x.shape[1], 200)).fit(x).transform(x)
    labels = Birch(n_clusters=n_clusters).fit(x).predict(x)
    counts = np.bincount(labels)
    for c in range(counts.size):
        if counts[c] == 1:
            labels[labels == c] = n_clusters
    return labels


def user_similarities_from_labels(ratings, labels):
    """recommender.py:263-288 given the cluster labels: per cluster the rows are centred on
    their non-zero mean IN PLACE IN THE INTEGER ARRAY (the reference's np.array(self.ratings[...])
    keeps the integer dtype, so the centred values are truncated toward zero), cosine, round to 3,
    zero diagonal and negatives; per user the top K = round(log_1.5 nu) of its cluster row.
    -> {u: {'indexes': int64[], 'values': float64[]}}"""
    from sklearn.metrics.pairwise import cosine_similarity
    ratings = np.asarray(ratings)
    labels = np.asarray(labels)
    nu = ratings.shape[0]
    K = round(math.log(nu, 1.5))
    out = {}
    for c in np.unique(labels):
        members = np.where(labels == c)[0]
        block = np.array(ratings[members])            # integer dtype survives: truncation below is the reference's
        for s in range(len(block)):
            nz = block[s] != 0
            block[s][nz] = block[s][nz] - np.mean(block[s][nz])
        sim = np.around(cosine_similarity(block), 3)
        np.fill_diagonal(sim, 0)
        sim[sim < 0] = 0
        for local, u in enumerate(members):
            order = np.argsort(sim[local])[::-1][:K]
            out[int(u)] = {"indexes": members[order].astype(np.int64), "values": sim[local][order]}
    return out


def user_similarities(ratings):
    """recommender.py:216-290 restated: cluster labels (sklearn, as the reference), then the centred cosine
    inside every cluster and the per-user cut."""
    return user_similarities_from_labels(ratings, user_cluster_labels(ratings))


def np_sum_order(a):
    """numpy's summation order for a contiguous float64 vector of <= 128 elements (pairwise_sum:
    plain loop below 8, else 8 running sums combined as ((0+1)+(2+3))+((4+5)+(6+7)), tail added
    one by one) -- what np.sum does inside weighted_average when numba's jit is the identity."""
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r += float(v)
        return r
    r = [float(a[k]) for k in range(8)]
    i = 8
    while i < n - (n % 8):
        for k in range(8):
            r[k] += float(a[i + k])
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += float(a[i])
        i += 1
    return res


def weighted_average(ratings_row, indexes, values, summation=np_sum_order):
    """recommender.py:36-47"""
    if len(indexes) == 0:
        return 0.0
    ur = np.asarray(ratings_row)[np.asarray(indexes)]
    vals = np.asarray(values, dtype=np.float64)
    wsum = summation(vals[ur != 0])
    if wsum == 0:
        return 0.0
    return summation(ur * vals) / wsum


def predict_scores(ratings, query_sims, user_sims, summation=np_sum_order):
    """The hybrid loop of compute_scores, recommender.py:301-331: every zero cell (i, j) gets
    round(...) of the blend of the query-side and user-side weighted averages (Python round =
    half to even).  -> int64 matrix like `finalPredictions`."""
    ratings = np.asarray(ratings)
    final = ratings.astype(np.int64).copy()
    for i, j in np.array(np.where(ratings == 0)).T:
        qp = 0.0
        if int(j) in query_sims:
            qp = weighted_average(ratings[i], query_sims[int(j)]["indexes"], query_sims[int(j)]["values"], summation)
        up = weighted_average(ratings.T[j], user_sims[int(i)]["indexes"], user_sims[int(i)]["values"], summation)
        if up == 0 and qp == 0:
            final[i][j] = 0
        elif up == 0:
            final[i][j] = round(qp * (QUERY_WEIGHT + (USER_WEIGHT * 0.5)) + DEFAULT_MEAN * (USER_WEIGHT * 0.5))
        elif qp == 0:
            final[i][j] = round(up * (USER_WEIGHT + (QUERY_WEIGHT * 0.5)) + DEFAULT_MEAN * (QUERY_WEIGHT * 0.5))
        else:
            final[i][j] = round(qp * QUERY_WEIGHT + up * USER_WEIGHT)
    return final
