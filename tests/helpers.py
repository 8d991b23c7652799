"""Shared test helpers: golden loading and the tie-aware top-K comparison."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FULL = ["cfg1_hotpath", "cfg1b_hotpath", "cfg2_hotpath", "full_p180", "full_p160", "full_p320", "full_p200", "full_p160_wrap", "full_p160_ties",
        "full_p100_r5", "full_p50_r5_wrap"]   # the last two: wide bands (r = 5), the reference's pick for PERM = 100 / 50
PIECES = ["pieces_p128_b32", "pieces_p256_b64", "pieces_p128_b32_wrap", "pieces_p96_b12"]   # last: r = 8


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def pairs_u64(p2):
    a = np.asarray(p2, dtype=np.uint64).reshape(-1, 2)
    return (a[:, 0] << np.uint64(32)) | a[:, 1]


def check_topk_tie_aware(g, pairs, milli, src, dst, val, K):
    """g: golden dict with qs_q/qs_off/qs_idx/qs_val captured from the reference's
    compute_querySimilarities.  (pairs, milli): every candidate pair with its rounded
    score*1000.  (src, dst, val): the build's top-K COO output.

    The reference's order among equal rounded values is arbitrary (np.argsort + set
    iteration), so equality is: same queries, same multiset of values per query, and
    every reference neighbour is a real candidate of that query with exactly that value.
    """
    score = {int(p): int(m) for p, m in zip(pairs, milli)}
    mine = {}
    for s, d, v in zip(src, dst, val):
        mine.setdefault(int(s), []).append((int(d), int(v)))
    ref_q = [int(q) for q in g["qs_q"]]
    assert sorted(mine.keys()) == sorted(ref_q), "set of queries with neighbours differs"
    deg = {}
    for p in pairs:
        i, j = int(p >> np.uint64(32)), int(p & np.uint64(0xFFFFFFFF))
        deg[i] = deg.get(i, 0) + 1
        deg[j] = deg.get(j, 0) + 1
    for n, q in enumerate(ref_q):
        lo, hi = int(g["qs_off"][n]), int(g["qs_off"][n + 1])
        ridx = g["qs_idx"][lo:hi]
        rval = g["qs_val"][lo:hi]
        assert len(ridx) == min(deg[q], K)
        m = mine[q]
        assert len(m) == len(ridx)
        mv = np.array([v for _, v in m], dtype=np.float64) / 1000.0
        # reference values are descending; so are ours
        assert np.all(np.diff(rval) <= 0)
        assert np.array_equal(mv, rval), "value multiset differs for query %d" % q
        # our order: value desc, id asc
        for (d0, v0), (d1, v1) in zip(m[:-1], m[1:]):
            assert v0 > v1 or (v0 == v1 and d0 < d1)
        for j, v in zip(ridx, rval):
            key = (min(q, int(j)) << 32) | max(q, int(j))
            assert key in score, "reference neighbour is not a candidate"
            assert score[key] / 1000.0 == v
        # strictly-above-cutoff neighbours must match exactly
        cutoff = rval[-1]
        ref_top = {int(j) for j, v in zip(ridx, rval) if v > cutoff}
        my_top = {d for d, v in m if v / 1000.0 > cutoff}
        assert ref_top == my_top


def recall_at_k(ref_src, ref_dst, ref_val, src, dst, val, k=10):
    """Tie-aware recall@k of (src,dst,val) against the reference lists: a returned
    neighbour counts as a hit if it is in the reference top-k or ties its k-th value."""
    ref = {}
    for s, d, v in zip(ref_src, ref_dst, ref_val):
        ref.setdefault(int(s), []).append((int(d), int(v)))
    got = {}
    for s, d, v in zip(src, dst, val):
        got.setdefault(int(s), []).append((int(d), int(v)))
    hit = tot = 0
    for q, lst in ref.items():
        top = lst[:k]
        cutoff = top[-1][1]
        ids = {d for d, _ in top}
        mine = got.get(q, [])[:k]
        tot += len(top)
        for d, v in mine:
            if d in ids or v == cutoff:
                hit += 1
    return hit / max(tot, 1)


GENERATOR_SETS = ["cfg1", "cfg1b", "cfg2"]   # data sets written by the reference's resources/generator.py: its default sizes
                                             # with two seeds, and 700 rows x 150 queries x 60 users


def generator_table_and_queries(sub):
    """the table columns (as the reference compares them: strings) and the (nq, nfeat) query matrix
    of a generator-default data set under tests/golden/<sub>/"""
    import pandas as pd
    gdir = os.path.join(GOLDEN, sub)
    dataset = pd.read_csv(os.path.join(gdir, "dataset.csv"), dtype=str)
    feats = list(dataset.columns)[1:]
    qrows = []
    with open(os.path.join(gdir, "queries.csv")) as fh:
        for line in fh:
            vals = line.rstrip("\n").split(",")
            el = ["" for _ in feats]
            for v in vals[1:]:
                a = v.split("=")
                el[feats.index(a[0])] = a[1]
            qrows.append(el)
    return [dataset[f].to_numpy() for f in feats], np.array(qrows, dtype=object)


# ---------------------------------------------------------------------------- N3: what main.py hands to init()
class FrameStandIn:
    """A pandas-backed stand-in for the `datatable.Frame` objects main.py:23-85 builds (datatable is not
    installable here): exactly the read-only surface a Frame offers to a consumer -- .names, .shape,
    .to_numpy(), .to_pandas() -- and nothing of pandas' own (no .columns, .astype, .drop), so a drop-in init()
    that only duck-types pandas fails on it the way it would on a real Frame."""

    def __init__(self, df):
        self._df = df

    @property
    def names(self):
        return tuple(str(c) for c in self._df.columns)

    @property
    def shape(self):
        return self._df.shape

    def to_numpy(self):
        return self._df.to_numpy()

    def to_pandas(self):
        return self._df.copy()


def fread_like(path, header=True, columns=None):
    """dt.fread(path[, header=False][, columns=[...]]) as main.py uses it, on pandas: the utility matrix's
    header names only the query columns while every row leads with the user id, so fread sees one column more
    than names and main.py:70-77 passes the full list (["user"] + queriesIDs)."""
    import pandas as pd
    if not header:
        return FrameStandIn(pd.read_csv(path, header=None))
    if columns is not None:
        return FrameStandIn(pd.read_csv(path, header=None, skiprows=1, names=list(columns)))
    return FrameStandIn(pd.read_csv(path))


def main_py_inputs(rec, gdir):
    """the objects main.py builds before recommender.init (same order, same calls on `rec`):
    -> (users, queries, queriesIDs, dataset, ratings)"""
    dataset = fread_like(os.path.join(gdir, "dataset.csv"))                       # main.py:23
    rec.datasetFeatures = list(dataset.names)[1::]                                # main.py:33
    users = fread_like(os.path.join(gdir, "users.csv"), header=False)             # main.py:42
    queries, qids = rec.parse_queries(os.path.join(gdir, "queries.csv"))          # main.py:59
    cols = ["user"] + qids                                                        # main.py:70
    ratings = fread_like(os.path.join(gdir, "utility_matrix.csv"), columns=cols)  # main.py:77
    return users, queries, qids, dataset, ratings
