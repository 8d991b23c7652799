#!/usr/bin/env python3
"""Capture golden input/output vectors from the reference's own Python code.

Runs ONLY in the build container (needs /root/reference; the GPU box never has it).
Writes small .npz / .csv fixtures under tests/golden/.  The fixtures are data
(inputs + the reference's outputs on them); no reference source is copied.

How the reference is run:
  * lsh.py imports as-is.
  * recommender.py / resources/generator.py import `datatable` and `numba`, which
    are not installed here and cannot be.  The hot-path functions never call
    either library (datatable: only Recommender.init / parse_queries; numba: only
    the @jit decorator on weighted_average), so two inert placeholder modules are
    put in sys.modules to satisfy the import statements.  Recommender.init() is
    bypassed by assigning the attributes it would have set.
  * generator.py is run with cwd = a scratch copy of resources/ (it reads
    ./input/* and writes ./output/*), with a pandas-backed `dt.fread`.
  * sys.dont_write_bytecode keeps __pycache__ out of the read-only reference tree.

Usage:  python tools/make_golden.py            (rewrites the round-1 base fixtures)
        python tools/make_golden.py NAME...    (only the named later additions: full_p100_r5,
                                                full_p50_r5_wrap, pieces_p96_b12, cfg1_scores, cfg1b, cfg2)
"""
import os
import sys
import io
import types
import shutil
import random
import tempfile
import contextlib

sys.dont_write_bytecode = True

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# --------------------------------------------------------------------------
# inert placeholders for the two absent third-party imports
# --------------------------------------------------------------------------
class _Frame:
    """pandas-backed stand-in for the few datatable.Frame calls generator.py makes."""

    def __init__(self, obj):
        self.df = obj if isinstance(obj, pd.DataFrame) else pd.DataFrame(obj)

    def __setitem__(self, key, value):  # dataset[:] = dt.str64
        self.df = self.df.astype(str)

    def to_pandas(self):
        return self.df

    def to_numpy(self):
        return self.df.to_numpy()

    @property
    def shape(self):
        return self.df.shape

    @property
    def names(self):
        return tuple(self.df.columns)


def _install_placeholders():
    dtmod = types.ModuleType("datatable")

    class _dt:
        str64 = "str64"
        Frame = _Frame

        @staticmethod
        def fread(path, header=None, columns=None):
            return _Frame(pd.read_csv(path))

    dtmod.dt = _dt
    dtmod.f = None
    dtmod.ifelse = None
    dtmod.update = None
    sys.modules["datatable"] = dtmod

    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda fn: fn)
    sys.modules["numba"] = nb


@contextlib.contextmanager
def _quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


# --------------------------------------------------------------------------
# helpers around the reference objects
# --------------------------------------------------------------------------
def _fresh_lsh_state(lsh_mod):
    # LSH.buckets is a class attribute that leaks across instances (lsh.py:9)
    lsh_mod.LSH.buckets = []
    lsh_mod.LSH.counter = 0


def _csr_from_shingles(shingles, nq):
    """row -> [queries]  ==>  CSR query -> sorted rows"""
    per_q = [[] for _ in range(nq)]
    for d in sorted(shingles.keys()):
        for q in shingles[d]:
            per_q[q].append(d)
    offsets = np.zeros(nq + 1, dtype=np.int64)
    for q in range(nq):
        offsets[q + 1] = offsets[q] + len(per_q[q])
    rows = np.array([d for lst in per_q for d in lst], dtype=np.int32)
    return offsets, rows


def _shingles_from_csr(offsets, rows, D):
    sh = {d: [] for d in range(D)}
    nq = len(offsets) - 1
    for q in range(nq):
        for d in rows[offsets[q]:offsets[q + 1]]:
            sh[int(d)].append(q)
    return sh


def _make_rec(recommender_mod, D, nq):
    rec = recommender_mod.Recommender()
    rec.dataset = pd.DataFrame({"c": np.zeros(D, dtype=np.int8)})
    rec.queriesIDs = np.array(["Q%d" % (i + 1) for i in range(nq)])
    rec.queries = np.zeros((nq, 5), dtype=object)  # only .size is read (recommender.py:131)
    rec.datasetFeatures = []
    rec.tupleCount = {}
    return rec


def _sim_dict_to_arrays(qs):
    qids = np.array(sorted(qs.keys()), dtype=np.int32)
    off = np.zeros(len(qids) + 1, dtype=np.int64)
    idx, val = [], []
    for n, q in enumerate(qids):
        i = np.asarray(qs[int(q)]["indexes"], dtype=np.int64)
        v = np.asarray(qs[int(q)]["values"], dtype=np.float64)
        idx.append(i)
        val.append(v)
        off[n + 1] = off[n] + len(i)
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    val = np.concatenate(val) if val else np.zeros(0, np.float64)
    return qids, off, idx.astype(np.int32), val


def _pairs_array(cands):
    a = np.array(sorted(cands), dtype=np.int32).reshape(-1, 2)
    return a


def _all_pair_cos(sig, pairs):
    """sklearn cosine for every candidate pair in the call form of recommender.py:203-204
    ([sig_i, sig_j] -> row 0), rounded as there.  Third-party library, not reference code:
    pins scores of pairs that the reference's top-K cut drops."""
    from sklearn.metrics.pairwise import cosine_similarity
    out = np.zeros(len(pairs), dtype=np.float64)
    for n, (i, j) in enumerate(pairs):
        out[n] = np.around(cosine_similarity([sig[i], sig[j]])[0][1], 3)
    return out


def run_lsh_only(lsh_mod, sig, b):
    _fresh_lsh_state(lsh_mod)
    l = lsh_mod.LSH(b)
    for s in sig:
        l.compute_buckets(s)
    c = l.get_candidates(sig)
    _fresh_lsh_state(lsh_mod)
    return c


def run_signatures(recommender_mod, offsets, rows, D, P, seed):
    nq = len(offsets) - 1
    rec = _make_rec(recommender_mod, D, nq)
    sh = _shingles_from_csr(offsets, rows, D)
    rec.compute_shingles = lambda: sh
    recommender_mod.PERM = P
    np.random.seed(seed)
    with _quiet():
        sig = rec.compute_signatures()
    return np.ascontiguousarray(sig)


def run_query_sims(recommender_mod, lsh_mod, offsets, rows, D, P, seed):
    nq = len(offsets) - 1
    rec = _make_rec(recommender_mod, D, nq)
    sh = _shingles_from_csr(offsets, rows, D)
    rec.compute_shingles = lambda: sh
    recommender_mod.PERM = P
    _fresh_lsh_state(lsh_mod)
    np.random.seed(seed)
    with _quiet() as buf:
        qs = rec.compute_querySimilarities()
    _fresh_lsh_state(lsh_mod)
    log = buf.getvalue()
    # "Max query candidates: K, Max bands: b, ..." (recommender.py:165)
    line = [l for l in log.splitlines() if l.startswith("Max query candidates")][0]
    K = int(line.split(":")[1].split(",")[0])
    b = int(line.split("Max bands:")[1].split(",")[0])
    return qs, K, b


# --------------------------------------------------------------------------
# synthetic answer sets for the fixtures (inputs are stored in the fixture,
# so this recipe does not need to be reproducible elsewhere)
# --------------------------------------------------------------------------
def synth_csr(nq, D, seed, cluster=8, mean=16, p_replace=0.15, n_empty=0, n_dup=0):
    rng = np.random.default_rng(seed)
    nb = max(1, nq // cluster)
    bases = []
    for _ in range(nb):
        s = max(1, int(rng.poisson(mean)))
        bases.append(rng.choice(D, size=min(s, D), replace=False))
    sets = []
    for q in range(nq):
        base = bases[q % nb].copy()
        m = rng.random(len(base)) < p_replace
        base[m] = rng.integers(0, D, size=int(m.sum()))
        sets.append(np.unique(base))
    for k in range(n_empty):
        sets[(7 * k + 3) % nq] = np.zeros(0, dtype=np.int64)
    for k in range(n_dup):
        sets[(11 * k + 5) % nq] = sets[1].copy()
    offsets = np.zeros(nq + 1, dtype=np.int64)
    for q in range(nq):
        offsets[q + 1] = offsets[q] + len(sets[q])
    rows = np.concatenate(sets).astype(np.int32) if nq else np.zeros(0, np.int32)
    return offsets, rows


# --------------------------------------------------------------------------
def fixture_generator_default(recommender_mod, lsh_mod, sub="cfg1", gen_seed=20250114, name="cfg1_hotpath", sizes=None):
    """config 1: resources/generator.py defaults -> 4 CSVs -> reference hot path.
    sizes = (MAX_DATA, MAX_QUERIES, MAX_USERS) overrides the generator's module constants (generator.py:14-16)."""
    scratch = tempfile.mkdtemp(prefix="qr_gen_")
    try:
        shutil.copytree(os.path.join(REF, "resources", "input"), os.path.join(scratch, "input"))
        cwd = os.getcwd()
        os.chdir(scratch)
        sys.path.insert(0, os.path.join(REF, "resources"))
        import generator  # noqa
        sys.path.pop(0)
        for attr, empty in (("user_tastes", {}), ("queries", []), ("user_queries", {}), ("usersIDs", []),
                            ("queriesIDs", [])):
            setattr(generator, attr, type(empty)())          # module-level state: start clean on a second run
        if sizes is not None:
            generator.MAX_DATA, generator.MAX_QUERIES, generator.MAX_USERS = sizes
        random.seed(gen_seed)
        np.random.seed(gen_seed)
        with _quiet():
            generator.get_data()
            generator.create_dataset()
            generator.create_users()
            generator.create_queries()
            generator.create_matrix()
        os.chdir(cwd)
        gdir = os.path.join(OUT, sub)
        os.makedirs(gdir, exist_ok=True)
        for fname in ("dataset", "users", "queries", "utility_matrix"):
            shutil.copy(os.path.join(scratch, "output", fname + ".csv"), os.path.join(gdir, fname + ".csv"))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)

    # feed the CSVs to the reference Recommender the way main.py would, minus datatable
    dataset = pd.read_csv(os.path.join(gdir, "dataset.csv"), dtype=str)
    feats = list(dataset.columns)[1:]
    qrows, qids = [], []
    with open(os.path.join(gdir, "queries.csv")) as fh:
        for line in fh:
            vals = line.rstrip("\n").split(",")
            qids.append(vals[0])
            el = ["" for _ in feats]
            for v in vals[1:]:
                a = v.split("=")
                el[feats.index(a[0])] = a[1]
            qrows.append(el)
    rec = recommender_mod.Recommender()
    rec.datasetFeatures = feats
    rec.dataset = dataset
    rec.queries = np.array(qrows, dtype=object)
    rec.queriesIDs = np.array(qids)
    rec.tupleCount = {}
    D = dataset.shape[0]
    nq = len(qids)
    P = 180
    seed = 42
    recommender_mod.PERM = P
    with _quiet():
        sh = rec.compute_shingles()
    offsets, rows = _csr_from_shingles(sh, nq)
    np.random.seed(seed)
    with _quiet():
        sig = np.ascontiguousarray(rec.compute_signatures())
    _fresh_lsh_state(lsh_mod)
    np.random.seed(seed)
    with _quiet() as buf:
        qs = rec.compute_querySimilarities()
    _fresh_lsh_state(lsh_mod)
    line = [l for l in buf.getvalue().splitlines() if l.startswith("Max query candidates")][0]
    K = int(line.split(":")[1].split(",")[0])
    b = int(line.split("Max bands:")[1].split(",")[0])
    cands = run_lsh_only(lsh_mod, sig, b)
    pairs = _pairs_array(cands)
    qids_a, off, idx, val = _sim_dict_to_arrays(qs)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        D=D, P=P, b=b, K=K, seed=seed, offsets=offsets, rows=rows,
        sig=sig.astype(np.int32), pairs=pairs, pair_cos=_all_pair_cos(sig, pairs),
        qs_q=qids_a, qs_off=off, qs_idx=idx, qs_val=val)
    print("%s: D=%d nq=%d P=%d b=%d K=%d pairs=%d sims=%d" % (name, D, nq, P, b, K, len(pairs), len(qids_a)))


def fixture_cfg1_scores(recommender_mod, lsh_mod, sub="cfg1", name="cfg1_scores"):
    """config 1 through the reference's whole compute_scores (query + user similarity + hybrid
    prediction loop, recommender.py:216-343) on the committed generator-default CSVs."""
    gdir = os.path.join(OUT, sub)
    dataset = pd.read_csv(os.path.join(gdir, "dataset.csv"), dtype=str)
    feats = list(dataset.columns)[1:]
    qrows, qids = [], []
    with open(os.path.join(gdir, "queries.csv")) as fh:
        for line in fh:
            vals = line.rstrip("\n").split(",")
            qids.append(vals[0])
            el = ["" for _ in feats]
            for v in vals[1:]:
                a = v.split("=")
                el[feats.index(a[0])] = a[1]
            qrows.append(el)
    users = pd.read_csv(os.path.join(gdir, "users.csv"), header=None)
    um = pd.read_csv(os.path.join(gdir, "utility_matrix.csv"))
    ratings = um.fillna(0).to_numpy().astype(np.int64)      # what Recommender.init leaves (NaN -> 0, :61-64)
    rec = recommender_mod.Recommender()
    rec.datasetFeatures = feats
    rec.dataset = dataset
    rec.queries = np.array(qrows, dtype=object)
    rec.queriesIDs = np.array(qids)
    rec.usersIDs = users.to_numpy().T[0]
    rec.ratings = ratings
    rec.tupleCount = {}
    recommender_mod.PERM = 180
    seed = 42
    _fresh_lsh_state(lsh_mod)
    np.random.seed(seed)
    with _quiet():
        us = rec.compute_userSimilarities()
    _fresh_lsh_state(lsh_mod)
    np.random.seed(seed)
    with _quiet():
        to_predict, final, missed = rec.compute_scores()
    _fresh_lsh_state(lsh_mod)
    nu = len(rec.usersIDs)
    K = max(len(us[u]["indexes"]) for u in us)
    us_idx = np.full((nu, K), -1, dtype=np.int64)
    us_val = np.zeros((nu, K), dtype=np.float64)
    for u in range(nu):
        n = len(us[u]["indexes"])
        us_idx[u, :n] = us[u]["indexes"]
        us_val[u, :n] = us[u]["values"]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, P=180, ratings=ratings,
                        us_idx=us_idx, us_val=us_val, final=final.to_numpy().astype(np.int64),
                        to_predict=np.asarray(to_predict, dtype=np.int64), missed=np.asarray(missed, dtype=np.int64))
    print("%s: users=%d queries=%d to_predict=%d missed=%d" % (name, nu, len(qids), len(to_predict), len(missed)))


def fixture_full(recommender_mod, lsh_mod, name, nq, D, P, seed, data_seed, **kw):
    offsets, rows = synth_csr(nq, D, data_seed, **kw)
    sig = run_signatures(recommender_mod, offsets, rows, D, P, seed)
    qs, K, b = run_query_sims(recommender_mod, lsh_mod, offsets, rows, D, P, seed)
    cands = run_lsh_only(lsh_mod, sig, b)
    pairs = _pairs_array(cands)
    qids_a, off, idx, val = _sim_dict_to_arrays(qs)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        D=D, P=P, b=b, K=K, seed=seed, offsets=offsets, rows=rows,
        sig=sig.astype(np.int32), pairs=pairs, pair_cos=_all_pair_cos(sig, pairs),
        qs_q=qids_a, qs_off=off, qs_idx=idx, qs_val=val)
    print("%s: D=%d nq=%d P=%d b=%d K=%d pairs=%d sims=%d" % (name, D, nq, P, b, K, len(pairs), len(qids_a)))


def fixture_pieces(recommender_mod, lsh_mod, name, nq, D, P, b, seed, data_seed, **kw):
    """BASELINE shapes the reference's band rule cannot select (128/32, 256/64):
    compute_signatures + LSH(b) pieces only."""
    offsets, rows = synth_csr(nq, D, data_seed, **kw)
    sig = run_signatures(recommender_mod, offsets, rows, D, P, seed)
    cands = run_lsh_only(lsh_mod, sig, b)
    pairs = _pairs_array(cands)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        D=D, P=P, b=b, seed=seed, offsets=offsets, rows=rows,
        sig=sig.astype(np.int32), pairs=pairs, pair_cos=_all_pair_cos(sig, pairs))
    print("%s: D=%d nq=%d P=%d b=%d pairs=%d" % (name, D, nq, P, b, len(pairs)))


def fixture_lsh_edge(lsh_mod):
    """lsh.py edge semantics on hand-made signatures: int16 wrap (65537 -> 1, 65535 -> -1),
    all -1 bands are 'empty', partially -1 bands are not, duplicates, P % b != 0 assertion,
    make_subvecs output."""
    rng = np.random.default_rng(5)
    P, b = 24, 6
    base = rng.integers(0, 100000, size=(40, P)).astype(np.int64)
    sig = base.copy()
    sig[1] = sig[0]                      # exact duplicate -> every band collides
    sig[2] = sig[0] + 65536              # equal after int16 wrap in every band
    sig[3, :4] = sig[0, :4]              # one shared band
    sig[4] = -1                          # empty
    sig[5] = -1                          # empty: must NOT pair with 4
    sig[6] = 65535                       # wraps to -1 in every band -> treated as empty
    sig[7, :4] = [-1, 65535, -1, -1]     # band 0 all "-1" after wrap -> empty band
    sig[8, :4] = [-1, 65535, -1, -1]
    sig[9, 4:8] = [-1, 7, -1, 7]         # partially -1 band
    sig[10, 4:8] = [65535, 7, -1, 65543]  # same after wrap -> candidate with 9
    sig[11:19] = sig[20]                 # bucket of 9 identical -> 36 pairs
    cands = run_lsh_only(lsh_mod, sig, b)
    _fresh_lsh_state(lsh_mod)
    l = lsh_mod.LSH(b)
    sub = l.make_subvecs(sig[2])
    _fresh_lsh_state(lsh_mod)
    assert_raised = False
    try:
        lsh_mod.LSH(5).make_subvecs(sig[0])
    except AssertionError:
        assert_raised = True
    _fresh_lsh_state(lsh_mod)
    np.savez_compressed(os.path.join(OUT, "lsh_edge.npz"), P=P, b=b, sig=sig,
                        pairs=_pairs_array(cands), subvecs_row2=sub,
                        assert_raised=assert_raised)
    print("lsh_edge: pairs=%d assert_raised=%s" % (len(cands), assert_raised))


def main():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py needs the reference tree at " + REF)
    os.makedirs(OUT, exist_ok=True)
    _install_placeholders()
    sys.path.insert(0, REF)
    import lsh as lsh_mod
    import recommender as recommender_mod
    sys.path.pop(0)

    only = set(sys.argv[1:])

    def want(name):
        return not only or name in only

    if only:
        # wide bands: PERM=100 makes the reference's rule pick b=20, r=5 (recommender.py:156-163)
        if want("full_p100_r5"):
            fixture_full(recommender_mod, lsh_mod, "full_p100_r5", nq=700, D=2500, P=100, seed=19, data_seed=7,
                         n_empty=4, n_dup=5)
        if want("full_p50_r5_wrap"):
            fixture_full(recommender_mod, lsh_mod, "full_p50_r5_wrap", nq=400, D=90000, P=50, seed=23, data_seed=9,
                         n_empty=2, n_dup=3, cluster=6, mean=8)
        if want("cfg1_scores"):
            fixture_cfg1_scores(recommender_mod, lsh_mod)
        if want("cfg1b"):     # a second generator-default data set (different generator seed)
            fixture_generator_default(recommender_mod, lsh_mod, sub="cfg1b", gen_seed=7, name="cfg1b_hotpath")
            fixture_cfg1_scores(recommender_mod, lsh_mod, sub="cfg1b", name="cfg1b_scores")
        if want("cfg2"):      # other sizes than the generator's defaults: 700 table rows, 150 queries, 60 users
            fixture_generator_default(recommender_mod, lsh_mod, sub="cfg2", gen_seed=11, name="cfg2_hotpath",
                                      sizes=(700, 150, 60))
            fixture_cfg1_scores(recommender_mod, lsh_mod, sub="cfg2", name="cfg2_scores")
        if want("pieces_p96_b12"):
            fixture_pieces(recommender_mod, lsh_mod, "pieces_p96_b12", nq=500, D=3000, P=96, b=12, seed=29,
                           data_seed=10, n_empty=3, n_dup=3, p_replace=0.05)
        return
    fixture_generator_default(recommender_mod, lsh_mod)
    # shapes the reference's own band rule selects (recommender.py:151-165)
    fixture_full(recommender_mod, lsh_mod, "full_p180", nq=1200, D=4096, P=180, seed=42, data_seed=1)
    fixture_full(recommender_mod, lsh_mod, "full_p160", nq=800, D=3000, P=160, seed=7, data_seed=2,
                 n_empty=5, n_dup=6)
    fixture_full(recommender_mod, lsh_mod, "full_p320", nq=400, D=2048, P=320, seed=9, data_seed=3)
    fixture_full(recommender_mod, lsh_mod, "full_p200", nq=400, D=1500, P=200, seed=11, data_seed=4)
    # int16 wrap inside the full path: D > 65536
    fixture_full(recommender_mod, lsh_mod, "full_p160_wrap", nq=500, D=100000, P=160, seed=13, data_seed=5,
                 n_empty=3, n_dup=4)
    # heavy top-K ties: tiny universe, big clusters
    fixture_full(recommender_mod, lsh_mod, "full_p160_ties", nq=300, D=64, P=160, seed=17, data_seed=6,
                 cluster=50, mean=4, p_replace=0.05)
    # BASELINE shapes (pieces only)
    fixture_pieces(recommender_mod, lsh_mod, "pieces_p128_b32", nq=1500, D=32768, P=128, b=32, seed=42,
                   data_seed=0, n_empty=4, n_dup=5)
    fixture_pieces(recommender_mod, lsh_mod, "pieces_p256_b64", nq=600, D=32768, P=256, b=64, seed=42,
                   data_seed=0)
    fixture_pieces(recommender_mod, lsh_mod, "pieces_p128_b32_wrap", nq=600, D=100000, P=128, b=32, seed=3,
                   data_seed=8, n_empty=2, n_dup=3)
    fixture_lsh_edge(lsh_mod)


if __name__ == "__main__":
    main()
