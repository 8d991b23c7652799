"""CPU tests: the oracle (oracle/) against golden vectors captured from the reference
(tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import (FULL, PIECES, GENERATOR_SETS, load, pairs_u64, check_topk_tie_aware,
                     generator_table_and_queries)


@pytest.mark.parametrize("name", FULL + PIECES)
def test_minhash_matches_reference(name):
    g = load(name)
    perm = O.legacy_permutations(int(g["seed"]), int(g["P"]), int(g["D"]))
    sig = O.minhash(g["offsets"], g["rows"], perm)
    assert sig.dtype == np.int32
    assert np.array_equal(sig, g["sig"])


@pytest.mark.parametrize("name", ["cfg1_hotpath", "cfg1b_hotpath", "full_p160_ties"])
def test_naive_minhash_matches_reference(name):
    g = load(name)
    perm = O.legacy_permutations(int(g["seed"]), int(g["P"]), int(g["D"]))
    sig = O.naive_minhash(g["offsets"], g["rows"], perm)
    assert np.array_equal(sig, g["sig"])


@pytest.mark.parametrize("name", FULL + PIECES)
def test_candidates_match_reference(name):
    g = load(name)
    P, b = int(g["P"]), int(g["b"])
    assert np.array_equal(O.candidates_from_sig(g["sig"], b), np.sort(pairs_u64(g["pairs"])))
    if P // b <= 4:                       # packed 64-bit keys are exact bucket ids
        keys = O.band_keys(g["sig"], b)
        pairs = O.candidates(keys, P // b)
        assert np.array_equal(pairs, np.sort(pairs_u64(g["pairs"])))
        assert O.emitted_pairs(keys, P // b) >= len(pairs)
    else:
        with pytest.raises(ValueError):
            O.band_keys(g["sig"], b)


def test_lsh_edge_semantics():
    g = load("lsh_edge")
    sig = g["sig"]
    b = int(g["b"])
    # int16 wrap happens on the low 16 bits; feed the oracle the wrapped int32 view
    sig32 = (sig & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    keys = O.band_keys(sig32, b)
    pairs = O.candidates(keys, sig.shape[1] // b)
    assert np.array_equal(pairs, np.sort(pairs_u64(g["pairs"])))
    naive = O.naive_candidates(sig, b)
    assert sorted(naive) == [tuple(x) for x in g["pairs"].tolist()]
    assert bool(g["assert_raised"])
    with pytest.raises(AssertionError):
        O.band_keys(sig32, 5)
    # make_subvecs view of row 2 (int16 wrap of +65536)
    r = sig.shape[1] // b
    assert np.array_equal(g["subvecs_row2"], sig[2].reshape(b, r).astype(np.int16))


@pytest.mark.parametrize("name", ["cfg1_hotpath", "pieces_p128_b32_wrap"])
def test_naive_candidates_match_reference(name):
    g = load(name)
    naive = O.naive_candidates(g["sig"], int(g["b"]))
    assert sorted(naive) == [tuple(x) for x in g["pairs"].tolist()]


@pytest.mark.parametrize("name", FULL + PIECES)
def test_scores_match_sklearn_rounding(name):
    g = load(name)
    pairs = np.sort(pairs_u64(g["pairs"]))
    # golden pair_cos is in the order of g["pairs"], which is already sorted
    assert np.array_equal(pairs, pairs_u64(g["pairs"]))
    m_exact = O.score_pairs(g["sig"], pairs, mode=1)
    m_skl = O.score_pairs(g["sig"], pairs, mode=0)
    assert np.array_equal(m_exact, m_skl)
    assert np.array_equal(m_exact / 1000.0, g["pair_cos"])


@pytest.mark.parametrize("name", FULL)
def test_topk_matches_reference_tie_aware(name):
    g = load(name)
    P, b, K = int(g["P"]), int(g["b"]), int(g["K"])
    r = O.query_similarities(g["offsets"], g["rows"], int(g["D"]), P, b, K, int(g["seed"]))
    check_topk_tie_aware(g, r["pairs"], r["milli"], r["src"], r["dst"], r["val"], K)
    d = O.sims_to_dict(r["src"], r["dst"], r["val"])
    assert sorted(d.keys()) == sorted(int(q) for q in g["qs_q"])


def test_band_rule_matches_reference_choices():
    # values observed from the reference run (golden 'b' and 'K')
    for name in FULL:
        g = load(name)
        assert O.select_bands(int(g["P"])) == int(g["b"])
        nq = len(g["offsets"]) - 1
        assert O.max_candidates(nq) == int(g["K"])
    for P in (128, 256):
        with pytest.raises(ValueError):
            O.select_bands(P)


def test_synth_generator_is_a_pure_function_of_index():
    off, rows = O.synth_csr(4000, 32768, seed=0)
    off2, rows2 = O.synth_csr(4000, 32768, seed=0, q0=1000, nq_local=500)
    assert np.array_equal(rows[off[1000]:off[1500]], rows2)
    sizes = np.diff(off)
    assert sizes.min() >= 1 and sizes.max() <= 48
    assert 14.0 < sizes.mean() < 18.0
    for q in (0, 17, 3999):
        s = rows[off[q]:off[q + 1]]
        assert np.all(np.diff(s) > 0) and s.min() >= 0 and s.max() < 32768


@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_answer_sets_match_reference_compute_shingles(sub):
    """N2: the restated compute_shingles against the CSR derived from the reference's own
    compute_shingles() output on the generator-default CSVs (tools/make_golden.py)."""
    g = load(sub + "_hotpath")
    cols, queries = generator_table_and_queries(sub)
    off, rows = O.answer_sets(cols, queries)
    assert np.array_equal(off, g["offsets"]) and np.array_equal(rows, g["rows"])


@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_user_similarity_and_prediction_loop_match_reference(sub):
    """N4 + N1: restated compute_userSimilarities / compute_scores against the reference's own
    outputs on the generator-default CSVs (tests/golden/<sub>_scores.npz)."""
    g, h = load(sub + "_scores"), load(sub + "_hotpath")
    us = O.user_similarities(g["ratings"])
    for u in range(len(g["ratings"])):
        n = int((g["us_idx"][u] >= 0).sum())
        assert np.array_equal(us[u]["values"], g["us_val"][u][:n])            # values exact; tie order is arbitrary
        nz = g["us_val"][u][:n] > 0
        assert sorted(us[u]["indexes"][nz].tolist()) == sorted(g["us_idx"][u][:n][nz].tolist())
    r = O.query_similarities(h["offsets"], h["rows"], int(h["D"]), int(h["P"]), int(h["b"]), int(h["K"]), int(h["seed"]))
    qs = O.sims_to_dict(r["src"], r["dst"], r["val"])
    final = O.predict_scores(g["ratings"], qs, us)
    assert np.array_equal(final, g["final"])
    assert np.array_equal(np.array(np.where(g["ratings"] == 0)).T, g["to_predict"])
    assert np.array_equal(np.array(np.where(final == 0)).T, g["missed"])
    # numba's sequential np.sum (the real reference) and numpy's pairwise order agree here
    seq = O.predict_scores(g["ratings"], qs, us, summation=lambda a: float(sum(float(x) for x in a)))
    assert np.array_equal(seq, g["final"])


@pytest.mark.parametrize("nq,D", [(1_000_000, 32768), (300_000, 100_000)])
def test_sklearn_order_and_exact_integer_scoring_agree_at_scale(nq, D):
    """a5 (recommender.py:203-204) at full config-2 size: the sklearn-order arithmetic (mode 0:
    normalise each row in float64, then dot) and the exact-integer form the HIP kernel computes
    (mode 1: int64 dot / (sqrt(na) * sqrt(nb))) must round to the same milli on EVERY candidate
    pair -- a flip at a x.xxx5 boundary would be a real parity finding for score.hip."""
    P, b = 128, 32
    off, rows = O.synth_csr(nq, D, seed=0)
    sig = O.minhash(off, rows, O.legacy_permutations(42, P, D))
    pairs = O.candidates(O.band_keys(sig, b), P // b)
    assert len(pairs) > 2 * nq
    m0 = O.score_pairs(sig, pairs, mode=0)
    m1 = O.score_pairs(sig, pairs, mode=1)
    assert np.array_equal(m0, m1), "%d of %d pairs round differently" % (int((m0 != m1).sum()), len(pairs))


def test_n4_exact_integer_cosine_and_sklearn_order_agree_on_100000_column_rows():
    """N4 (recommender.py:263-272) at the bench shape -- 2000 users x 100 000 queries, 40 clusters of 50 users:
    the reference's own call, np.around(cosine_similarity(c_scores), 3) on the truncated centred integer rows
    (sklearn: normalise each row in float64, then a BLAS Gram matrix over 100 000 columns), against the
    exact-integer form the device computes (qrlsh/users.py -> qrlsh_score_pairs: int64 dot / (sqrt(na) *
    sqrt(nb)) in float64, rint(. * 1000)) on EVERY pair of every cluster.  The float summation error of a
    100 000-term dot product is far larger than at P = 128; a pair that rounds differently would be a parity
    finding for qrlsh/users.py."""
    from sklearn.metrics.pairwise import cosine_similarity
    rng = np.random.RandomState(7)
    nu, nq, ncl = 2000, 100_000, 40
    labels = rng.randint(0, ncl, size=nu)
    flips = total = 0
    for c in range(ncl):
        members = np.flatnonzero(labels == c)
        block = rng.randint(1, 101, size=(len(members), nq)).astype(np.int64)
        block[rng.rand(len(members), nq) < 0.75] = 0
        if c % 4 == 0:       # correlated users: a shared taste vector, so that cosines leave the neighbourhood of 0
            base = rng.randint(1, 101, size=nq)
            keep = rng.rand(len(members), nq) < 0.5
            block = np.where((block != 0) & keep, base[None, :], block)
        for s in range(len(block)):                         # the reference's in-place centring in an INTEGER array
            nz = block[s] != 0
            block[s][nz] = block[s][nz] - np.mean(block[s][nz])
        ref = np.around(cosine_similarity(block), 3)        # recommender.py:270
        gram = block @ block.T                              # exact in int64: |c| <= 100, 1e5 terms
        norm = np.sqrt(np.diag(gram).astype(np.float64))
        with np.errstate(invalid="ignore", divide="ignore"):
            cos = gram.astype(np.float64) / (norm[:, None] * norm[None, :])
        cos[~np.isfinite(cos)] = 0.0
        milli = np.rint(cos * 1000.0)
        iu = np.triu_indices(len(members), 1)
        total += len(iu[0])
        flips += int((milli[iu] / 1000.0 != ref[iu]).sum())
        assert np.abs(ref[iu]).max() > (0.05 if c % 4 == 0 else 0.0)
    assert total > 45_000
    assert flips == 0, "%d of %d pairs round differently" % (flips, total)
