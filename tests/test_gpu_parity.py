"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the
C ABI (ctypes, qrlsh.ops), against the oracle and the golden vectors captured from the
reference.  Integer / index work is bit-exact; the cosine is compared after the reference's
own rounding to 3 decimals (np.around, recommender.py:203)."""
import os

import numpy as np
import pytest
import torch

from helpers import (FULL, PIECES, GOLDEN, GENERATOR_SETS, load, pairs_u64, check_topk_tie_aware,
                     generator_table_and_queries)

pytestmark = pytest.mark.gpu

import qrlsh  # noqa: E402
from qrlsh import ops, pipeline  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def u64(t):
    return t.cpu().numpy().view(np.uint64)


# ---------------------------------------------------------------------------- a1 / a2
@pytest.mark.parametrize("force_i32", [False, True])
@pytest.mark.parametrize("name", FULL + PIECES)
def test_minhash_matches_reference_golden(name, force_i32):
    g = load(name)
    P, D, b = int(g["P"]), int(g["D"]), int(g["b"])
    perms = ops.legacy_permutations(P, D, seed=int(g["seed"]))
    table = ops.perm_table(perms, DEV, force_i32=force_i32)
    sig, norm2, keys = ops.minhash(dev(g["offsets"]), dev(g["rows"]), table, b=b)
    torch.cuda.synchronize()
    assert np.array_equal(sig.cpu().numpy(), g["sig"])
    assert np.array_equal(norm2.cpu().numpy(), (g["sig"].astype(np.int64) ** 2).sum(1))
    if P // b <= 4:
        assert np.array_equal(u64(keys).T, O.band_keys(g["sig"], b))
    else:   # wide bands: hashed bucket ids -- equal tuples <=> equal ids on this data, empty tuple -> ~0
        lo = (g["sig"].astype(np.int64) & 0xFFFF).reshape(len(g["sig"]), b, P // b)
        k = u64(keys).T
        for band in (0, b - 1):
            _, inv_t = np.unique(lo[:, band, :], axis=0, return_inverse=True)
            _, inv_k = np.unique(k[:, band], return_inverse=True)
            assert len(set(zip(inv_t.ravel().tolist(), inv_k.ravel().tolist()))) == inv_t.max() + 1 == inv_k.max() + 1
        assert np.all(k[np.all(lo == 0xFFFF, axis=2)] == np.uint64(0xFFFFFFFFFFFFFFFF))
    # standalone a2 agrees with the fused one
    k2, n2 = ops.band_keys(sig, b, want_norm=True)
    assert torch.equal(k2, keys) and torch.equal(n2, norm2)
    assert torch.equal(ops.row_norms(sig), norm2)
    if ops.can_compact(table):
        sig16, n16, k16 = ops.minhash(dev(g["offsets"]), dev(g["rows"]), table, b=b, compact=True)
        assert sig16.dtype == torch.int16 and torch.equal(ops.sig_to_int32(sig16), sig)
        assert torch.equal(n16, norm2) and torch.equal(k16, keys)
    else:
        with pytest.raises(ValueError):
            ops.minhash(dev(g["offsets"]), dev(g["rows"]), table, b=b, compact=True)


def test_minhash_odd_shapes():
    rng = np.random.default_rng(3)
    for (nq, D, P, b) in [(1, 7, 5, 5), (3, 70000, 12, 4), (130, 50, 15, 5), (257, 300, 520, 130), (65, 1000, 33, 11)]:
        sets = [np.unique(rng.integers(0, D, size=rng.integers(0, 90))) for _ in range(nq)]
        off = np.zeros(nq + 1, np.int64)
        off[1:] = np.cumsum([len(s) for s in sets])
        rows = (np.concatenate(sets) if nq else np.zeros(0)).astype(np.int32)
        perm = O.legacy_permutations(1, P, D)
        ref = O.minhash(off, rows, perm)
        for force in (False, True):
            table = ops.perm_table(perm, DEV, force_i32=force)
            sig, norm2, keys = ops.minhash(dev(off), dev(rows), table, b=b)
            assert np.array_equal(sig.cpu().numpy(), ref)
            assert np.array_equal(u64(keys).T, O.band_keys(ref, b))


def test_minhash_empty_input_and_errors():
    perm = O.legacy_permutations(0, 8, 16)
    table = ops.perm_table(perm, DEV)
    sig, norm2, keys = ops.minhash(dev(np.zeros(1, np.int64)), dev(np.zeros(0, np.int32)), table, b=4)
    assert sig.shape == (0, 8) and keys.shape == (4, 0)
    with pytest.raises(AssertionError):
        ops.minhash(dev(np.zeros(2, np.int64)), dev(np.zeros(0, np.int32)), table, b=3)
    with pytest.raises(ValueError):           # wide bands need the signatures for verification
        ops.candidate_pairs(dev(np.zeros((2, 4), np.int64)), 5)
    # malformed answer sets are refused before the kernel gathers by their row ids (D = 16 here)
    good_off, good_rows = np.array([0, 2, 2, 5], np.int64), np.array([3, 15, 0, 7, 9], np.int32)
    ops.minhash(dev(good_off), dev(good_rows), table, b=4)
    for off, rows, what in [(good_off, np.array([3, 16, 0, 7, 9], np.int32), "row id outside"),
                            (good_off, np.array([3, -1, 0, 7, 9], np.int32), "row id outside"),
                            (np.array([0, 3, 2, 5], np.int64), good_rows, "offsets decrease"),
                            (np.array([1, 2, 2, 5], np.int64), good_rows, "offsets\\[0\\]"),
                            (np.array([0, 2, 2, 4], np.int64), good_rows, "offsets\\[-1\\]")]:
        with pytest.raises(ValueError, match=what):
            ops.minhash(dev(off), dev(rows), table, b=4)


# ---------------------------------------------------------------------------- sort
@pytest.mark.parametrize("n", [0, 1, 63, 4095, 4096, 4097, 70001])
@pytest.mark.parametrize("nbatch", [1, 3])
def test_radix_sort_matches_numpy_stable(n, nbatch):
    rng = np.random.default_rng(n + nbatch)
    for (lo, hi, bits) in [(0, 64, 64), (0, 24, 20), (32, 56, 64), (8, 16, 64)]:
        k = rng.integers(0, 2 ** bits, size=(nbatch, n), dtype=np.uint64)
        v = rng.integers(0, 2 ** 31, size=(nbatch, n), dtype=np.int64).astype(np.int32)
        ks, vs = ops.sort_u64(dev(k.view(np.int64)), dev(v), lo, hi)
        torch.cuda.synchronize()
        mask = np.uint64((1 << (hi - lo)) - 1)
        for bi in range(nbatch):
            digit = (k[bi] >> np.uint64(lo)) & mask
            order = np.argsort(digit, kind="stable")
            assert np.array_equal(u64(ks)[bi], k[bi][order])
            assert np.array_equal(vs.cpu().numpy()[bi], v[bi][order])
        # keys only
        ks2, none = ops.sort_u64(dev(k.view(np.int64)), None, lo, hi)
        assert none is None and torch.equal(ks2, ks)
    # folded digits: pair words i << 32 | j ordered as i << w | j
    for w in (5, 20, 27):
        i = rng.integers(0, 2 ** w, size=(nbatch, n), dtype=np.uint64)
        j = rng.integers(0, 2 ** w, size=(nbatch, n), dtype=np.uint64)
        k = (i << np.uint64(32)) | j
        ks, _ = ops.sort_u64(dev(k.view(np.int64)), None, 0, 2 * w, fold=w)
        for bi in range(nbatch):
            assert np.array_equal(u64(ks)[bi], np.sort(k[bi]))


def test_mix_sort_groups_equal_keys():
    rng = np.random.default_rng(0)
    n, b = 50000, 3
    k = rng.integers(0, 3000, size=(b, n), dtype=np.uint64) * np.uint64(0x100000001)
    sk, sid = ops.bucket_sort(dev(k.view(np.int64)), hash_bits=32)
    sk, sid = u64(sk), sid.cpu().numpy()
    mix = np.array([[qrlsh._lib.load().qrlsh_mix64_host(int(x)) >> 32 for x in row] for row in sk[:, :2000]])
    assert np.all(np.diff(mix.astype(np.int64), axis=1) >= 0)
    for bi in range(b):
        assert np.array_equal(k[bi][sid[bi]], sk[bi])
        same = sk[bi][1:] == sk[bi][:-1]
        assert np.all(sid[bi][1:][same] > sid[bi][:-1][same])
        # every key forms exactly one run
        nruns = 1 + np.count_nonzero(~same)
        assert nruns == len(np.unique(k[bi]))


# ---------------------------------------------------------------------------- a3
def _rows_case(rng, nq, lens, jmax, dup):
    """pairs grouped by i (rows in ascending i, j shuffled inside a row, each j repeated up to `dup` times)"""
    out = []
    for i, ln in enumerate(lens):
        if ln == 0:
            continue
        base = rng.integers(i + 1, jmax, size=max(1, ln // dup + 1))
        js = rng.choice(base, size=ln)
        out.append((np.uint64(i) << np.uint64(32)) | js.astype(np.uint64))
    return np.concatenate(out) if out else np.zeros(0, np.uint64)


@pytest.mark.parametrize("id_bits", [20, 32])
@pytest.mark.parametrize("case", ["short", "mixed", "long_rows", "chunk_edges", "one_row", "tiny", "empty"])
def test_row_unique_equals_sorted_set(case, id_bits):
    """qrlsh_row_unique_* (rows de-duplicated and ordered in LDS) == np.unique of the same words"""
    rng = np.random.default_rng(11)
    if case == "short":
        lens = rng.poisson(17, size=20000)
    elif case == "mixed":
        lens = rng.poisson(12, size=8000)
        lens[rng.integers(0, 8000, size=40)] = rng.integers(300, 1000, size=40)
    elif case == "long_rows":            # rows past the chunk image go to the one-workgroup-per-row kernel
        lens = rng.poisson(9, size=6000)
        lens[rng.integers(0, 6000, size=25)] = rng.integers(1025, 12288, size=25)
        lens[[0, 5999]] = [12288, 3000]
        lens[3000:3003] = [5000, 4000, 2049]     # back to back
    elif case == "chunk_edges":          # rows of exactly the chunk size and its neighbours, back to back
        lens = np.array([1024, 1, 1023, 1024, 1024, 2, 1020, 5, 1000, 1024, 23, 1024, 1] * 3)
    elif case == "one_row":
        lens = np.array([900])
    elif case == "tiny":
        lens = np.array([1, 0, 2, 1])
    else:
        lens = np.array([], dtype=np.int64)
    words = _rows_case(rng, len(lens), lens, 1 << min(id_bits, 31), dup=4)
    got = ops.row_unique(dev(words.view(np.int64)))
    assert got is not None
    assert np.array_equal(u64(got), np.unique(words))
    if id_bits == 20 and case not in ("long_rows", "chunk_edges"):   # (those would outgrow the long-row table)
        # rows = groups of 2^g consecutive i, in any order of i inside the group
        for g in (1, 4):
            order = np.argsort((words >> np.uint64(32 + g)), kind="stable")
            blocks = words[order]
            sh = rng.permutation(len(blocks))     # shuffle inside groups: sort the shuffled words by group only
            blocks = blocks[sh][np.argsort((blocks[sh] >> np.uint64(32 + g)), kind="stable")]
            got = ops.row_unique(dev(blocks.view(np.int64)), g, 20)
            assert got is not None and np.array_equal(u64(got), np.unique(words)), g


@pytest.mark.parametrize("case,id_bits,g", [("short", 20, 8), ("short", 24, 8), ("mixed", 20, 8), ("mixed", 24, 7),
                                             ("hot", 20, 8), ("hot", 24, 8), ("big", 20, 8), ("big", 24, 6), ("short", 27, 5),
                                             ("short", 29, 3),
                                             ("sparse", 24, 8), ("one", 20, 8), ("empty", 20, 8)])
def test_region_unique_equals_sorted_set(case, id_bits, g):
    """qrlsh_region_unique_* (one workgroup per 2^g consecutive queries, LDS hash set + per-i placement) ==
    np.unique of the same words; the words only need to be ordered by i >> g"""
    rng = np.random.default_rng(21)
    nids = min((1 << id_bits) - 1, 3_000_000)
    nrows = 30000
    if case == "short":
        lens = rng.poisson(17, size=nrows)
    elif case == "mixed":
        lens = rng.poisson(12, size=nrows)
        lens[rng.integers(0, nrows, size=60)] = rng.integers(300, 3000, size=60)
    elif case == "hot":        # single queries with tens of thousands of emitted words, few of them distinct
        lens = rng.poisson(9, size=nrows)
        lens[[0, 777, 778, nrows - 1]] = [40000, 25000, 18000, 30000]
    elif case == "big":        # queries with 7 - 11 thousand DISTINCT partners: the big-image kernel's regions
        lens = rng.poisson(9, size=nrows)
        lens[[5, 300, 900, 9000, nrows - 2]] = [84000, 100000, 90000, 130000, 120000]
    elif case == "sparse":     # most regions empty
        lens = np.zeros(nrows, dtype=np.int64)
        lens[rng.integers(0, nrows, size=300)] = rng.integers(1, 200, size=300)
    elif case == "one":
        lens = np.array([900])
    else:
        lens = np.array([], dtype=np.int64)
    words = _rows_case(rng, len(lens), lens, nids, dup=12 if case in ("hot", "big") else 4)
    words = words[rng.permutation(len(words))]
    grouped = words[np.argsort(words >> np.uint64(32 + g), kind="stable")]     # by region only
    got = ops.region_unique(dev(grouped.view(np.int64)), g, id_bits, nids)
    assert got is not None
    assert np.array_equal(u64(got), np.unique(words))
    # the same straight from the unordered words: dealt into fixed regions by the histogram-free partition
    # (qrlsh_pair_regions_scatter), the region finish on those.  The words sit in the first 30 000 of nids ids: told how
    # many words a query emits, the regions are sized for that density; not told, they overflow ("cap") -- never a wrong list
    wpq = len(words) / max(1, len(lens))
    got, why = ops.region_unique_scattered(dev(words.view(np.int64)), g, id_bits, nids, words_per_query=wpq)
    if -(-nids // (1 << g)) > 65536:           # more regions than two levels of 256 digits reach: not served
        assert got is None and why == "cap"
    elif case in ("hot", "big", "mixed", "one"):   # single queries with tens of thousands of words outgrow any sensible
        # region ("one": 900 words per query would size 4096 regions of 700 K words -- refused as too large)
        assert (got is None and why == "cap") or np.array_equal(u64(got), np.unique(words))
    else:
        assert why == "" and np.array_equal(u64(got), np.unique(words))
    got2, why2 = ops.region_unique_scattered(dev(words.view(np.int64)), g, id_bits, nids)
    assert (got2 is None and why2 == "cap") or np.array_equal(u64(got2), np.unique(words))


def test_region_unique_reports_overflow_and_unique_pairs_falls_back():
    rng = np.random.default_rng(22)
    lens = rng.poisson(10, size=3000)
    lens[1500] = 60000                    # one i with ~15000 DISTINCT partners: more than even the big image holds
    words = _rows_case(rng, len(lens), lens, 1 << 20, dup=4)
    grouped = words[np.argsort(words >> np.uint64(40), kind="stable")]
    assert ops.region_unique(dev(grouped.view(np.int64)), 8, 20, 1 << 20) is None
    stats = {}
    got = ops.unique_pairs(dev(words[rng.permutation(len(words))].view(np.int64)), (1 << 20) - 1, stats)
    assert stats["dedup_path"] == "full-sort" and np.array_equal(u64(got), np.unique(words))
    stats = {}
    lens[1500] = 10
    words = _rows_case(rng, len(lens), lens, 1 << 20, dup=3)
    got = ops.unique_pairs(dev(words[rng.permutation(len(words))].view(np.int64)), (1 << 20) - 1, stats)
    assert stats["dedup_path"].startswith("regions-in-lds") and stats["group_bits"] == 8
    assert np.array_equal(u64(got), np.unique(words))
    # id widths that leave no room for group bits keep the single-i rows
    stats = {}
    got = ops.unique_pairs(dev(words[rng.permutation(len(words))].view(np.int64)), (1 << 31) - 1, stats)
    assert stats["dedup_path"] == "rows-in-lds" and np.array_equal(u64(got), np.unique(words))


def test_row_unique_reports_overflow_and_unique_pairs_falls_back():
    rng = np.random.default_rng(12)
    lens = rng.poisson(10, size=3000)
    lens[1500] = 20000                    # one i with more emitted pairs than even the long-row kernel's table holds
    words = _rows_case(rng, len(lens), lens, 1 << 20, dup=3)
    assert ops.row_unique(dev(words.view(np.int64))) is None
    shuffled = words[rng.permutation(len(words))]
    stats = {}
    got = ops.unique_pairs(dev(shuffled.view(np.int64)), 1 << 31, stats)     # 32-bit ids: the single-i row form
    assert stats["dedup_path"] == "full-sort"
    assert np.array_equal(u64(got), np.unique(words))
    stats = {}
    lens[1500] = 10
    words = _rows_case(rng, len(lens), lens, 1 << 20, dup=3)
    got = ops.unique_pairs(dev(words[rng.permutation(len(words))].view(np.int64)), 1 << 31, stats)
    assert stats["dedup_path"] == "rows-in-lds" and np.array_equal(u64(got), np.unique(words))


@pytest.mark.parametrize("name", FULL + PIECES)
def test_candidates_match_reference_golden(name):
    from lsh import LSH
    g = load(name)
    l = LSH(int(g["b"]))
    l.compute_buckets_batch(g["sig"])
    stats = {}
    arr = l.get_candidates_array(stats)
    assert np.array_equal(u64(arr), np.sort(pairs_u64(g["pairs"])))
    if int(g["P"]) // int(g["b"]) <= 4:
        assert stats["emitted_pairs"] == O.emitted_pairs(O.band_keys(g["sig"], int(g["b"])), int(g["P"]) // int(g["b"]))
    else:
        assert stats["hash_collision_pairs_dropped"] == 0


def test_lsh_dropin_surface_and_edge_semantics():
    from lsh import LSH
    g = load("lsh_edge")
    sig, b = g["sig"], int(g["b"])
    l = LSH(b)
    for s in sig:
        l.compute_buckets(s)           # one call per query, as recommender.py:173-174 does
    assert l.counter == len(sig)
    cands = l.get_candidates(sig)
    assert isinstance(cands, set)
    assert sorted(cands) == [tuple(x) for x in g["pairs"].tolist()]
    assert np.array_equal(l.make_subvecs(sig[2]), g["subvecs_row2"])
    with pytest.raises(AssertionError):
        LSH(5).make_subvecs(sig[0])
    # no state leaks between instances (the reference's class-level buckets do leak)
    l2 = LSH(b)
    assert l2.counter == 0 and l2.get_candidates(None) == set()
    # lazily materialised dict buckets look like the reference's
    bk = l.buckets
    assert len(bk) == b and all(isinstance(d, dict) for d in bk)
    ref = O.naive_candidates(sig, b)
    from itertools import combinations
    mine = set()
    for d in bk:
        for key, hits in d.items():
            if len(hits) > 1 and set(key.split(",")) != {"-1"}:
                mine.update(combinations(hits, 2))
    assert mine == ref


def test_duplicate_heavy_bucket():
    # 300 identical queries: every band has a bucket of 300 -> 44850 pairs, 32x duplicated
    sig = np.tile(np.arange(128, dtype=np.int32)[None, :], (300, 1))
    extra = np.random.default_rng(1).integers(0, 30000, size=(100, 128)).astype(np.int32)
    allsig = np.concatenate([extra[:50], sig, extra[50:]])
    keys = ops.band_keys(dev(allsig), 32)
    st = {}
    pairs = ops.candidate_pairs(keys, 4, st)
    ref = O.candidates(O.band_keys(allsig, 32), 4)
    assert np.array_equal(u64(pairs), ref)
    assert len(ref) == 300 * 299 // 2 and st["emitted_pairs"] == 32 * len(ref)


# ---------------------------------------------------------------------------- a5
@pytest.mark.parametrize("name", FULL + PIECES)
def test_scores_match_reference_rounding(name):
    g = load(name)
    sig = dev(g["sig"])
    pairs = dev(pairs_u64(g["pairs"]).view(np.int64))
    norm2 = ops.row_norms(sig)
    milli, cosv, _ = ops.score_pairs(sig, norm2, pairs, want_cos=True)
    # np.around(x, 3) == rint(x * 1000) / 1000
    assert np.array_equal(milli.cpu().numpy() / 1000.0, g["pair_cos"])
    m_ref, c_ref = O.score_pairs(g["sig"], pairs_u64(g["pairs"]), mode=1, want_cos=True)
    assert np.array_equal(milli.cpu().numpy(), m_ref)
    assert np.array_equal(cosv.cpu().numpy(), c_ref)  # same correctly-rounded fp64 formula
    if int(g["D"]) <= 65535:                           # compact uint16 rows: identical scores
        s16 = dev(np.where(g["sig"] < 0, 0xFFFF, g["sig"]).astype(np.uint16).view(np.int16))
        m16, c16, _ = ops.score_pairs(s16, norm2, pairs, want_cos=True)
        assert torch.equal(m16, milli) and torch.equal(c16, cosv)


def test_score_non_multiple_of_4_and_zero_rows():
    rng = np.random.default_rng(2)
    sig = rng.integers(-1, 1000, size=(50, 15)).astype(np.int32)
    sig[7] = 0
    pairs = np.array([(i << 32) | j for i in range(50) for j in range(i + 1, 50)], dtype=np.uint64)
    d = dev(sig)
    milli, _, _ = ops.score_pairs(d, ops.row_norms(d), dev(pairs.view(np.int64)))
    assert np.array_equal(milli.cpu().numpy(), O.score_pairs(sig, pairs, mode=1))
    assert np.array_equal(milli.cpu().numpy(), O.score_pairs(sig, pairs, mode=0))
    # norms summed inside the kernel (norm2 = None) == precomputed norms: int32 rows (odd and vector widths) and
    # compact uint16 rows with -1 entries
    for P in (15, 16, 128):
        s32 = rng.integers(-1, 30000, size=(50, P)).astype(np.int32)
        s32[3] = -1
        s32[9] = 0
        ds = dev(s32)
        ref = O.score_pairs(s32, pairs, mode=1)
        assert np.array_equal(ops.score_pairs(ds, None, dev(pairs.view(np.int64)))[0].cpu().numpy(), ref)
        assert np.array_equal(ops.score_pairs(ds, ops.row_norms(ds), dev(pairs.view(np.int64)))[0].cpu().numpy(), ref)
        if P % 8 == 0:
            d16 = dev(np.where(s32 < 0, 0xFFFF, s32).astype(np.uint16).view(np.int16))
            assert np.array_equal(ops.score_pairs(d16, None, dev(pairs.view(np.int64)))[0].cpu().numpy(), ref)
            m2, rev = ops.score_pairs_rev(d16, None, dev(pairs.view(np.int64)), 6)
            assert np.array_equal(m2.cpu().numpy(), ref)


@pytest.mark.parametrize("name", FULL)
def test_full_path_matches_reference_tie_aware(name):
    g = load(name)
    P, D, b, K = int(g["P"]), int(g["D"]), int(g["b"]), int(g["K"])
    table = ops.perm_table(ops.legacy_permutations(P, D, seed=int(g["seed"])), DEV)
    res = pipeline.query_similarities(dev(g["offsets"]), dev(g["rows"]), table, b, K)
    torch.cuda.synchronize()
    assert np.array_equal(res.sig_int32().cpu().numpy(), g["sig"])
    assert np.array_equal(u64(res.pairs), np.sort(pairs_u64(g["pairs"])))
    check_topk_tie_aware(g, u64(res.pairs), res.milli.cpu().numpy(), res.src.cpu().numpy(), res.dst.cpu().numpy(),
                         res.val.cpu().numpy(), K)
    # and exactly equal to the oracle, which uses the same documented tie-break
    r = O.query_similarities(g["offsets"], g["rows"], D, P, b, K, int(g["seed"]))
    assert np.array_equal(res.src.cpu().numpy(), r["src"])
    assert np.array_equal(res.dst.cpu().numpy(), r["dst"])
    assert np.array_equal(res.val.cpu().numpy(), r["val"])


@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_recommender_dropin_on_generator_default_inputs(sub):
    """config 1: the CSVs produced by the reference's resources/generator.py, through the
    drop-in Recommender, under the same np.random.seed as the golden capture."""
    import pandas as pd
    import recommender as R
    g = load(sub + "_hotpath")
    gdir = os.path.join(GOLDEN, sub)
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
    rec = R.Recommender()
    rec.verbose = False
    rec.datasetFeatures = feats
    rec.dataset = dataset
    rec.queries = np.array(qrows, dtype=object)
    rec.queriesIDs = np.array(qids)
    rec.tupleCount = {}
    R.PERM = int(g["P"])
    off, rows = rec.answer_sets()
    assert np.array_equal(off, g["offsets"]) and np.array_equal(rows, g["rows"])
    np.random.seed(int(g["seed"]))
    sig = rec.compute_signatures()
    assert sig.dtype == np.int64 and np.array_equal(sig, g["sig"])
    np.random.seed(int(g["seed"]))
    qs = rec.compute_querySimilarities()
    assert sorted(qs.keys()) == sorted(int(q) for q in g["qs_q"])
    res = rec.last_result
    assert res.b == int(g["b"]) and res.K == int(g["K"])
    check_topk_tie_aware(g, u64(res.pairs), res.milli.cpu().numpy(), res.src.cpu().numpy(), res.dst.cpu().numpy(),
                         res.val.cpu().numpy(), int(g["K"]))
    for q, e in qs.items():
        assert e["indexes"].dtype == np.int64 and e["values"].dtype == np.float64
    R.PERM = 128
    with pytest.raises(ValueError):
        rec._band_rule()
    R.PERM = 180


# ---------------------------------------------------------------------------- synthetic + sizes
def test_synth_generator_matches_oracle_twin():
    for (nq, D, q0, nl) in [(5000, 32768, 0, None), (5000, 100000, 1234, 777), (64, 50, 0, None)]:
        off, rows = qrlsh.synth_csr(nq, D, seed=5, q0=q0, nq_local=nl, device=DEV)
        roff, rrows = O.synth_csr(nq, D, seed=5, q0=q0, nq_local=nl)
        assert np.array_equal(off.cpu().numpy(), roff)
        assert np.array_equal(rows.cpu().numpy(), rrows)


@pytest.mark.parametrize("nq,D,P,b", [(20000, 32768, 128, 32), (30000, 100000, 128, 32), (12000, 32768, 256, 64),
                                      (200000, 32768, 128, 32), (25000, 32768, 100, 20), (9000, 70000, 96, 12),
                                      (30000, 50000, 128, 32),    # compact rows whose values pass 2^15: no dot2 form
                                      (1_000_000, 100000, 128, 32)])   # SURVEY 8d's parity-only run: int32 table and rows, int16 wrap in the keys
def test_pipeline_equals_oracle_on_synthetic(nq, D, P, b):
    K = pipeline.max_candidates(nq)
    O.set_threads(16)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    ho, hr = off.cpu().numpy(), rows.cpu().numpy()
    sig = O.minhash(ho, hr, perms)
    assert np.array_equal(res.sig_int32().cpu().numpy(), sig)
    pairs = O.candidates_from_sig(sig, b)
    assert np.array_equal(u64(res.pairs), pairs)
    milli = O.score_pairs(sig, pairs, mode=1)
    assert np.array_equal(res.milli.cpu().numpy(), milli)
    s, d, v = O.topk(pairs, milli, K)
    assert np.array_equal(res.src.cpu().numpy(), s)
    assert np.array_equal(res.dst.cpu().numpy(), d)
    assert np.array_equal(res.val.cpu().numpy(), v)


def test_full_size_config2_properties_and_oracle():
    """BASELINE config 2 (1 M queries, P=128, b=32): exact equality with the oracle for the
    integer stages (it finishes in seconds on the box's host cores) plus size-independent
    properties of the outputs."""
    nq, D, P, b = 1_000_000, 32768, 128, 32
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    pairs = u64(res.pairs)
    i, j = (pairs >> np.uint64(32)).astype(np.int64), (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all(i < j) and j.max() < nq
    assert np.all(pairs[1:] > pairs[:-1])                       # sorted, unique
    src, dst, val = res.src.cpu().numpy(), res.dst.cpu().numpy(), res.val.cpu().numpy()
    key = (src.astype(np.int64) << 32) | (1000 - val).astype(np.int64) << 21 | dst
    assert np.all(key[1:] > key[:-1])                           # (src, value desc, dst asc), no repeats
    assert np.bincount(src).max() <= K
    assert val.min() >= -1000 and val.max() <= 1000
    # every kept edge is a candidate pair
    ek = (np.minimum(src, dst).astype(np.uint64) << np.uint64(32)) | np.maximum(src, dst).astype(np.uint64)
    assert np.all(np.isin(ek[::97], pairs))
    # oracle, exact
    ho, hr = off.cpu().numpy(), rows.cpu().numpy()
    sig = O.minhash(ho, hr, perms)
    assert np.array_equal(res.sig_int32().cpu().numpy(), sig)
    opairs = O.candidates(O.band_keys(sig, b), P // b)
    assert np.array_equal(pairs, opairs)
    assert np.array_equal(res.milli.cpu().numpy(), O.score_pairs(sig, opairs, mode=1))


def _check_properties(res, nq, K):
    """size-independent properties of a hot-path result; -> (pairs u64, src, dst, val) host arrays"""
    pairs = u64(res.pairs)
    i, j = (pairs >> np.uint64(32)).astype(np.int64), (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all(i < j) and j.max() < nq
    assert np.all(pairs[1:] > pairs[:-1])                       # sorted, unique
    src, dst, val = res.src.cpu().numpy(), res.dst.cpu().numpy(), res.val.cpu().numpy()
    same = src[1:] == src[:-1]                                  # (src, value desc, dst asc), no repeats
    assert np.all((src[1:] > src[:-1]) | (same & ((val[1:] < val[:-1]) | ((val[1:] == val[:-1]) & (dst[1:] > dst[:-1])))))
    assert np.bincount(src).max() <= K
    assert val.min() >= -1000 and val.max() <= 1000
    ek = (np.minimum(src, dst).astype(np.uint64) << np.uint64(32)) | np.maximum(src, dst).astype(np.uint64)
    assert np.all(np.isin(ek[::997], pairs))                    # kept edges are candidate pairs
    return pairs, src, dst, val


def _check_against_oracle(res, off, rows, perms, b, K, nq):
    pairs, src, dst, val = _check_properties(res, nq, K)
    P = perms.shape[0]
    sig = O.minhash(off.cpu().numpy(), rows.cpu().numpy(), perms)
    assert np.array_equal(res.sig_int32().cpu().numpy(), sig)
    opairs = O.candidates(O.band_keys(sig, b), P // b)
    assert np.array_equal(pairs, opairs)
    milli = O.score_pairs(sig, opairs, mode=1)
    assert np.array_equal(res.milli.cpu().numpy(), milli)
    s, d, v = O.topk(opairs, milli, K)
    assert np.array_equal(src, s) and np.array_equal(dst, d) and np.array_equal(val, v)


def test_full_size_config3_equals_oracle():
    """BASELINE configs[2] = SURVEY config 3 (10 M queries, P=128, b=32, one GPU): the workload bench.py
    times.  Two-step partition (T > 8), popular queries through the big-image and histogram kernels; signatures,
    candidate pairs, scores and top-K are compared exactly with the oracle (16 s on the box's host cores)."""
    nq, D, P, b = 10_000_000, 32768, 128, 32
    K = pipeline.max_candidates(nq)
    assert K == 40
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    assert res.stats["bucket_path"] == "partition+lds" and res.stats["dedup_path"] == "regions-in-lds (scattered)"
    assert res.stats["part_bits"] > 8                         # two-step partition
    O.set_threads(16)
    _check_against_oracle(res, off, rows, perms, b, K, nq)
    del res
    torch.cuda.empty_cache()


def test_full_size_256_perm_64_bands_equals_oracle():
    """the config-4/5 shape (P=256, b=64) at 1 M queries, exact against the oracle"""
    nq, D, P, b = 1_000_000, 32768, 256, 64
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    O.set_threads(16)
    _check_against_oracle(res, off, rows, perms, b, K, nq)


def test_more_than_2_pow_24_queries_equals_oracle():
    """nq > 2^24 (the per-rank record count of configs 4-5): ids need more than 24 bits of the partition's id
    word (which carries nothing else: the partition stores mix64(key) and takes every part number from it),
    T = 12; tiny P / b keep it cheap.  Also drives the count-then-fill API, whose sort-based partition takes its non-staged
    scatter for nq > 2^24 (sort.hip: bucket_partition), and the plain-layout size rules."""
    nq, D, P, b = 17_000_000, 32768, 8, 2
    assert nq > (1 << 24)
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, DEV)
    res = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    assert res.stats["bucket_path"] == "partition+lds"
    O.set_threads(16)
    _check_against_oracle(res, off, rows, perms, b, K, nq)
    # count-then-fill API on the same keys: the sort-based partition, non-staged for nq > 2^24
    _, _, keys = ops.minhash(off, rows, table, b=b, compact=True)
    slow = ops.emit_pairs_fast(keys, P // b, one_pass=False)
    assert slow is not None and slow.numel() == res.stats["emitted_pairs"]
    assert np.array_equal(u64(ops.unique_pairs(slow, nq)), u64(res.pairs))
    # keys as a band-partitioned all-to-all delivers them, read in place at nq > 2^24
    world = 4
    nql = nq // world
    recv = keys.view(b, world, nql).permute(1, 0, 2).contiguous()
    chunked = ops.emit_pairs_fast(recv.view(-1), P // b, chunks=(world, b, nql))
    assert chunked is not None and chunked.numel() == slow.numel()
    assert np.array_equal(u64(ops.unique_pairs(chunked, nq)), u64(res.pairs))


def test_wide_ids_beyond_2_pow_26_queries():
    """nq > 2^26: two ids + 11 score bits do not fit one 64-bit top-K key, the key + payload edge format
    is selected by the size itself (not forced).  One band of four permutations keeps the work small;
    checked through size-independent properties and, exactly, against the oracle's candidate / score /
    top-K stages fed with the device's own (already verified kernel) signatures."""
    nq, D, P, b = 68_000_000, 32768, 4, 1
    assert ops.wide_ids(ops.id_bits_for(nq))
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, mean=4.0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    pairs, src, dst, val = _check_properties(res, nq, K)
    sig = res.sig_int32().cpu().numpy()
    n_chk = 2_000_000                                             # signatures: first 2 M queries against the oracle
    ho = off[: n_chk + 1].cpu().numpy()
    assert np.array_equal(sig[:n_chk], O.minhash(ho, rows[: int(ho[-1])].cpu().numpy(), perms))
    O.set_threads(16)
    opairs = O.candidates(O.band_keys(sig, b), P // b)
    assert np.array_equal(pairs, opairs)
    milli = O.score_pairs(sig, opairs, mode=1)
    assert np.array_equal(res.milli.cpu().numpy(), milli)
    s, d, v = O.topk(opairs, milli, K)
    assert np.array_equal(src, s) and np.array_equal(dst, d) and np.array_equal(val, v)


# ---------------------------------------------------------------------------- sharded driver
def _run_dist_gpu(tmp_path, world, nq, D, P, b, mode, backend, port, sig_mode="auto", mean=16.0, env_extra=None):
    import subprocess
    import sys as _sys
    torch.cuda.synchronize()
    torch.cuda.empty_cache()     # the ranks share this process's GPU: give back what earlier tests left cached
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    # the ranks are started directly (env-variable rendezvous), not through torch.distributed.run: its agent
    # process counts against the 6 processes a GPU box admits on its card (this process + at most 5 ranks)
    args = [_sys.executable, os.path.join(root, "tests", "dist_gpu_worker.py"),
            str(tmp_path), str(nq), str(D), str(P), str(b), mode, backend, sig_mode, str(mean)]
    procs, logs = [], []
    for rk in range(world):
        e = dict(env, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        logs.append(open(os.path.join(str(tmp_path), "rank%d.log" % rk), "w+"))
        procs.append(subprocess.Popen(args, env=e, stdout=logs[-1], stderr=subprocess.STDOUT))
    failed, deadline = False, 900
    for pr in procs:
        try:
            pr.wait(timeout=deadline)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
        failed = failed or pr.returncode != 0
        if failed:
            deadline = 20          # a peer is gone: the others cannot finish their collectives
    texts = []
    for f in logs:
        f.seek(0)
        texts.append(f.read()[-3000:])
        f.close()
    assert not failed, "\n".join("--- rank %d (rc %s)\n%s" % (k, pr.returncode, t)
                                  for k, (pr, t) in enumerate(zip(procs, texts)))
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]


def _check_sharded_against(outs, res, nq, world):
    """the ranks' outputs against a single-GPU HotPathResult of the same queries"""
    from dist_worker import pair_host
    nql = -(-nq // world)
    pairs = np.concatenate([o["pairs"] for o in outs]).view(np.uint64)
    milli = np.concatenate([o["milli"] for o in outs])
    order = np.argsort(pairs, kind="stable")
    assert np.array_equal(pairs[order], u64(res.pairs))            # disjoint shares, union = the single-GPU list
    assert np.array_equal(milli[order], res.milli.cpu().numpy())
    for r, o in enumerate(outs):
        p = o["pairs"].view(np.uint64)
        assert np.all(p[1:] > p[:-1]) and np.all(pair_host(p, nql) == r)
    assert np.array_equal(np.concatenate([o["src"] for o in outs]), res.src.cpu().numpy())
    assert np.array_equal(np.concatenate([o["dst"] for o in outs]), res.dst.cpu().numpy())
    assert np.array_equal(np.concatenate([o["val"] for o in outs]), res.val.cpu().numpy())
    assert sum(int(o["emitted"]) for o in outs) == res.stats["emitted_pairs"]


@pytest.mark.parametrize("world,mode,backend,sig_mode,nq", [(1, "all_to_all", "nccl", "auto", 40000),
                                                            (2, "all_to_all", "gloo", "auto", 40000),
                                                            (4, "all_gather", "gloo", "fetch", 40000),
                                                            (3, "all_to_all", "gloo", "fetch", 40001),   # padded last shard
                                                            (2, "all_to_all", "gloo", "all_gather", 39999),
                                                            (3, "all_to_all", "gloo", "sets", 40001),
                                                            (5, "all_to_all", "gloo", "auto", 40003),    # auto = sets
                                                            (3, "all_to_all", "gloo", "recompute", 40001),
                                                            (2, "all_to_all", "gloo", "recompute", 30001)])
def test_sharded_driver_on_gpu_equals_single_gpu(tmp_path, world, mode, backend, sig_mode, nq):
    D, P, b = 32768, 128, 32
    if nq == 30001:      # 15001 rows of 100 uint16 per shard: the row blocks of the replicated table lose the
        P, b = 100, 20   # kernel's 16-byte alignment (copy path); r = 5: hashed bucket ids + verification
    outs = _run_dist_gpu(tmp_path, world, nq, D, P, b, mode, backend, 29571 + world + len(sig_mode) + nq % 5, sig_mode)
    nql = -(-nq // world)
    for o in outs:
        if sig_mode != "auto":
            assert str(o["sig_exchange"]) == sig_mode
        if world > 1 and str(o["sig_exchange"]) == "fetch":
            assert 0 <= int(o["fetched"]) <= (world - 1) * nql   # only rows of the other shards, each once
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    perms = ops.legacy_permutations(P, D, seed=42)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, DEV), b, K)
    torch.cuda.synchronize()
    _check_sharded_against(outs, res, nq, world)
    # ... and the ranks' outputs against the ORACLE directly (not only through the one-GPU result): pairs, scores, top-K
    ref = O.query_similarities(off.cpu().numpy(), rows.cpu().numpy(), D, P, b, K, 42) if P // b <= 4 else None
    if ref is not None:
        pairs = np.concatenate([o["pairs"] for o in outs]).view(np.uint64)
        order = np.argsort(pairs, kind="stable")
        assert np.array_equal(pairs[order], ref["pairs"])
        assert np.array_equal(np.concatenate([o["milli"] for o in outs])[order], ref["milli"])
        for k in ("src", "dst", "val"):
            assert np.array_equal(np.concatenate([o[k] for o in outs]), ref[k])


def test_rccl_executes_every_collective_shape_at_world_1():
    """RCCL (backend "nccl") on the one GPU of this box: every collective shape the N-rank step issues -- variable
    all-to-all with split lists (int64 / int32 / int16 rows as a 2-D byte view / empty), the size exchange, synchronous
    and asynchronous all-gathers on the second communicator -- moves the right bytes, and the sharded driver with
    force_collectives=True (no one-rank short cut: bucket-id exchange, pair hosting exchange, remote-row machinery,
    edge exchange, received-edge top-K) equals the one-GPU pipeline bit for bit in every exchange / signature mode,
    at 40 k and at 1 M queries (tests/dist_nccl_worker.py; N > 1 needs more GPUs than this pool hands out)."""
    import subprocess
    import sys as _sys
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    p = subprocess.run([_sys.executable, os.path.join(root, "tests", "dist_nccl_worker.py"), "40000", "1000000"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    assert p.stdout.count("forced world-1 step ok") == 10 and "collective shapes ok" in p.stdout


def test_sharded_driver_retries_a_too_small_pair_buffer(tmp_path):
    """the one-pass emit sizes its output from what the last call of the shape produced (ops._EMIT_HINT); a
    guess that is far too small (here: 16 pairs, on every rank) must cost a second, exactly sized run and
    nothing else -- same pairs, scores and top-K as the one-GPU pipeline"""
    nq, D, P, b, world = 30000, 32768, 128, 32, 2
    outs = _run_dist_gpu(tmp_path, world, nq, D, P, b, "all_to_all", "gloo", 29597, "fetch",
                         env_extra={"QRLSH_TEST_TINY_EMIT_HINT": "1"})
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    res = pipeline.query_similarities(off, rows, ops.perm_table(ops.legacy_permutations(P, D, seed=42), DEV), b, K)
    torch.cuda.synchronize()
    _check_sharded_against(outs, res, nq, world)


def test_sharded_config3_shape_four_ranks(tmp_path):
    """BASELINE configs[3]'s workload (10 M queries x 128 / 32) through the sharded driver, four ranks sharing the
    GPU (gloo, collectives staged through the host): two-step partition per owned band over all 10 M
    queries, balanced pair hosting, row fetch, edge exchange, re-based top-K -- equal to the one-GPU result"""
    nq, D, P, b, world = 10_000_000, 32768, 128, 32, 4
    outs = _run_dist_gpu(tmp_path, world, nq, D, P, b, "all_to_all", "gloo", 29611, "fetch")
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    res = pipeline.query_similarities(off, rows, ops.perm_table(ops.legacy_permutations(P, D, seed=42), DEV), b, K)
    torch.cuda.synchronize()
    _check_sharded_against(outs, res, nq, world)
    sizes = [len(o["pairs"]) for o in outs]
    assert max(sizes) < 1.05 * sum(sizes) / world                # the scoring work is split evenly


def test_sharded_config3_shape_five_ranks_all_gather_of_bucket_ids(tmp_path):
    """configs[3] with the exchange BASELINE's north_star names -- an ALL-GATHER of the bucket ids -- and the
    signatures of remote queries computed from the replicated answer sets ("sets": what `auto` picks for the 8-GPU run;
    the four-rank test above covers the row fetch), at the largest rank count this box allows: a GPU box admits
    at most 6 processes on its card: five ranks + this process (the 8-rank layout -- 4 bands per rank, shards of
    1.25 M -- is covered on the CPU by tests/test_dist_cpu.py at world 8 and, rank by rank at configs[4]'s size, by
    test_config4_rank_slice below).  32 bands over 5 ranks is an UNEVEN band split (7, 7, 7, 7, 4)."""
    nq, D, P, b, world = 10_000_000, 32768, 128, 32, 5
    outs = _run_dist_gpu(tmp_path, world, nq, D, P, b, "all_gather", "gloo", 29631, "sets")
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    res = pipeline.query_similarities(off, rows, ops.perm_table(ops.legacy_permutations(P, D, seed=42), DEV), b, K)
    torch.cuda.synchronize()
    _check_sharded_against(outs, res, nq, world)
    sizes = [len(o["pairs"]) for o in outs]
    assert max(sizes) < 1.05 * sum(sizes) / world
    for o in outs:
        assert str(o["sig_exchange"]) == "sets"
    del res
    torch.cuda.empty_cache()


def _host_threads():
    """OpenMP threads for the oracle on this box: every core the process may run on, capped by a cgroup quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("QRLSH_TEST_THREADS", "64"))))


def test_config4_one_rank_at_true_volume():
    """BASELINE configs[4] (100 M queries x 256-perm MinHash, 64 bands, 8 ranks) as ONE rank of the eight sees it, at
    the volume it really receives, through to the top-K rows of its 12.5 M queries.  Everything the other seven ranks
    would contribute is computed here too (their answer sets, their keys, the words they emit):

      1. rank 0's MinHash of its 12.5 M queries (P = 256); the other shards' likewise (they stand for what arrives);
      2. ALL 64 bands, as eight emitter ranks of 8 bands each: for emitter e the keys of its bands over all 100 M ids in
         the [rank][band][queries] layout the band-partitioned all-to-all delivers, partition (T = 15, two steps, 27-bit
         ids) + LDS finish, the emitted words grouped by scoring rank (qr_pair_host) -- rank 0's share of each is what
         emitter e sends it: ~93 M words x 8 = ~740 M received words;
      3. ONE de-duplication of everything received (region form, 27 id bits);
      4. the signatures of the remote queries its pairs touch by the "sets" route: remote-id set, their answer sets
         taken out of the replicated [world][max_nnz] / [world][nql + 1] arrays (16-bit row ids, 32-bit offsets, as
         gathered by qrlsh.dist), ONE MinHash over them;
      5. scoring against the two-piece row table [own rows | computed remote rows];
      6. both directed edges of every pair, grouped by owner; the edges rank 0 keeps, plus the edges the other ranks
         send it (those of the pairs THEY host that touch rank 0's queries: de-duplicated and scored here on their
         behalf), re-based (qrlsh_edges_localize) and cut by the select-form top-K with ties by id.

    Checked exactly against the oracle: the signatures of samples of the first / last shard; for every emitter the
    candidate pairs of its 8 bands x 100 M keys (oracle bucketing), filtered by host -- their union equals the
    de-duplicated pairs; the remote signatures against the rows the owners computed; the scores of a 2 M-pair sample
    and, against the one-piece table, of all; the final top-K rows of rank 0's first 200 000 queries against the
    oracle's top-K over every candidate pair of all 64 bands that touches them; and through size-independent
    properties for the rest.  The oracle's bucketing of 8 x 100 M keys per emitter is most of this test's time: it is
    done for as many emitters as QRLSH_CFG4_ORACLE_BUDGET_S (default 330 s) allows -- all eight on the box's cores when
    nothing else loads them; with fewer, the pair list is checked to CONTAIN the oracle's pairs of the checked emitters
    and the top-K cut is compared with numpy on the kernel's own input edges."""
    import time
    from qrlsh import dist as qdist
    from dist_worker import pair_host
    nq, D, P, b, world, rank = 100_000_000, 32768, 256, 64, 8, 0
    r = P // b
    K = pipeline.max_candidates(nq)
    q0, n_real, nql = qdist.shard_range(nq, world, rank)
    ranges = qdist.band_owner_ranges(b, world)
    nb = ranges[rank][1] - ranges[rank][0]
    assert (nql, nb, n_real, K, q0) == (12_500_000, 8, 12_500_000, 45, 0)
    oracle_budget = float(os.environ.get("QRLSH_CFG4_ORACLE_BUDGET_S", "330"))
    exact_groups = 0
    S = 200_000                                   # queries whose final top-K rows are compared with the oracle's
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, DEV)
    O.set_threads(_host_threads())
    stage_ms = {}
    t_test = time.perf_counter()

    def timed(label, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stage_ms[label] = stage_ms.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    # ---- 1. every shard's answer sets, signatures, norms and band keys
    keys_all = torch.empty((world, world, nb, nql), dtype=torch.int64, device=DEV)   # [emitter][rank][band][query]: 51 GB
    sig_all = torch.empty((nq, P), dtype=torch.int16, device=DEV)                    # 51 GB (the owners' rows: checks only)
    norm_all = torch.empty((nq,), dtype=torch.int64, device=DEV)
    keys = torch.empty((b, nql), dtype=torch.int64, device=DEV)
    rows16, offs32 = [], []
    for s_ in range(world):
        off, rows = qrlsh.synth_csr(nq, D, seed=0, q0=s_ * nql, nq_local=nql, device=DEV)
        blk = slice(s_ * nql, (s_ + 1) * nql)
        if s_ == rank:
            timed("1  MinHash of the rank's 12.5 M queries (P = 256, keys + norms fused)",
                  lambda: ops.minhash(off, rows, table, b=b, compact=True, validate=False, out=(sig_all[blk], norm_all[blk], keys)))
        ops.minhash(off, rows, table, b=b, compact=True, validate=(s_ == rank), out=(sig_all[blk], norm_all[blk], keys))
        for e in range(world):
            keys_all[e, s_].copy_(keys[ranges[e][0]:ranges[e][1]])
        if s_ in (0, world - 1):        # the first 100 000 signatures of the first / last shard against the oracle
            n_chk = 100_000
            ho = off[: n_chk + 1].cpu().numpy()
            osig = O.minhash(ho, rows[: int(ho[-1])].cpu().numpy(), perms)
            assert np.array_equal(ops.sig_to_int32(sig_all[s_ * nql: s_ * nql + n_chk]).cpu().numpy(), osig)
            assert np.array_equal(keys[:, :n_chk].cpu().numpy().view(np.uint64), O.band_keys(osig, b).T)
        rows16.append(rows.to(torch.int16))      # what the answer-set gather of qrlsh.dist puts on the wire
        offs32.append(off.to(torch.int32))
        del off, rows
    del keys
    max_nnz = max(t.numel() for t in rows16)
    ra_w = torch.zeros((world, max_nnz), dtype=torch.int16, device=DEV)
    oa_w = torch.stack(offs32)
    for s_ in range(world):
        ra_w[s_, : rows16[s_].numel()].copy_(rows16[s_])
    del rows16, offs32

    # ---- 2. the eight emitters: partition + finish of 8 bands x 100 M ids each, rank 0's share of the words
    be = qdist.HipBackend()
    assert ops.part_bits_for(nq) == 15
    received, others, host_parts, sample_parts = [], [], [], []
    n_emitted = 0
    t_oracle = 0.0
    for e in range(world):
        recv = keys_all[e].view(-1)
        if e == 0:
            be.emit_pairs_chunked(recv, world, nb, nql, r)            # (settles the pair-buffer size guess)
        emitted = timed("3  partition + finish, 8 bands x 100 M ids (per emitter)", lambda: be.emit_pairs_chunked(recv, world, nb, nql, r))
        assert be.stats["bucket_path"] == "partition+lds", "emitter %d fell to the general sort path (6 x slower)" % e
        n_emitted += emitted.numel()
        grouped, _ = timed("4a grouping of an emitter's words by scoring rank", lambda: ops.sort_u64(emitted, None, host_shard=nql))
        del emitted
        bounds = ops.owner_bounds(grouped, -1, nql, world).tolist()
        shares = np.diff(bounds)
        assert shares.max() < 1.05 * shares.mean()                     # the coin splits evenly
        received.append(grouped[bounds[rank]:bounds[rank + 1]].clone())
        # what the OTHER ranks host of the pairs that touch rank 0's queries (i < j and q0 = 0: i below nql)
        rest = grouped[bounds[rank + 1]:]
        others.append(rest[rest < (nql << 32)].clone())
        del grouped, rest
        if exact_groups == e and t_oracle * (e + 1) <= oracle_budget * max(e, 1):   # (the next one still fits the budget)
            t0 = time.perf_counter()
            kq = keys_all[e].permute(0, 2, 1).reshape(nq, nb).contiguous().cpu().numpy().view(np.uint64)   # [query][band]
            opairs = O.candidates(kq, r)
            del kq
            host_parts.append(O.filter_pair_host(opairs, nql, rank))
            sample_parts.append(opairs[: int(np.searchsorted(opairs, np.uint64(S) << np.uint64(32)))].copy())
            del opairs
            t_oracle += time.perf_counter() - t0
            exact_groups = e + 1
    del keys_all
    torch.cuda.empty_cache()

    # ---- 3. one de-duplication of everything rank 0 received
    got = torch.cat(received)
    n_received = got.numel()
    del received
    stats = {}
    pairs = timed("4b de-duplication of the %d words received from the eight emitters" % n_received,
                  lambda: ops.unique_pairs(got, nq, stats, words_per_query=n_received / (2 * nql)))
    del got
    # (at this volume the words are dominated by the pairs of giant buckets -- a query of a 20 000-member bucket has
    #  thousands of DISTINCT partners, more than a region's LDS set holds -- so the region form may give up and the
    #  general sort + unique take over: same result; which one ran is printed below)
    dedup_path = stats["dedup_path"]
    hp = u64(pairs)
    assert np.all(hp[1:] > hp[:-1])
    i, j = (hp >> np.uint64(32)).astype(np.int64), (hp & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all(i < j) and j.max() < nq
    assert np.all(pair_host(hp[::37], nql) == rank)
    if exact_groups >= world:
        t0 = time.perf_counter()
        want = O.sort_unique(np.concatenate(host_parts))
        assert np.array_equal(hp, want)
        del want
        t_oracle += time.perf_counter() - t0
    else:                                # every pair the oracle hosts here from the checked emitters is present
        for part in host_parts:
            assert np.all(np.isin(part[::17], hp))
    del host_parts

    # ---- 4. "sets": remote ids, their answer sets out of the replicated arrays, one MinHash
    rid = timed("5a remote-id set of the hosted pairs", lambda: ops.remote_ids(pairs, q0, nql, nq, world))
    n_remote = int(rid.bounds[-1].item())
    need = ops.remote_id_list(rid, n_remote)
    touched = np.unique(np.concatenate([i[::64], j[::64]]))
    hn = need.cpu().numpy()
    assert np.all(hn[1:] > hn[:-1]) and np.all(hn >= q0 + nql) and np.all(np.isin(touched[touched >= nql], hn))
    del touched, i, j
    off_b, rows_b = timed("5b answer sets of the remote queries out of the replicated shards", lambda: ops.gather_sets(need, oa_w, ra_w, nql))
    sig_b = torch.empty((n_remote, P), dtype=torch.int16, device=DEV)        # (25 GB: allocated outside the timed call)
    norm_b = torch.empty((n_remote,), dtype=torch.int64, device=DEV)
    timed("5c MinHash of the %d remote queries the pairs touch" % n_remote,
          lambda: ops.minhash(off_b, rows_b, table, b=None, want_norm=True, compact=True, validate=False, out=(sig_b, norm_b, None)))
    del off_b, rows_b, ra_w, oa_w
    for c0 in range(0, n_remote, 8_000_000):      # = the rows their owners computed
        idx = need[c0:c0 + 8_000_000]
        assert torch.equal(sig_b[c0:c0 + 8_000_000], sig_all[idx]) and torch.equal(norm_b[c0:c0 + 8_000_000], norm_all[idx])
    del idx

    # ---- 5. scoring against the two-piece table
    local = ops.remap_pairs_ids(pairs, rid)
    milli = timed("6  scoring of the %d hosted pairs (two-piece row table)" % pairs.numel(),
                  lambda: ops.score_pairs_split(sig_all[q0:q0 + nql], norm_all[q0:q0 + nql], sig_b, norm_b, local))
    whole, _, _ = ops.score_pairs(sig_all, norm_all, pairs)            # the same pairs against the one-piece table
    assert torch.equal(milli, whole)
    del whole, sig_b, norm_b, local, rid, need
    step = max(1, len(hp) // 2_000_000)        # oracle scores on a sample of 2 M pairs (rows re-indexed into a small table)
    sp = hp[::step]
    si, sj = (sp >> np.uint64(32)).astype(np.int64), (sp & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ids = np.unique(np.concatenate([si, sj]))
    small = ops.sig_to_int32(sig_all[torch.from_numpy(ids).to(DEV)]).cpu().numpy()
    rp = (np.searchsorted(ids, si).astype(np.uint64) << np.uint64(32)) | np.searchsorted(ids, sj).astype(np.uint64)
    assert np.array_equal(milli.cpu().numpy()[::step], O.score_pairs(small, rp, mode=1))
    del small, rp, sp, si, sj, ids

    # ---- 6. edges: kept here + sent by the other ranks; re-based; select-form top-K
    ib = ops.id_bits_for(nq)
    assert ib == 27 and ops.wide_ids(ib)
    ek, ed = timed("7a both directed edges of every hosted pair", lambda: be.edges(pairs, milli, ib, True))
    ek, ed, ebounds = timed("7b grouping of the edges by the owner of their src", lambda: be.group_edges_by_owner(ek, ed, 11, nql, world))
    eb = ebounds.tolist()
    assert eb[-1] == 2 * pairs.numel()
    keep_k, keep_d = ek[eb[rank]:eb[rank + 1]].clone(), ed[eb[rank]:eb[rank + 1]].clone()
    del ek, ed
    # the other ranks' part: the pairs they host that touch rank 0's queries, scored on their behalf
    ow = torch.cat(others)
    del others
    opairs_dev = ops.unique_pairs(ow, nq, {}, words_per_query=ow.numel() / (2 * nql))
    del ow
    assert not bool(torch.isin(opairs_dev[::1009], pairs).any())      # hosted elsewhere: disjoint from the pairs scored here
    omilli, _, _ = ops.score_pairs(sig_all, norm_all, opairs_dev)
    ok_, od_ = be.edges(opairs_dev, omilli, ib, True)
    ok_, od_, ob_ = be.group_edges_by_owner(ok_, od_, 11, nql, world)
    ob = ob_.tolist()
    ein = torch.cat([keep_k, ok_[ob[rank]:ob[rank + 1]]])
    din = torch.cat([keep_d, od_[ob[rank]:ob[rank + 1]]])
    n_kept, n_sent = keep_k.numel(), ob[rank + 1] - ob[rank]
    del keep_k, keep_d, ok_, od_
    src, dst, val = timed("7c top-K of the rank's 12.5 M queries on the %d edges received (select form, ties by id)" % ein.numel(),
                          lambda: be.topk_local(ein, din, K, ib, q0, nql))
    hs, hd, hv = src.cpu().numpy(), dst.cpu().numpy(), val.cpu().numpy()
    assert hs.min() >= q0 and hs.max() < q0 + nql
    same = hs[1:] == hs[:-1]                                     # (src, value desc, dst asc), no repeats, <= K per src
    assert np.all((hs[1:] > hs[:-1]) | (same & ((hv[1:] < hv[:-1]) | ((hv[1:] == hv[:-1]) & (hd[1:] > hd[:-1])))))
    assert np.bincount(hs).max() <= K
    # the cut itself, exactly, for the first S queries: numpy on the very edges the kernel was given
    m_s = ein < (S << 11)                                        # (wide-id edge key = src << 11 | 1000 - milli; dst beside it)
    hk, edd = ein[m_s].cpu().numpy().view(np.uint64), din[m_s].cpu().numpy().astype(np.int64)
    es, einv = (hk >> np.uint64(11)).astype(np.int64), (hk & np.uint64(0x7FF)).astype(np.int64)
    del hk, m_s, ein, din
    o = np.lexsort((edd, einv, es))
    es, einv, edd = es[o], einv[o], edd[o]
    kp = np.ones(len(es), dtype=bool)
    kp[K:] = es[K:] != es[:-K]
    cut = int(np.searchsorted(hs, S))
    assert np.array_equal(hs[:cut], es[kp]) and np.array_equal(hd[:cut], edd[kp]) and np.array_equal(hv[:cut], 1000 - einv[kp])
    if exact_groups >= world:            # ... and end to end: the oracle's candidates of all 64 bands that touch those queries
        t0 = time.perf_counter()
        sp = O.sort_unique(np.concatenate(sample_parts))
        si, sj = (sp >> np.uint64(32)).astype(np.int64), (sp & np.uint64(0xFFFFFFFF)).astype(np.int64)
        ids = np.unique(np.concatenate([si, sj]))
        small = ops.sig_to_int32(sig_all[torch.from_numpy(ids).to(DEV)]).cpu().numpy()
        rp = (np.searchsorted(ids, si).astype(np.uint64) << np.uint64(32)) | np.searchsorted(ids, sj).astype(np.uint64)
        sm = O.score_pairs(small, rp, mode=1)
        os_, od2, ov = O.topk(sp, sm, K)
        c2 = int(np.searchsorted(os_, S))
        assert np.array_equal(hs[:cut], os_[:c2]) and np.array_equal(hd[:cut], od2[:c2]) and np.array_equal(hv[:cut], ov[:c2])
        t_oracle += time.perf_counter() - t0
    print("configs[4] one rank at true volume: emitted by the eight emitters %d words, received here %d, unique hosted pairs %d, "
          "remote queries touched %d of %d, edges kept %d + received %d, top-K rows %d; de-duplication path %s; oracle-exact "
          "emitters %d of 8 (oracle %.0f s of %.0f s)" % (n_emitted, n_received, len(hp), n_remote, nq - nql, n_kept, n_sent, len(hs),
                                                          dedup_path, min(exact_groups, world), t_oracle, time.perf_counter() - t_test))
    for k_ in sorted(stage_ms):
        n_calls = 8 if k_.startswith(("3 ", "4a")) else 1
        print("configs[4] one rank at true volume:   %-110s %8.1f ms%s" % (k_, stage_ms[k_], " (eight emitters: %.1f each)" % (stage_ms[k_] / 8) if n_calls == 8 else ""))
    del sig_all, norm_all
    torch.cuda.empty_cache()


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """the driver's scaling command is the plain `python bench.py --gpus N ...`: with WORLD_SIZE unset bench.py must
    start the N ranks itself (as child processes, before anything touches the GPU), relay rank 0's ONE JSON line and
    return the launcher's code.  Rehearsed here with two gloo ranks sharing this GPU on a small workload: the line is
    well-formed, names two ranks and carries the same candidate / top-K counts as the one-GPU pipeline."""
    import json
    import subprocess
    import sys as _sys
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    nq = 200_000
    p = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu",
                        "--steps", "2", "--warmup", "1", "--nq-total", str(nq), "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout, everything else on stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "signatures/s"
    assert d["config"]["queries_total"] == nq and d["config"]["queries_per_rank"] == nq // 2
    assert d["config"]["signature_exchange"] == "recompute" and d["value"] > 0
    assert set(d["phases_rank0"]["bytes_sent"]) == {"0_answer_sets", "4_pairs", "6_edges"}
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, 32768, seed=0, device=DEV)
    res = pipeline.query_similarities(off, rows, ops.perm_table(ops.legacy_permutations(128, 32768, seed=42), DEV), 32, K)
    assert (d["unique_pairs"], d["kept_edges"], d["emitted_pairs"]) == (res.pairs.numel(), res.src.numel(),
                                                                       res.stats["emitted_pairs"])


def test_sharded_driver_wide_ids_beyond_2_pow_26(tmp_path):
    """config-5-sized id space through the SHARDED driver on the device: nq_total > 2^26 (key + payload
    edges, re-based to 64-bit local keys on arrival), more than 2^24 records per owned band (32-bit ids
    through the partition, which reads the exchanged key layout in place), two gloo ranks sharing the GPU."""
    nq, D, P, b, world = 68_000_000, 32768, 8, 2, 2       # r = 4: buckets stay small (r = 2 keys collide in the thousands)
    outs = _run_dist_gpu(tmp_path, world, nq, D, P, b, "all_to_all", "gloo", 29597, "fetch")
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    res = pipeline.query_similarities(off, rows, ops.perm_table(ops.legacy_permutations(P, D, seed=42), DEV), b, K)
    torch.cuda.synchronize()
    assert ops.wide_ids(ops.id_bits_for(nq))
    _check_sharded_against(outs, res, nq, world)


def test_multi_gpu_glue_kernels_match_numpy():
    from dist_worker import pair_host
    rng = np.random.default_rng(3)
    nq, q0, nql, n, world = 100000, 40000, 20000, 50000, 5
    a = rng.integers(q0, q0 + nql, size=n).astype(np.uint64)          # one endpoint local ...
    c = rng.integers(0, nq, size=n).astype(np.uint64)                 # ... the other anywhere
    c[c == a] += np.uint64(1)
    i, j = np.minimum(a, c), np.maximum(a, c)
    pairs = (i << np.uint64(32)) | j
    # scoring rank of a pair: one pass that groups by it + the split points
    g, _ = ops.sort_u64(dev(pairs.view(np.int64)), None, host_shard=nql)
    host = pair_host(pairs, nql)
    order = np.argsort(host, kind="stable")
    assert np.array_equal(u64(g), pairs[order])
    assert np.array_equal(ops.owner_bounds(g, -1, nql, world).cpu().numpy(), np.searchsorted(host[order], np.arange(world + 1)))
    assert 0.4 < np.mean(host == (i // np.uint64(nql)).astype(np.int64)) < 0.9    # a coin, not "always the smaller id"
    # remote id set, request list, re-indexed pairs
    ends = np.concatenate([i, j]).astype(np.int64)
    need = np.unique(ends[(ends < q0) | (ends >= q0 + nql)])
    rid = ops.remote_ids(dev(pairs.view(np.int64)), q0, nql, nq, world)
    assert np.array_equal(rid.bounds.cpu().numpy(), np.searchsorted(need, np.arange(world + 1) * nql))
    assert np.array_equal(ops.remote_id_list(rid, len(need)).cpu().numpy(), need)

    def slot(x):
        x = x.astype(np.int64)
        return np.where((x >= q0) & (x < q0 + nql), x - q0, nql + np.searchsorted(need, x)).astype(np.uint64)
    local = u64(ops.remap_pairs_ids(dev(pairs.view(np.int64)), rid))
    assert np.array_equal(local, (slot(i) << np.uint64(32)) | slot(j))
    rid0 = ops.remote_ids(dev(pairs[:0].view(np.int64)), q0, nql, nq + 7, world)        # no pairs, nids % 32 != 0
    assert not rid0.bounds.cpu().numpy().any()
    # rows on request, and scoring against the two-piece table == scoring against the whole one
    P = 24
    sig = rng.integers(-1, 30000, size=(nq, P)).astype(np.int32)
    norm = (sig.astype(np.int64) ** 2).sum(1)
    for sg in (dev(sig), dev(np.where(sig < 0, 0xFFFF, sig).astype(np.uint16).view(np.int16))):
        rows_b, norms_b = ops.gather_rows(sg, dev(norm), dev(need), 0)
        assert np.array_equal(rows_b.cpu().numpy(), sg.cpu().numpy()[need]) and np.array_equal(norms_b.cpu().numpy(), norm[need])
        whole = ops.score_pairs(sg, dev(norm), dev(pairs.view(np.int64)))[0]
        split = ops.score_pairs_split(sg[q0:q0 + nql].contiguous(), dev(norm[q0:q0 + nql]), rows_b, norms_b, dev(local.view(np.int64)))
        assert np.array_equal(split.cpu().numpy(), whole.cpu().numpy())
    # edges: interleaved, packed and key + payload; re-based at the owner
    milli = rng.integers(0, 1001, size=n).astype(np.int32)
    inv = (1000 - milli).astype(np.uint64)
    ib = 17
    e = u64(ops.pair_edges_interleaved(dev(pairs.view(np.int64)), dev(milli), ib))
    assert np.array_equal(e[0::2], (i << np.uint64(ib + 11)) | (inv << np.uint64(ib)) | j)
    assert np.array_equal(e[1::2], (j << np.uint64(ib + 11)) | (inv << np.uint64(ib)) | i)
    ek, ed = ops.pair_edges_interleaved(dev(pairs.view(np.int64)), dev(milli), ib, wide=True)
    assert np.array_equal(u64(ek)[0::2], (i << np.uint64(11)) | inv) and np.array_equal(ed.cpu().numpy()[1::2], i.astype(np.int32))
    fwd, rev = ops.pair_edges(dev(pairs.view(np.int64)), dev(milli), ib)
    assert np.array_equal(u64(fwd), e[0::2]) and np.array_equal(u64(rev), e[1::2])
    # top-K at the owner from edges in arbitrary order == the one-GPU top-K restricted to its queries
    K = 5
    s0, d0, v0 = ops.topk_edges(dev(e.view(np.int64)), K, ib)           # pair order is NOT sorted here: sort first
    sp = np.sort(pairs)
    o2 = np.argsort(pairs, kind="stable")
    e_sorted = u64(ops.pair_edges_interleaved(dev(sp.view(np.int64)), dev(milli[o2]), ib))
    s0, d0, v0 = (t.cpu().numpy() for t in ops.topk_edges(dev(e_sorted.view(np.int64)), K, ib))
    mine = (e >> np.uint64(ib + 11) >= np.uint64(q0)) & (e >> np.uint64(ib + 11) < np.uint64(q0 + nql))
    shuffled = e[mine][rng.permutation(int(mine.sum()))]
    s1, d1, v1 = (t.cpu().numpy() for t in ops.topk_edges_local(dev(shuffled.view(np.int64)), None, K, ib, q0, nql))
    sel = (s0 >= q0) & (s0 < q0 + nql)
    assert np.array_equal(s1, s0[sel]) and np.array_equal(d1, d0[sel]) and np.array_equal(v1, v0[sel])
    wk, wd = u64(ek)[mine], ed.cpu().numpy()[mine]
    perm = rng.permutation(len(wk))
    s2, d2, v2 = (t.cpu().numpy() for t in ops.topk_edges_local(dev(wk[perm].view(np.int64)), dev(wd[perm]), K, ib, q0, nql))
    assert np.array_equal(s2, s1) and np.array_equal(d2, d1) and np.array_equal(v2, v1)


# ---------------------------------------------------------------------------- fast bucket path
@pytest.mark.parametrize("wide", [False, True])
def test_local_topk_select_form_equals_sort_form(wide):
    """the top-K of the sharded driver's step 7: the edges a rank receives (any order, from several scoring ranks),
    re-based to its id range, cut by the select form (sorted on the src bits only, every edge ranks itself in its
    query's run) and by the full (src, value, dst) sort -- identical COO, short / medium / long lists and ties included"""
    rng = np.random.default_rng(31 + wide)
    ib, q0, nql, K = (27 if wide else 22), 3_000_000, 50_000, 40
    deg = rng.integers(0, 30, size=nql)
    deg[rng.choice(nql, 40, replace=False)] = rng.integers(100, 3000, size=40)      # long lists
    src = np.repeat(np.arange(nql), deg) + q0
    n = len(src)
    dst = rng.integers(0, 1 << ib, size=n)
    inv = rng.integers(0, 40, size=n)                                               # few values: many ties at the cut
    perm = rng.permutation(n)
    src, dst, inv = src[perm], dst[perm], inv[perm]
    if wide:
        keys = (src.astype(np.uint64) << np.uint64(11)) | inv.astype(np.uint64)
        e, d = dev(keys.view(np.int64)), dev(dst.astype(np.int32))
    else:
        keys = (src.astype(np.uint64) << np.uint64(ib + 11)) | (inv.astype(np.uint64) << np.uint64(ib)) | dst.astype(np.uint64)
        e, d = dev(keys.view(np.int64)), None
    # the data must be able to tell "ties at the cut in arrival order" from "ties at the cut by ascending id" (what the
    # sort form -- and the reference's documented tie-break -- keeps): count the long lists where the two differ
    differ = 0
    for q in np.flatnonzero(deg > 64)[:20]:
        sel = np.flatnonzero(src == q + q0)                      # arrival order
        iv, ids = inv[sel], dst[sel]
        cut = np.sort(iv)[K - 1]
        ties, want = np.flatnonzero(iv == cut), K - int((iv < cut).sum())
        differ += set(ids[ties[:want]]) != set(np.sort(ids[ties])[:want])
    assert differ > 0
    a = ops.topk_edges_local(e.clone(), d, K, ib, q0, nql, select=True)
    b = ops.topk_edges_local(e.clone(), d, K, ib, q0, nql, select=False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    s_, _, v_ = (t.cpu().numpy() for t in a)
    assert s_.min() >= q0 and s_.max() < q0 + nql and np.bincount(s_ - q0).max() == K and len(s_) == int(np.minimum(deg, K).sum())
    assert (v_ == 1000 - np.sort(inv)[0]).any()


def test_gather_sets_out_of_replicated_shards():
    """qrlsh_gather_sets_* ("sets" mode of the sharded driver): the answer sets of chosen global query ids out of the
    per-shard arrays as the all-gather leaves them -- 16- / 32-bit row ids, 32- / 64-bit offsets, padded shards --
    equal the sets themselves"""
    rng = np.random.default_rng(9)
    world, nql, D = 5, 3001, 70000
    offs, rows = [], []
    for g in range(world):
        lens = rng.integers(0, 40, size=nql)
        if g == world - 1:
            lens[-700:] = 0                               # padded tail of the last shard
        o = np.concatenate(([0], np.cumsum(lens)))
        offs.append(o)
        rows.append(rng.integers(0, D, size=int(o[-1])))
    max_nnz = max(len(r) for r in rows)
    ids = np.sort(rng.choice(world * nql, size=4000, replace=False)).astype(np.int64)
    want_off = np.concatenate(([0], np.cumsum([offs[q // nql][q % nql + 1] - offs[q // nql][q % nql] for q in ids])))
    want_rows = np.concatenate([rows[q // nql][offs[q // nql][q % nql]:offs[q // nql][q % nql + 1]] for q in ids])
    for odt, rdt in ((np.int32, np.int16), (np.int64, np.int32), (np.int32, np.int32)):
        oa = np.stack(offs).astype(odt)
        ra = np.zeros((world, max_nnz), dtype=np.int64)
        for g in range(world):
            ra[g, :len(rows[g])] = rows[g] if rdt == np.int32 else rows[g] % 65536
        ra = ra.astype(np.uint16).view(np.int16) if rdt == np.int16 else ra.astype(np.int32)
        off, got = ops.gather_sets(dev(ids), dev(oa), dev(ra), nql)
        assert np.array_equal(off.cpu().numpy(), want_off)
        assert np.array_equal(got.cpu().numpy(), want_rows if rdt == np.int32 else want_rows % 65536)
    off, got = ops.gather_sets(dev(ids[:0]), dev(oa), dev(ra), nql)
    assert off.tolist() == [0] and got.numel() == 0


def test_fast_bucket_path_equals_general_path():
    rng = np.random.default_rng(11)
    for (nq, b, nkeys) in [(1, 2, 5), (50, 3, 7), (5000, 4, 900), (70000, 8, 20000), (300000, 2, 40)]:
        k = rng.integers(0, nkeys, size=(b, nq), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        k[0, : nq // 3] = np.uint64(0xFFFFFFFFFFFFFFFF)         # "empty" keys must never pair (lsh.py:47)
        if nkeys == 40:
            # 300000 / 40 = 7500 copies per key: beyond the LDS image.  With the big-part limit lowered to one image the
            # partition path reports the overflow (None: the caller takes the general path); at the default limit
            # such parts are worked in blocks (test_popular_keys_beyond_the_lds_image_stay_on_the_fast_path)
            from qrlsh import _lib
            old = _lib.load().qrlsh_set_big_part_limit(6144)
            try:
                with_cap = ops.emit_pairs_fast(dev(k.view(np.int64)), 4)
            finally:
                _lib.load().qrlsh_set_big_part_limit(old)
            assert with_cap is None
            continue
        sk, sid = ops.bucket_sort(dev(k.view(np.int64)))
        gen = np.sort(u64(ops.emit_pairs(sk, sid, 4)))
        for T in (None, 9, 12, 16):                               # one- and two-pass partitions
            for kw in (dict(one_pass=False),                      # count-then-fill
                       dict(),                                    # cursor-reserved ranges, default capacity guess
                       dict(capacity=max(1, gen.size // 3)),      # guess too small: counted, retried exactly sized
                       dict(capacity=gen.size)):                  # exact
                fast = ops.emit_pairs_fast(dev(k.view(np.int64)), 4, part_bits=T, **kw)
                assert fast is not None and fast.numel() == gen.size
                assert np.array_equal(np.sort(u64(fast)), gen)


@pytest.mark.parametrize("nq,b,world,T", [(40000, 6, 4, None), (90000, 3, 2, 9), (4096, 5, 8, None)])
def test_chunked_key_layout_is_read_in_place(nq, b, world, T):
    """keys as a band-partitioned all-to-all delivers them ([rank][band][queries of the rank]) give the
    same pairs as the band-major matrix"""
    rng = np.random.default_rng(nq)
    k = rng.integers(0, nq // 3, size=(b, nq), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    k[1, ::7] = np.uint64(0xFFFFFFFFFFFFFFFF)
    nql = nq // world
    recv = np.ascontiguousarray(k.reshape(b, world, nql).transpose(1, 0, 2))      # [world][b][nql]
    plain = ops.emit_pairs_fast(dev(k.view(np.int64)), 4, part_bits=T)
    chunked = ops.emit_pairs_fast(dev(recv.view(np.int64)), 4, part_bits=T, chunks=(world, b, nql))
    assert plain is not None and chunked is not None and chunked.numel() == plain.numel()
    assert np.array_equal(np.sort(u64(chunked)), np.sort(u64(plain)))


def _heavy_key_case(rng, nq, b, heavy):
    """band-major keys [b][nq]: ordinary buckets of 2 .. 5 members + in band t one key with heavy[t] copies"""
    keys = rng.integers(1, 1 << 62, size=(b, nq), dtype=np.int64)
    for band in range(b):
        for size in (2, 3, 5):
            ids = rng.choice(nq, size=(2000, size), replace=False)
            keys[band, ids] = rng.integers(1, 1 << 62, size=(2000, 1), dtype=np.int64)
    for band, n in enumerate(heavy):
        if n:
            keys[band, rng.choice(nq, size=n, replace=False)] = 0x1234567 + band
    return keys


def _check_emitted_against_oracle(keys, r, want_path, heavy):
    stats = {}
    emitted = ops.emit_pairs_any(dev(keys), r, stats)
    assert stats["bucket_path"] == want_path, stats
    kq = np.ascontiguousarray(keys.T).view(np.uint64)
    want = O.candidates(kq, r)
    assert emitted.numel() == O.emitted_pairs(kq, r) >= sum(n * (n - 1) // 2 for n in heavy)
    got = np.unique(u64(emitted))
    assert np.array_equal(got, want)          # (a pair of queries that share both bands' heavy keys is emitted twice)
    del emitted
    torch.cuda.empty_cache()
    return stats


def test_popular_keys_beyond_the_lds_image_stay_on_the_fast_path():
    """lsh.py:42-53 on buckets with thousands of members (at 100 M queries over D = 32768 rows the band key of the
    luckiest (row, band) is shared by ~20 000 unrelated queries: configs[4]).  A part's region holds ONE LDS image of
    the finish; what a popular key adds spills into the overflow pool, the part's records are gathered there and
    bucket_finish_big_kernel works them in blocks: 7 000 copies (two blocks), 13 000 (three, cross joins between all of
    them), 40 000 (seven blocks), beside ordinary small buckets -- exact against the oracle, and without leaving the
    partition + LDS path.  Only a key with more copies than the limit (qrlsh_set_big_part_limit; here lowered to
    12 000 records) takes the general path, with the same result."""
    from qrlsh import _lib
    nq, b, r = 1_000_000, 2, 4
    rng = np.random.default_rng(21)
    _check_emitted_against_oracle(_heavy_key_case(rng, nq, b, (7000, 13000)), r, "partition+lds", (7000, 13000))
    _check_emitted_against_oracle(_heavy_key_case(rng, nq, b, (7000, 40000)), r, "partition+lds", (7000, 40000))
    old = _lib.load().qrlsh_set_big_part_limit(12_000)
    try:
        _check_emitted_against_oracle(_heavy_key_case(rng, nq, b, (7000, 13000)), r, "general-sort", (7000, 13000))
    finally:
        assert _lib.load().qrlsh_set_big_part_limit(old) == 12_000


def test_popular_key_in_the_small_part_form_of_the_finish():
    """the 512-thread / 4096-slot form of the finish (parts of 1024 .. 2800 records on average: 5.5 M queries at T = 11
    are 2686 per part) with popular keys of every kind it hands on: a part between 4097 and 6144 records (1 700
    copies: one block of the big kernel, its region's 4096 records + a spill), one beyond 6144 (4 000 copies: two
    blocks), one far beyond (15 000) -- exact against the oracle, on the partition + LDS path"""
    nq, b, r = 5_500_000, 3, 4
    assert ops.part_bits_for(nq) == 11 and 1024 <= (nq >> 11) <= 2800
    rng = np.random.default_rng(22)
    heavy = (1700, 4000, 15000)
    O.set_threads(_host_threads())
    _check_emitted_against_oracle(_heavy_key_case(rng, nq, b, heavy), r, "partition+lds", heavy)


def test_overflowing_part_falls_back_to_general_path():
    """tiny input (regions of 2 x mean + 128 records): 7001 identical signatures spill into the pool and stay on the
    partition path; with the big-part limit lowered below that the step takes the general path -- same pairs"""
    from qrlsh import _lib
    sig = np.random.default_rng(5).integers(0, 30000, size=(9000, 4)).astype(np.int32)
    sig[1000:8000] = sig[0]                                      # 7001 identical -> one part > FIN_CAP
    ref = O.candidates(O.band_keys(sig, 1), 4)
    assert len(ref) >= 7001 * 7000 // 2
    st = {}
    pairs = ops.candidate_pairs(ops.band_keys(dev(sig), 1), 4, st)
    assert st["bucket_path"] == "partition+lds"
    assert np.array_equal(u64(pairs), ref)
    old = _lib.load().qrlsh_set_big_part_limit(6144)
    try:
        st = {}
        pairs = ops.candidate_pairs(ops.band_keys(dev(sig), 1), 4, st)
        assert st["bucket_path"] == "general-sort"
        assert np.array_equal(u64(pairs), ref)
    finally:
        _lib.load().qrlsh_set_big_part_limit(old)
    st2 = {}
    ops.candidate_pairs(ops.band_keys(dev(sig[:900]), 1), 4, st2)
    assert st2["bucket_path"] == "partition+lds"


# ---------------------------------------------------------------------------- N2: answer sets
def test_device_answer_sets_match_reference_and_oracle():
    from qrlsh import answers
    for sub in GENERATOR_SETS:
        g = load(sub + "_hotpath")
        cols, queries = generator_table_and_queries(sub)
        idx = answers.build_answer_index(cols, DEV)
        off, rows = answers.answer_sets(idx, answers.encode_queries(idx, queries))
        assert np.array_equal(off.cpu().numpy(), g["offsets"]) and np.array_equal(rows.cpu().numpy(), g["rows"])
    # random tables: D not a multiple of 32, > 64 words per row, absent values, unconstrained queries
    rng = np.random.default_rng(9)
    for (D, nfeat, card, nq) in [(1, 1, 1, 3), (33, 2, 3, 40), (1000, 5, 7, 300), (70001, 3, 50, 500), (5000, 6, 2, 200)]:
        cols = [rng.integers(0, card, size=D).astype(str) for _ in range(nfeat)]
        q = np.full((nq, nfeat), "", dtype=object)
        for i in range(nq):
            for f in range(nfeat):
                u = rng.random()
                if u < 0.5:
                    q[i, f] = str(rng.integers(0, card))
                elif u < 0.55:
                    q[i, f] = "no-such-value"
        q[0, :] = ""                                   # fully unconstrained -> every table row
        idx = answers.build_answer_index(cols, DEV)
        off, rows = answers.answer_sets(idx, answers.encode_queries(idx, q))
        roff, rrows = O.answer_sets(cols, q)
        assert np.array_equal(off.cpu().numpy(), roff) and np.array_equal(rows.cpu().numpy(), rrows)
        assert int(off[1]) == D


# ---------------------------------------------------------------------------- degenerate inputs
def test_pipeline_degenerate_inputs():
    P, D, b = 16, 50, 4
    perms = ops.legacy_permutations(P, D, seed=1)
    table = ops.perm_table(perms, DEV)

    def run(sets, K=5):
        off = np.zeros(len(sets) + 1, np.int64)
        off[1:] = np.cumsum([len(s) for s in sets])
        rows = (np.concatenate(sets) if len(sets) and off[-1] else np.zeros(0)).astype(np.int32)
        res = pipeline.query_similarities(dev(off), dev(rows), table, b, K)
        torch.cuda.synchronize()
        ref = O.query_similarities(off, rows, D, P, b, K, 1)
        assert np.array_equal(res.sig_int32().cpu().numpy(), ref["sig"])
        assert np.array_equal(u64(res.pairs), ref["pairs"])
        assert np.array_equal(res.milli.cpu().numpy(), ref["milli"])
        assert np.array_equal(res.src.cpu().numpy(), ref["src"]) and np.array_equal(res.dst.cpu().numpy(), ref["dst"])
        assert np.array_equal(res.val.cpu().numpy(), ref["val"])
        return res

    e = np.zeros(0, np.int32)
    assert run([]).pairs.numel() == 0                                   # no queries at all
    assert run([np.array([3, 7], np.int32)]).pairs.numel() == 0         # one query
    assert run([e, e, e, e]).pairs.numel() == 0                         # only empty answer sets: never candidates
    r = run([np.array([1, 2, 3], np.int32)] * 2 + [e])                  # two identical queries + an empty one
    assert u64(r.pairs).tolist() == [1] and r.val.cpu().tolist() == [1000, 1000]
    r = run([np.array([5], np.int32)] * 40, K=3)                        # one bucket of 40, K smaller than the degree
    assert r.pairs.numel() == 40 * 39 // 2 and np.bincount(r.src.cpu().numpy()).tolist() == [3] * 40
    # top-3 of query 0 under the documented tie-break: all values tie at 1000 -> smallest ids
    assert r.dst.cpu().numpy()[:3].tolist() == [1, 2, 3]


def test_verify_pairs_is_the_exact_candidate_predicate():
    g = load("pieces_p96_b12")                                   # r = 8
    sig, b = g["sig"], int(g["b"])
    n = len(sig)
    rng = np.random.default_rng(4)
    true_pairs = pairs_u64(g["pairs"])
    rnd = np.array([(min(i, j) << 32) | max(i, j) for i, j in rng.integers(0, n, size=(3000, 2)) if i != j], dtype=np.uint64)
    allp = np.unique(np.concatenate([true_pairs, rnd]))
    truth = np.isin(allp, true_pairs)
    for s_dev in (dev(sig), dev(np.where(sig < 0, 0xFFFF, sig).astype(np.uint16).view(np.int16))):
        flags = ops.verify_pairs(s_dev, b, dev(allp.view(np.int64))).cpu().numpy().astype(bool)
        assert np.array_equal(flags, truth)
        kept = ops.drop_unverified(s_dev, b, dev(allp.view(np.int64)))
        assert np.array_equal(u64(kept), true_pairs)


def test_wide_id_edge_format_gives_the_same_topk():
    nq, D, P, b = 30000, 32768, 128, 32
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=DEV)
    table = ops.perm_table(ops.legacy_permutations(P, D, seed=42), DEV)
    a = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    for kw in (dict(wide_ids=True), dict(topk="sort"), dict(topk="sort", wide_ids=True)):
        w = pipeline.query_similarities(off, rows, table, b, K, **kw)
        torch.cuda.synchronize()
        assert torch.equal(a.src, w.src) and torch.equal(a.dst, w.dst) and torch.equal(a.val, w.val), kw
        assert torch.equal(a.milli, w.milli)
    assert not ops.wide_ids(26) and ops.wide_ids(27)


@pytest.mark.parametrize("K", [1, 3, 40])
def test_topk_select_form_equals_sort_form_and_numpy(K):
    """qrlsh_topk_select_* (reverse words sorted on j, every edge ranks itself in its query's two runs) ==
    qrlsh_topk_* on all directed edge keys == a numpy lexsort, on pair lists with heavy value ties, very
    popular queries (thousands of neighbours) and queries without any"""
    rng = np.random.default_rng(7 + K)
    nq = 60000
    i = rng.integers(0, nq - 1, size=300000)
    j = rng.integers(0, nq, size=300000)
    hot = rng.integers(0, nq, size=20000)
    i = np.concatenate([i, np.full(20000, 777), hot[:9000], rng.integers(0, 50, size=5000)])      # 777 and 31000: popular
    j = np.concatenate([j, hot, np.full(9000, 31000), rng.integers(0, 50, size=5000)])
    keep = i != j
    lo, hi = np.minimum(i, j)[keep], np.maximum(i, j)[keep]
    pairs = np.unique((lo.astype(np.uint64) << np.uint64(32)) | hi.astype(np.uint64))
    n = len(pairs)
    milli = rng.integers(990, 1001, size=n).astype(np.int32)          # few distinct values: ties everywhere
    milli[rng.integers(0, n, size=n // 10)] = rng.integers(-1000, 1001, size=n // 10)
    ib = ops.id_bits_for(nq)
    pi, pj = (pairs >> np.uint64(32)).astype(np.int64), (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    src = np.concatenate([pi, pj]); dst = np.concatenate([pj, pi]); val = np.concatenate([milli, milli]).astype(np.int64)
    o = np.lexsort((dst, -val, src))
    src, dst, val = src[o], dst[o], val[o]
    first = np.r_[0, np.flatnonzero(src[1:] != src[:-1]) + 1]
    pos = np.arange(len(src)) - np.repeat(first, np.diff(np.r_[first, len(src)]))
    sel = pos < K
    dp, dm = dev(pairs.view(np.int64)), dev(milli)
    inv = (1000 - milli).astype(np.uint64)
    for wide in (False, True):
        if wide:
            rev = (dev(((pj.astype(np.uint64) << np.uint64(11)) | inv).view(np.int64)), dev(pi.astype(np.int32)))
        else:
            rev = dev(((pj.astype(np.uint64) << np.uint64(ib + 11)) | (inv << np.uint64(ib)) | pi.astype(np.uint64)).view(np.int64))
        s1, d1, v1 = (t.cpu().numpy() for t in ops.topk_select(dp, dm, rev, K, ib, nq))
        assert np.array_equal(s1, src[sel]) and np.array_equal(d1, dst[sel]) and np.array_equal(v1, val[sel])
    e = ops.pair_edges_interleaved(dp, dm, ib)
    s2, d2, v2 = (t.cpu().numpy() for t in ops.topk_edges(e, K, ib))
    assert np.array_equal(s2, src[sel]) and np.array_equal(d2, dst[sel]) and np.array_equal(v2, val[sel])
    z = ops.topk_select(dp[:0], dm[:0], dev(np.zeros(0, np.int64)), K, ib, nq)
    assert all(t.numel() == 0 for t in z)


# ---------------------------------------------------------------------------- N1 / N3 / N4
@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_compute_scores_dropin_matches_reference_on_generator_default_inputs(sub):
    """main.py's flow on the generator's CSVs (cfg1 / cfg1b: its defaults, two seeds; cfg2: 60 users x 150
    queries) through the drop-in Recommender: pandas ingest (N3), device answer sets (N2), hot path, user
    similarity (N4: scikit-learn clustering on the host, centred cosine + cut on the device), device prediction
    loop (N1) -> the reference's finalPredictions, cell for cell."""
    import pandas as pd
    import recommender as R
    g = load(sub + "_scores")
    gdir = os.path.join(GOLDEN, sub)
    rec = R.Recommender()
    rec.verbose = False
    dataset = pd.read_csv(os.path.join(gdir, "dataset.csv"))
    rec.datasetFeatures = list(dataset.columns)[1:]
    users = pd.read_csv(os.path.join(gdir, "users.csv"), header=None)
    queries, qids = rec.parse_queries(os.path.join(gdir, "queries.csv"))
    ratings = pd.read_csv(os.path.join(gdir, "utility_matrix.csv"))
    ratings.insert(0, "user", users[0].to_numpy())
    ratings.columns = ["user"] + qids                      # main.py:70: cols = ["user"] + queriesIDs
    rec.init(users, queries, qids, dataset, ratings)
    assert np.array_equal(rec.ratings, g["ratings"])
    R.PERM = int(g["P"])
    np.random.seed(int(g["seed"]))
    to_predict, final, missed = rec.compute_scores()
    assert list(final.columns) == qids and list(final.index) == users[0].tolist()
    assert np.array_equal(final.to_numpy(), g["final"])
    assert np.array_equal(to_predict, g["to_predict"]) and np.array_equal(missed, g["missed"])
    answers = iter(["abc", "3", "0", "2", "no"])
    rec.top_k_queries(to_predict, final, missed, ask=lambda prompt: next(answers))


@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_main_py_call_order_with_frame_inputs_reaches_golden_final(sub):
    """N3: main.py unchanged as the caller -- its call order (main.py:23-93) with Frame-like inputs (the
    datatable-Frame stand-in of tests/helpers.py: no pandas surface) through init -> compute_scores gives the
    reference's finalPredictions, for both orders of weighted_average's sums (the fixtures hold no cell on
    which they differ; the drop-in defaults to numba's sequential order)."""
    import recommender as R
    from helpers import main_py_inputs
    g = load(sub + "_scores")
    for order in ("sequential", "pairwise"):
        rec = R.Recommender()
        rec.verbose = False
        assert rec.sum_order == "sequential"
        rec.sum_order = order
        users, queries, qids, dataset, ratings = main_py_inputs(rec, os.path.join(GOLDEN, sub))
        rec.init(users, queries, qids, dataset, ratings)
        R.PERM = int(g["P"])
        np.random.seed(int(g["seed"]))
        to_predict, final, missed = rec.compute_scores()
        assert list(final.columns) == qids and list(final.index) == users.to_numpy().T[0].tolist()
        assert np.array_equal(final.to_numpy(), g["final"])
        assert np.array_equal(to_predict, g["to_predict"]) and np.array_equal(missed, g["missed"])
        csv_rows = final.to_numpy().tolist()            # main.py:103-107 (export) works on the returned frame
        csv_rows[0].insert(0, final.index.values[0])
        assert len(csv_rows[0]) == len(qids) + 1
    R.PERM = 180


def test_predict_c_entry_flags_a_too_long_list_without_walking_it():
    """qrlsh_predict called directly with kq below the real longest list (and a CSR list beyond 64)
    raises *too_long_out and never reads past the transposed workspace: the cells of such a query get 0."""
    import ctypes
    from qrlsh import _lib
    lib = _lib.load()
    nu, nq = 3, 300
    rng = np.random.RandomState(5)
    ratings = rng.randint(0, 5, size=(nu, nq)).astype(np.int32)
    deg = np.full(nq, 4, dtype=np.int64)
    deg[7] = 9          # longer than the kq the caller states
    deg[11] = 70        # longer than PRED_MAXK
    q_off = np.concatenate(([0], np.cumsum(deg))).astype(np.int64)
    q_idx = rng.randint(0, nq, size=int(q_off[-1])).astype(np.int32)
    q_val = np.round(rng.rand(int(q_off[-1])), 3)
    u_idx = np.full((nu, 1), -1, dtype=np.int32)
    u_val = np.zeros((nu, 1))
    d = [dev(x) for x in (ratings, q_off, q_idx, q_val, u_idx, u_val)]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    for kq in (4, 0):
        out = torch.full((nu, nq), -7, dtype=torch.int32, device=DEV)
        flag = torch.zeros((1,), dtype=torch.int32, device=DEV)
        ws = torch.zeros((max(16, int(lib.qrlsh_predict_workspace_bytes(nu, nq, kq))),), dtype=torch.uint8, device=DEV)
        rc = lib.qrlsh_predict(vp(d[0]), nu, nq, vp(d[1]), vp(d[2]), vp(d[3]), vp(d[4]), vp(d[5]), 0, 0.6, 0.4, 60.0,
                               _lib.SUM_PAIRWISE, vp(out), vp(flag), kq, vp(ws) if kq else None, ws.numel(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(flag.item()) == 1
        o = out.cpu().numpy()
        bad = [11] if kq == 0 else [7, 11]
        for j in bad:
            assert np.array_equal(o[:, j], np.where(ratings[:, j] != 0, ratings[:, j], 0))
        assert (o != -7).all()


def _check_user_sims_tie_aware(mine, ref, K):
    """mine / ref: {u: {'indexes', 'values'}}.  The reference keeps zero-valued entries (the user itself, negative
    cosines) when a cluster has fewer than K positive neighbours and orders ties arbitrarily (np.argsort); the
    device lists hold the positive entries only, ties by ascending id.  Equal means: same positive value
    multiset per user, every device neighbour strictly above the cut-off value is a reference neighbour and
    vice versa."""
    assert sorted(mine.keys()) == sorted(ref.keys())
    for u in ref:
        rv = np.asarray(ref[u]["values"], dtype=np.float64)
        ri = np.asarray(ref[u]["indexes"])
        pos = rv > 0
        mv, mi = np.asarray(mine[u]["values"]), np.asarray(mine[u]["indexes"])
        assert len(mv) <= K and np.all(mv > 0) and np.all(np.diff(mv) <= 0)
        assert np.array_equal(np.sort(mv), np.sort(rv[pos])), u
        if len(mv):
            cut = mv.min()
            assert set(mi[mv > cut].tolist()) == set(ri[pos & (rv > cut)].tolist()), u
        for a, b, x, y in zip(mi[:-1], mi[1:], mv[:-1], mv[1:]):
            assert x > y or a < b


@pytest.mark.parametrize("sub", GENERATOR_SETS)
def test_device_user_similarity_matches_reference(sub):
    """N4 (recommender.py:263-288) on the device -- integer-truncated centring, cosine of every pair of users of
    a cluster, per-user cut -- against the reference's own compute_userSimilarities output (golden) and the
    oracle restatement, with the cluster labels of the reference's scikit-learn call"""
    from qrlsh import users
    g = load(sub + "_scores")
    ratings = g["ratings"]
    nu = ratings.shape[0]
    K = users.max_candidates(nu)
    labels = O.user_cluster_labels(ratings)
    assert np.array_equal(labels, users.cluster_labels(ratings))
    src, dst, val = users.user_similarities(ratings, labels, K, DEV)
    mine = users.sims_to_dict(src, dst, val, nu)
    gold = {u: {"indexes": g["us_idx"][u][g["us_idx"][u] >= 0], "values": g["us_val"][u][g["us_idx"][u] >= 0]}
            for u in range(nu)}
    _check_user_sims_tie_aware(mine, gold, K)
    _check_user_sims_tie_aware(mine, O.user_similarities_from_labels(ratings, labels), K)
    # the drop-in's form: the device's scores cut by the reference's own numpy call -> the reference's lists
    # exactly, tie order and zero-valued entries included
    pairs, milli = users.cluster_pair_scores(ratings, labels, DEV)
    exact = users.reference_cut(pairs, milli, labels, K)
    for u in range(nu):
        assert np.array_equal(exact[u]["indexes"], gold[u]["indexes"]) and np.array_equal(exact[u]["values"], gold[u]["values"])
    # the centred rows themselves: truncation toward zero, zeros untouched
    c = users.center_rows(dev(ratings.astype(np.int32))).cpu().numpy()[:, :ratings.shape[1]]
    ref = ratings.astype(np.int64).copy()
    for r in ref:
        nz = r != 0
        if nz.any():
            r[nz] = r[nz] - np.mean(r[nz])
    assert np.array_equal(c, ref)


def test_device_pca_features_give_the_reference_clusters():
    """N4's clustering features on the device (recommender.py:226-234: StandardScaler + PCA): the column statistics and
    the Gram matrix of the standardized ratings (float64 MFMA, qrlsh_user_gram) against numpy / scikit-learn; the PCA
    scores U sqrt(lambda) against scikit-learn's transform up to the sign of a component; and what they are FOR -- the
    BIRCH labels -- equal to the reference's own scikit-learn call on the three generator sets, and, on a clustered
    2000 x 20 000 matrix (where scikit-learn itself takes its randomized solver), the same partition of the users."""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    from qrlsh import users
    rng = np.random.default_rng(8)
    # 1. statistics + Gram on an odd shape (tile edges, several column slices, constant columns)
    nu, nq = 333, 70_001
    r = rng.integers(0, 101, size=(nu, nq)).astype(np.int32)
    r[rng.random((nu, nq)) < 0.6] = 0
    r[:, 5] = 7                       # constant column: scale 1, standardized values 0
    r[:, 9] = 0
    sc = StandardScaler().fit(r)
    mean, inv, gram = users.standardized_gram(dev(r))
    assert np.array_equal(mean.cpu().numpy(), sc.mean_)
    assert np.allclose(1.0 / inv.cpu().numpy(), sc.scale_, rtol=1e-15, atol=0)
    z = sc.transform(r)
    g_ref = z @ z.T
    assert np.allclose(gram.cpu().numpy(), g_ref, rtol=1e-11, atol=1e-7)
    assert np.array_equal(gram.cpu().numpy(), gram.cpu().numpy().T)          # stored symmetric
    # 2. scores against scikit-learn's, component by component up to sign (leading, well separated components)
    for sub in GENERATOR_SETS:
        ratings = load(sub + "_scores")["ratings"]
        f_dev = users.pca_features(ratings, DEV).cpu().numpy()
        f_ref = users.pca_features_host(ratings)
        assert f_dev.shape == f_ref.shape
        lead = min(20, f_ref.shape[1] // 2)
        assert np.allclose(np.abs(f_dev[:, :lead]), np.abs(f_ref[:, :lead]), rtol=1e-6, atol=1e-8)
        # pairwise distances are what BIRCH sees: equal to ~1e-9
        d_dev = np.linalg.norm(f_dev[:, None, :] - f_dev[None, :, :], axis=2)
        d_ref = np.linalg.norm(f_ref[:, None, :] - f_ref[None, :, :], axis=2)
        assert np.allclose(d_dev, d_ref, rtol=1e-7, atol=1e-6)
        # 3. the labels
        want = O.user_cluster_labels(ratings)
        assert np.array_equal(users.cluster_labels(ratings, device=DEV), want)
    # 4. a clustered matrix at a size where scikit-learn's PCA takes its RANDOMIZED solver (random_state=None: the
    #    reference's own features differ from run to run there, so only a partition that the data defines is a target):
    #    as many groups of like-minded users as BIRCH is asked for clusters, every group recovered by both routes
    nu, nq = 2000, 20_000
    ng = round(nu ** (1 / 1.3))
    proto = rng.integers(1, 101, size=(ng, nq))
    grp = np.arange(nu) % ng
    r = proto[grp] + rng.integers(-3, 4, size=(nu, nq))
    r[rng.random((nu, nq)) < 0.5] = 0
    r = np.clip(r, 0, 100).astype(np.int32)
    truth = grp[:, None] == grp[None, :]
    lab_dev = users.cluster_labels(r, device=DEV)
    assert np.array_equal(lab_dev[:, None] == lab_dev[None, :], truth)
    lab_ref = users.cluster_labels(r)
    assert np.array_equal(lab_ref[:, None] == lab_ref[None, :], truth)


def test_device_user_similarity_on_random_clusters():
    """larger, synthetic: 3000 users in clusters of 1 .. 400, 257 queries (row stride padding), ratings with
    rows of a single distinct value (zero norm after centring) and empty rows"""
    from qrlsh import users
    rng = np.random.default_rng(44)
    nu, nq = 3000, 257
    ratings = (rng.integers(1, 101, size=(nu, nq)) * (rng.random((nu, nq)) < 0.3)).astype(np.int64)
    ratings[5] = 0
    ratings[6] = 0
    ratings[6, :40] = 77
    labels = rng.integers(0, 60, size=nu)
    labels[:400] = 1000
    labels[401] = 2000                                   # a cluster of one
    K = users.max_candidates(nu)
    src, dst, val = users.user_similarities(ratings, labels, K, DEV)
    _check_user_sims_tie_aware(users.sims_to_dict(src, dst, val, nu), O.user_similarities_from_labels(ratings, labels), K)


def _random_prediction_case(rng, nu, nq, Kq, Ku, fill):
    ratings = (rng.integers(1, 101, size=(nu, nq)) * (rng.random((nu, nq)) < fill)).astype(np.int64)
    qs, src, dst, mil = {}, [], [], []
    for j in range(nq):
        if rng.random() < 0.8:
            n = int(rng.integers(1, Kq + 1))
            idx = rng.choice(nq, size=min(n, nq), replace=False)
            v = np.sort(rng.integers(0, 1001, size=len(idx)))[::-1]
            qs[j] = {"indexes": idx.astype(np.int64), "values": v / 1000.0}
            src += [j] * len(idx); dst += idx.tolist(); mil += v.tolist()
    us = {}
    for u in range(nu):
        n = int(rng.integers(1, Ku + 1))
        idx = rng.choice(nu, size=min(n, nu), replace=False)
        us[u] = {"indexes": idx.astype(np.int64), "values": np.sort(rng.integers(0, 1001, size=len(idx)))[::-1] / 1000.0}
    coo = (torch.tensor(src, dtype=torch.int32), torch.tensor(dst, dtype=torch.int32), torch.tensor(mil, dtype=torch.int32))
    return ratings, qs, us, coo


def test_prediction_kernel_equals_oracle_on_random_inputs():
    """N1 (recommender.py:36-47, 301-331): the kernel against the oracle restatement in BOTH summation orders
    -- numpy's pairwise sum (the reference as plain Python: what the cfg1 fixtures pin) and the sequential one
    (numba's np.sum: the reference with numba installed, no fixture possible here) -- on lists up to the
    64-entry limit; longer lists are refused, not truncated."""
    from qrlsh import predict

    def seq_sum(a):
        r = 0.0
        for v in a:
            r += float(v)
        return r
    rng = np.random.default_rng(12)
    for (nu, nq, Kq, Ku, fill) in [(7, 9, 3, 2, 0.5), (40, 60, 17, 11, 0.3), (25, 30, 20, 12, 0.8), (5, 5, 1, 1, 0.0),
                                   (70, 90, 64, 64, 0.6), (33, 80, 41, 7, 0.9)]:
        ratings, qs, us, coo = _random_prediction_case(rng, nu, nq, Kq, Ku, fill)
        for tl in (True, False):     # query lists transposed to [longest][nq] (default) / read in their CSR form
            out = predict.fill_predictions(ratings, *coo, us, device=DEV, transpose_lists=tl)
            assert np.array_equal(out.cpu().numpy(), O.predict_scores(ratings, qs, us))
            out = predict.fill_predictions(ratings, *coo, us, device=DEV, sum_order="sequential", transpose_lists=tl)
            assert np.array_equal(out.cpu().numpy(), O.predict_scores(ratings, qs, us, summation=seq_sum))
    ratings, qs, us, coo = _random_prediction_case(rng, 80, 100, 80, 10, 0.5)
    assert max(len(v["indexes"]) for v in qs.values()) > 64
    for tl in (True, False):         # refused on the host / by the kernel's flag
        with pytest.raises(ValueError, match="more than 64 neighbours"):
            predict.fill_predictions(ratings, *coo, us, device=DEV, transpose_lists=tl)
    ratings, qs, us, coo = _random_prediction_case(rng, 80, 100, 10, 80, 0.5)
    with pytest.raises(ValueError, match="at most 64"):
        predict.fill_predictions(ratings, *coo, us, device=DEV)


def test_prediction_row_form_stages_the_row_in_lds_or_reads_it_from_memory():
    """the sweep's row form (one workgroup per user slice, the user's row staged in LDS as bytes) against the
    oracle and against the CSR-form kernel, on rows that span several workgroup strides and slices; ratings beyond
    255 (not the reference's domain, but legal int32 input) make a workgroup read its row from memory instead --
    in some rows only, in all rows -- with the same results"""
    from qrlsh import predict
    rng = np.random.default_rng(77)
    ratings, qs, us, coo = _random_prediction_case(rng, 12, 3000, 20, 6, 0.6)
    ref = O.predict_scores(ratings, qs, us)
    for variant in range(3):
        r = ratings.copy()
        if variant == 1:
            r[3, 17] = 256; r[7, 2999] = 100000; r[9, 0] = -4           # three rows fall back
        if variant == 2:
            r = r * 1000                                                # every row falls back
        want = ref if variant == 0 else O.predict_scores(r, qs, us)
        a = predict.fill_predictions(r, *coo, us, device=DEV, transpose_lists=True)
        b = predict.fill_predictions(r, *coo, us, device=DEV, transpose_lists=False)
        assert np.array_equal(a.cpu().numpy(), want) and torch.equal(a, b)
    # more users than one slice each, a row length that is no multiple of anything
    ratings, qs, us, coo = _random_prediction_case(rng, 150, 2501, 28, 19, 0.25)
    a = predict.fill_predictions(ratings, *coo, us, device=DEV, transpose_lists=True)
    b = predict.fill_predictions(ratings, *coo, us, device=DEV, transpose_lists=False)
    assert torch.equal(a, b)
    sub = O.predict_scores(ratings, qs, us)[:3]
    assert np.array_equal(a.cpu().numpy()[:3], sub)


def test_answer_sets_one_sweep_equals_count_then_fill():
    from qrlsh import answers
    rng = np.random.default_rng(31)
    for (D, card, two) in [(20000, 30, True), (3000, 4, False), (64 * 32, 32, False)]:
        cols = [rng.integers(0, card, size=D).astype(str) for _ in range(3)]
        nq = 4000
        q = np.full((nq, 3), "", dtype=object)
        q[:, 0] = rng.integers(0, card, size=nq).astype(str)
        if two:
            q[:, 2] = rng.integers(0, card, size=nq).astype(str)
        idx = answers.build_answer_index(cols, DEV)
        qr = answers.encode_queries(idx, q)
        a_off, a_rows = answers.answer_sets(idx, qr, one_sweep=True)
        b_off, b_rows = answers.answer_sets(idx, qr, one_sweep=False)
        assert torch.equal(a_off, b_off) and torch.equal(a_rows, b_rows)
        sizes = np.diff(a_off.cpu().numpy())
        assert (sizes.max() <= 64) == two or not two     # first case exercises the compact path, the others the refill
        roff, rrows = O.answer_sets(cols, q[:300])
        assert np.array_equal(a_off.cpu().numpy()[:301], roff) and np.array_equal(a_rows.cpu().numpy()[:roff[-1]], rrows)
