"""CPU tests of the host layer: the C-ABI library loads and exports every symbol the header
declares (no compute calls -- there is no GPU here), and the pure-host logic around it."""
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "qrlsh.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qrlsh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from qrlsh import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libqrlsh.so does not export %s" % n
    # and the ctypes table covers exactly the header
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_library_host_only_entry_points():
    from qrlsh import _lib
    lib = _lib.load()
    assert lib.qrlsh_version() >= 1
    assert lib.qrlsh_sort_workspace_bytes(4096 * 3, 2) == 2 * 256 * (3 + 1) * 4
    assert lib.qrlsh_pairs_workspace_bytes(1024 * 5 + 1, 3) == 3 * 6 * 8
    assert lib.qrlsh_compact_workspace_bytes(2048 * 2) == (2 + 1 + 1) * 8    # tile counts + scan chunk totals
    # the device mixer is a bijection: spot-check injectivity and the known splitmix64 vector
    vals = {lib.qrlsh_mix64_host(i) for i in range(10000)}
    assert len(vals) == 10000


def test_argument_errors_do_not_need_a_gpu():
    from qrlsh import _lib
    lib = _lib.load()
    rc = lib.qrlsh_band_keys(None, 10, 12, 5, None, None, None)       # P % b != 0 -> EINVAL (lsh.py:20)
    assert rc == _lib.QRLSH_EINVAL
    assert b"divisible" in lib.qrlsh_last_error() or b"bad arguments" in lib.qrlsh_last_error()
    with pytest.raises(_lib.QrlshError):
        _lib.check(rc)
    rc = lib.qrlsh_sort_u64(None, None, None, None, -1, 1, 0, 8, 0, 0, None, 0, None)
    assert rc == _lib.QRLSH_EINVAL
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.QRLSH_EUNSUPPORTED)


def test_host_helpers_match_reference_rules():
    from qrlsh import ops, pipeline
    from oracle import oracle as O
    for P, b in [(180, 60), (160, 40), (200, 50), (320, 80)]:
        assert pipeline.select_bands(P) == b == O.select_bands(P)
    for P in (128, 256):
        with pytest.raises(ValueError):
            pipeline.select_bands(P)
    for nq, K in [(100, 11), (1_000_000, 34), (10_000_000, 40), (100_000_000, 45)]:
        assert pipeline.max_candidates(nq) == K == round(math.log(nq, 1.5))
    assert [ops.id_bits_for(n) for n in (1, 2, 3, 1024, 1025, 1_000_000)] == [1, 1, 2, 10, 11, 20]
    assert ops.hash_bits_for(1_000_000) == 24 and ops.hash_bits_for(10_000_000) == 32 and ops.hash_bits_for(10) == 8


def test_host_size_rules():
    """the host-side rules that pick kernel variants (no GPU needed)"""
    from qrlsh import _lib, ops
    lib = _lib.load()
    # id bits / partition depth / hash bits
    assert [ops.id_bits_for(n) for n in (1, 2, 3, 1 << 20, (1 << 20) + 1)] == [1, 1, 2, 20, 21]
    assert ops.part_bits_for(1_000_000) == 8 and ops.part_bits_for(1_200_000) == 9 and ops.part_bits_for(10_000_000) == 12
    assert ops.part_bits_for(10 ** 9) == 16
    # rows of 2^g consecutive i: g = id bits above a multiple of 8 when <= 4, and only for short rows
    assert [ops.row_group_bits(b) for b in (8, 9, 12, 13, 16, 17, 20, 21, 23, 24)] == [0, 1, 4, 0, 0, 1, 4, 0, 0, 0]
    assert ops.row_group_bits(20, 17.7) == 4 and ops.row_group_bits(20, 64.0) == 4 and ops.row_group_bits(20, 65.0) == 0
    # wide ids: two ids + 11 score bits must fit 64 bits for the packed edge key
    assert not ops.wide_ids(26) and ops.wide_ids(27)
    # scratch sizes grow with the problem and the one-kernel partition needs room for its fixed regions
    # regions of ONE LDS image of the finish + the overflow pool (1/16 of the records, at least 1 M)
    assert lib.qrlsh_bucket_part_words(1_000_000, 32, 8) == 32 * 256 * 6144 + 32 * 1_000_000 // 16
    assert lib.qrlsh_bucket_part_words(1000, 4, 8) >= 4 * 1000
    assert lib.qrlsh_bucket_tmp_words(1_000_000, 32, 8) == 0
    assert lib.qrlsh_bucket_tmp_words(10_000_000, 32, 12) > 32 * 10_000_000
    assert lib.qrlsh_bucket_part_words(10_000_000, 32, 12) == 32 * 4096 * 4096 + 32 * 10_000_000 // 16   # small-part image
    assert lib.qrlsh_bucket_part_words(10_000_000, 32, 12) < 2 * 32 * 10_000_000      # reserved < 2 x the records
    assert lib.qrlsh_bucket_part_words(100_000_000, 8, 15) == 8 * 32768 * 6144 + 8 * 100_000_000 // 16
    assert lib.qrlsh_bucket_part_words(100_000_000, 8, 15) < 2.1 * 8 * 100_000_000
    assert lib.qrlsh_bucket_part_words(1 << 25, 2, 9) == 2 * (1 << 25)          # ids past 24 bits: sort-based layout
    # fixed regions of the histogram-free pair grouping: 10 M ids in regions of 256 queries = 39 063 regions, dealt by two
    # levels of 8 bits (153 coarse digits x 256); capacity 3 x the mean + 4096, rounded to 64 words
    n, nids = 190_012_232, 10_000_000
    assert lib.qrlsh_pair_regions_count(n, nids, 8, 0.0) == 153 * 256
    cap = lib.qrlsh_pair_regions_cap(n, nids, 8, 0.0)
    assert cap % 64 == 0 and 3 * (n // 39063) + 4096 <= cap < 3 * (n // 39063) + 4096 + 128
    assert lib.qrlsh_pair_regions_words(n, nids, 8, 0.0) == 153 * 256 * cap
    assert lib.qrlsh_pair_regions_tmp_words(n, nids, 8, 0.0) > 2 * n            # 153 coarse regions of 2.5 x the mean
    assert lib.qrlsh_pair_regions_cap(n, nids, 8, 100.0) > cap                    # a caller's density hint can only enlarge them
    assert lib.qrlsh_pair_regions_tmp_words(1000, 200, 7, 0.0) == 0               # up to 256 regions: one level
    assert lib.qrlsh_pair_regions_count(1000, 200, 7, 0.0) == 2
    assert lib.qrlsh_pair_regions_words(10 ** 9, 100_000_000, 5, 0.0) == 0        # more than 65536 regions: not served
    assert lib.qrlsh_set_big_part_limit(0) == 16 * 6144 and lib.qrlsh_set_big_part_limit(5000) == 16 * 6144
    assert lib.qrlsh_set_big_part_limit(10 ** 9) == 5000 and lib.qrlsh_set_big_part_limit(0) == 16 * 6144
    assert lib.qrlsh_user_gram_workspace_bytes(2000, 100_000) == 16 * 2000 * 2000 * 8
    assert lib.qrlsh_row_unique_workspace_bytes(0) > 0
    assert lib.qrlsh_row_unique_workspace_bytes(10 ** 8) > lib.qrlsh_row_unique_workspace_bytes(10 ** 6)


def test_legacy_permutations_follow_the_global_stream():
    from qrlsh import ops
    from oracle import oracle as O
    a = ops.legacy_permutations(5, 100, seed=7)
    np.random.seed(7)
    b = ops.legacy_permutations(5, 100)                  # consumes np.random like recommender.py:120
    np.random.seed(7)
    c = np.stack([np.random.permutation(100) for _ in range(5)])
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, O.legacy_permutations(7, 5, 100))


def test_perm_table_layout():
    from qrlsh import ops, _lib
    perms = ops.legacy_permutations(13, 300, seed=1)
    t = ops.perm_table(perms, device="cpu")
    assert t.code == _lib.PERM_U16 and t.P_stride == 16 and tuple(t.tab.shape) == (300, 16)
    assert np.array_equal(t.tab.numpy().view(np.uint16)[:, :13], perms.T.astype(np.uint16))
    t2 = ops.perm_table(ops.legacy_permutations(6, 70000, seed=1), device="cpu")
    assert t2.code == _lib.PERM_I32 and t2.P_stride == 8 and t2.tab.dtype.itemsize == 4


def test_sims_to_dict_shape_of_reference_return_value():
    from qrlsh import pipeline
    src = np.array([2, 2, 2, 5], dtype=np.int32)
    dst = np.array([9, 4, 7, 2], dtype=np.int32)
    val = np.array([1000, 875, 875, 12], dtype=np.int32)
    d = pipeline.sims_to_dict(src, dst, val)
    assert sorted(d) == [2, 5]
    assert d[2]["indexes"].dtype == np.int64 and d[2]["values"].dtype == np.float64
    assert d[2]["indexes"].tolist() == [9, 4, 7] and d[2]["values"].tolist() == [1.0, 0.875, 0.875]
    assert pipeline.sims_to_dict(src[:0], dst[:0], val[:0]) == {}


def test_no_product_module_imports_the_oracle():
    """the oracle is test infrastructure: nothing under the product package may import it,
    load its library or add its directory to the path"""
    pkg = os.path.join(ROOT, "query-recommendation-system_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|libqroracle|CDLL\([^)]*oracle", re.M)
    checked = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")) or f == "Makefile":
                checked += 1
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
    assert checked >= 10


def test_generator_write_csv_shim(tmp_path, monkeypatch):
    """main.py:2,109 uses `from resources import generator; generator.write_csv(...)`"""
    from resources import generator
    monkeypatch.chdir(tmp_path)
    generator.write_csv("final_predictions", ["Q1", "Q2"], [["U1", 3, 4], ["U2", 0, 7]])
    generator.write_csv("noheader", None, [[1, 2]])
    assert open(tmp_path / "output" / "final_predictions.csv").read().splitlines() == ["Q1,Q2", "U1,3,4", "U2,0,7"]
    assert open(tmp_path / "output" / "noheader.csv").read().splitlines() == ["1,2"]


@pytest.mark.parametrize("sub", ["cfg1", "cfg1b", "cfg2"])
def test_init_accepts_what_main_py_passes(sub):
    """N3: main.py:23-85 hands datatable Frames to Recommender.init (recommender.py:51-64).  A Frame-like stand-in
    (tests/helpers.py: .names / .shape / .to_numpy() / .to_pandas() only) and plain pandas frames must give the
    same state, equal to what the reference's own init leaves behind (the golden `ratings`)."""
    import pandas as pd
    import recommender as R
    from helpers import GOLDEN, load, main_py_inputs
    g = load(sub + "_scores")
    gdir = os.path.join(GOLDEN, sub)
    rec = R.Recommender()
    users, queries, qids, dataset, ratings = main_py_inputs(rec, gdir)
    assert not hasattr(dataset, "columns") and not hasattr(ratings, "astype")   # really not pandas
    rec.init(users, queries, qids, dataset, ratings)
    assert np.array_equal(rec.ratings, g["ratings"]) and rec.ratings.dtype == np.int64
    assert rec.usersIDs.tolist() == pd.read_csv(os.path.join(gdir, "users.csv"), header=None)[0].tolist()
    assert list(rec.queriesIDs) == qids and rec.queries.shape == (len(qids), len(rec.datasetFeatures))
    assert all(isinstance(v, str) for v in rec.dataset.iloc[0].tolist())          # dataset[:] = dt.str64
    assert rec.dataset["age"].iloc[0] == str(pd.read_csv(os.path.join(gdir, "dataset.csv"))["age"].iloc[0])
    # the same through pandas frames
    rec2 = R.Recommender()
    rec2.datasetFeatures = rec.datasetFeatures
    rec2.init(users.to_pandas(), queries, qids, dataset.to_pandas(), ratings.to_pandas())
    assert np.array_equal(rec2.ratings, rec.ratings) and rec2.dataset.equals(rec.dataset)
    assert np.array_equal(rec2.usersIDs, rec.usersIDs) and np.array_equal(rec2.queries, rec.queries)
