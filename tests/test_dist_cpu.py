"""world_size > 1 on CPU (gloo): the query-sharded driver's host logic -- shard ranges, band
ownership, bucket-id exchange (both modes), pair / reverse-edge exchange -- must reproduce the
single-process result exactly.  Compute is the oracle (test-only backend); the HIP kernels
themselves are covered by the -m gpu tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _run(tmp_path, world, nq, D, P, b, mode, port, extra=()):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(tmp_path), str(nq), str(D), str(P), str(b), mode] + list(extra)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,mode,b,extra,nq", [(2, "all_to_all", 8, (), 600), (2, "all_gather", 8, ("sig=fetch",), 600),
                                                  (3, "all_to_all", 8, ("sig=fetch",), 600),
                                                  (3, "all_to_all", 8, ("sig=all_gather",), 601),   # 601 = 3 * 201 - 2:
                                                  (3, "all_to_all", 8, ("sig=fetch",), 601),        # padded last shard
                                                  (4, "all_gather", 8, ("sig=fetch",), 598),
                                                  (2, "all_to_all", 8, ("sig=recompute",), 600),    # answer sets replicated,
                                                  (3, "all_to_all", 8, ("sig=recompute",), 601),    # no bucket-id / row exchange
                                                  (4, "all_to_all", 8, (), 598),                    # (auto up to 4 ranks)
                                                  (2, "all_to_all", 4, ("wide", "sig=recompute"), 599),
                                                  (3, "all_to_all", 8, ("sig=sets",), 601),         # replicated sets, signatures on demand
                                                  (5, "all_gather", 8, (), 603),                    # (auto from five ranks)
                                                  (8, "all_to_all", 32, ("sig=sets",), 1001),
                                                  (8, "all_to_all", 8, ("sig=fetch",), 603),        # the driver's N = 8:
                                                  (8, "all_to_all", 32, ("sig=fetch",), 1001),      # one band / four bands per rank
                                                  (2, "all_to_all", 4, ("wide", "sig=fetch"), 600),
                                                  (2, "all_to_all", 4, ("wide", "sig=all_gather"), 599),
                                                  # ONE rank made to take every exchange step (force_collectives: what
                                                  # the -m gpu RCCL rehearsal runs with nccl on the one-GPU box)
                                                  (1, "all_to_all", 8, ("force", "sig=sets"), 600),
                                                  (1, "all_gather", 8, ("force", "sig=fetch"), 600),
                                                  (1, "all_to_all", 8, ("force", "sig=recompute"), 600),
                                                  (1, "all_to_all", 4, ("force", "wide", "sig=all_gather"), 600)])
def test_sharded_equals_single_process(tmp_path, world, mode, b, extra, nq):
    # b = 4 with P = 32 is a wide band (r = 8: hashed bucket ids + verification); "wide" forces the
    # key + payload edge format used when two ids + 11 score bits do not fit 64 bits
    D, P = 512, 32
    port = 29531 + world + (0 if mode == "all_to_all" else 7) + len(extra) * 11 + sum(map(len, extra)) + nq % 7 * 13
    outs = _run(tmp_path, world, nq, D, P, b, mode, port, extra)
    for o in outs:     # the signature exchange that was asked for is the one that ran ("auto": either)
        want = [e[4:] for e in extra if e.startswith("sig=")]
        assert not want or str(o["sig_exchange"]) == want[0]
        assert (str(o["sig_exchange"]) == "fetch") == (int(o["fetched"]) >= 0)
    K = O.max_candidates(nq)
    off, rows = O.synth_csr(nq, D, seed=3, cluster=4, mean=6.0)
    ref = O.query_similarities(off, rows, D, P, b, K, 42)
    assert np.array_equal(np.concatenate([o["sig"] for o in outs]), ref["sig"])       # pad rows are not returned
    # candidate pairs: every rank holds the sorted share it scored; the shares are disjoint and their union is
    # the single-process list
    pairs = np.concatenate([o["pairs"] for o in outs]).view(np.uint64)
    milli = np.concatenate([o["milli"] for o in outs])
    order = np.argsort(pairs, kind="stable")
    assert np.array_equal(pairs[order], ref["pairs"])                # nothing missing, nothing twice
    assert np.array_equal(milli[order], ref["milli"])
    assert np.array_equal(np.concatenate([o["src"] for o in outs]), ref["src"])
    assert np.array_equal(np.concatenate([o["dst"] for o in outs]), ref["dst"])
    assert np.array_equal(np.concatenate([o["val"] for o in outs]), ref["val"])
    if P // b <= 4:
        keys = O.band_keys(ref["sig"], b)
        assert sum(int(o["emitted"]) for o in outs) == O.emitted_pairs(keys, P // b)
    nql = -(-nq // world)
    from dist_worker import pair_host
    sizes = []
    for r, o in enumerate(outs):
        p = o["pairs"].view(np.uint64)
        assert np.all(p[1:] > p[:-1])                                # sorted unique share
        assert np.all(pair_host(p, nql) == r)                        # scored where qr_pair_host says
        sizes.append(len(p))
        assert len(o["src"]) == 0 or (o["src"].min() >= r * nql and o["src"].max() < min((r + 1) * nql, nq))
    assert max(sizes) <= 2.0 * sum(sizes) / world + 16               # the coin splits the pairs about evenly


@pytest.mark.parametrize("world,extra", [(2, ("sig=recompute",)), (3, ("sig=sets",))])
def test_replicated_answer_sets_with_more_than_65536_table_rows(tmp_path, world, extra):
    """the answer-set gather of the recompute / sets modes sends 16-bit row ids only while the table has at most 65536
    rows; beyond (here D = 70 000: int32 permutation table, int32 signature rows, int16 wrap in the band keys) the
    ids travel as 32-bit words -- same results as one process"""
    nq, D, P, b = 300, 70000, 32, 8
    outs = _run(tmp_path, world, nq, D, P, b, "all_to_all", 29700 + world, extra)
    K = O.max_candidates(nq)
    off, rows = O.synth_csr(nq, D, seed=3, cluster=4, mean=6.0)
    assert rows.max() > 65535
    ref = O.query_similarities(off, rows, D, P, b, K, 42)
    assert np.array_equal(np.concatenate([o["sig"] for o in outs]), ref["sig"])
    pairs = np.concatenate([o["pairs"] for o in outs]).view(np.uint64)
    order = np.argsort(pairs, kind="stable")
    assert np.array_equal(pairs[order], ref["pairs"])
    assert np.array_equal(np.concatenate([o["milli"] for o in outs])[order], ref["milli"])
    for k in ("src", "dst", "val"):
        assert np.array_equal(np.concatenate([o[k] for o in outs]), ref[k])


def test_shard_ranges_tile_the_queries():
    import qrlsh.dist as qd
    for nq in (1, 7, 600, 601, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            nxt = 0
            for g in range(w):
                q0, n, nql = qd.shard_range(nq, w, g)
                assert nql == -(-nq // w) and 0 <= n <= nql
                assert q0 == nxt or n == 0
                nxt = q0 + n
            assert nxt == nq


def test_band_owner_ranges_cover_all_bands():
    import qrlsh.dist as qd
    for b in (1, 5, 8, 32, 60, 64):
        for w in (1, 2, 3, 4, 8):
            rg = qd.band_owner_ranges(b, w)
            assert len(rg) == w and rg[0][0] == 0 and rg[-1][1] == b
            assert all(rg[i][1] == rg[i + 1][0] for i in range(w - 1))
