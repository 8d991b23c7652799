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


def _run(tmp_path, world, nq, D, P, b, mode, port, extra=()):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(tmp_path), str(nq), str(D), str(P), str(b), mode] + list(extra)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,mode,b,extra", [(2, "all_to_all", 8, ()), (2, "all_gather", 8, ("sig=fetch",)),
                                               (3, "all_to_all", 8, ("sig=fetch",)),
                                               (3, "all_to_all", 8, ("sig=all_gather",)),
                                               (2, "all_to_all", 4, ("wide", "sig=fetch")),
                                               (2, "all_to_all", 4, ("wide", "sig=all_gather"))])
def test_sharded_equals_single_process(tmp_path, world, mode, b, extra):
    # b = 4 with P = 32 is a wide band (r = 8: hashed bucket ids + verification); "wide" forces the
    # key + payload edge format used when two ids + 11 score bits do not fit 64 bits
    nq, D, P = 600, 512, 32
    port = 29531 + world + (0 if mode == "all_to_all" else 7) + len(extra) * 11 + sum(map(len, extra))
    outs = _run(tmp_path, world, nq, D, P, b, mode, port, extra)
    for o in outs:     # the signature exchange that was asked for is the one that ran ("auto": either)
        want = [e[4:] for e in extra if e.startswith("sig=")]
        assert not want or str(o["sig_exchange"]) == want[0]
        assert (str(o["sig_exchange"]) == "fetch") == (int(o["fetched"]) >= 0)
    K = O.max_candidates(nq)
    off, rows = O.synth_csr(nq, D, seed=3, cluster=4, mean=6.0)
    ref = O.query_similarities(off, rows, D, P, b, K, 42)
    assert np.array_equal(np.concatenate([o["sig"] for o in outs]), ref["sig"])
    pairs = np.concatenate([o["pairs"] for o in outs]).view(np.uint64)
    assert np.array_equal(pairs, ref["pairs"])                     # rank order == global order, no duplicates
    assert np.array_equal(np.concatenate([o["milli"] for o in outs]), ref["milli"])
    assert np.array_equal(np.concatenate([o["src"] for o in outs]), ref["src"])
    assert np.array_equal(np.concatenate([o["dst"] for o in outs]), ref["dst"])
    assert np.array_equal(np.concatenate([o["val"] for o in outs]), ref["val"])
    if P // b <= 4:
        keys = O.band_keys(ref["sig"], b)
        assert sum(int(o["emitted"]) for o in outs) == O.emitted_pairs(keys, P // b)
    nql = nq // world
    for r, o in enumerate(outs):                                   # ownership: i (and src) in the rank's range
        i = o["pairs"].view(np.uint64) >> np.uint64(32)
        assert len(i) == 0 or (i.min() >= r * nql and i.max() < (r + 1) * nql)
        assert len(o["src"]) == 0 or (o["src"].min() >= r * nql and o["src"].max() < (r + 1) * nql)


def test_band_owner_ranges_cover_all_bands():
    import qrlsh.dist as qd
    for b in (1, 5, 8, 32, 60, 64):
        for w in (1, 2, 3, 4, 8):
            rg = qd.band_owner_ranges(b, w)
            assert len(rg) == w and rg[0][0] == 0 and rg[-1][1] == b
            assert all(rg[i][1] == rg[i + 1][0] for i in range(w - 1))
