#!/usr/bin/env python3
"""Run the hot path once at a large size (default BASELINE config 3: 10 M queries x 128/32 on one
MI355X), print per-phase and per-kernel times, check size-independent properties, and
(optionally) compare exactly with the CPU oracle.  usage: big_run.py [--nq N] [--oracle]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import qrlsh  # noqa: E402
from qrlsh import ops, pipeline, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=10_000_000)
    ap.add_argument("--perm", type=int, default=128)
    ap.add_argument("--bands", type=int, default=32)
    ap.add_argument("--drows", type=int, default=32768)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = "cuda"
    nq, P, b, D = a.nq, a.perm, a.bands, a.drows
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=dev)
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, dev)
    res = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        tm = {}
        res = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    rep = _lib.prof_report()
    _lib.prof_enable(False)
    print("nq=%d P=%d b=%d K=%d nnz=%d  step %.3f ms = %.1f M signatures/s; emitted=%d unique=%d kept=%d path=%s"
          % (nq, P, b, K, rows.numel(), dt * 1e3, nq / dt / 1e6, res.stats["emitted_pairs"], res.pairs.numel(),
             res.src.numel(), res.stats.get("bucket_path")))
    for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        print("  %-18s launches/run %5.1f  ms/run %8.4f" % (k, c / a.reps, ms / a.reps))
    print("  peak device memory %.2f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
    pairs = res.pairs.cpu().numpy().view(np.uint64)
    i, j = (pairs >> np.uint64(32)).astype(np.int64), (pairs & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all(i < j) and j.max() < nq and np.all(pairs[1:] > pairs[:-1])
    src, dst, val = res.src.cpu().numpy(), res.dst.cpu().numpy(), res.val.cpu().numpy()
    ib = ops.id_bits_for(nq)
    key = (src.astype(np.int64) << (ib + 11)) | ((1000 - val).astype(np.int64) << ib) | dst
    assert np.all(key[1:] > key[:-1]) and np.bincount(src).max() <= K
    print("  properties ok (pairs sorted unique i<j; top-K ordered, <= K per query)")
    if a.oracle:
        from oracle import oracle as O
        O.set_threads(16)
        t0 = time.perf_counter()
        sig = O.minhash(off.cpu().numpy(), rows.cpu().numpy(), perms)
        opairs = O.candidates(O.band_keys(sig, b), P // b)
        milli = O.score_pairs(sig, opairs, mode=1)
        s, d, v = O.topk(opairs, milli, K)
        print("  oracle (16 threads) %.2f s" % (time.perf_counter() - t0))
        assert np.array_equal(res.sig_int32().cpu().numpy(), sig)
        assert np.array_equal(pairs, opairs)
        assert np.array_equal(res.milli.cpu().numpy(), milli)
        assert np.array_equal(src, s) and np.array_equal(dst, d) and np.array_equal(val, v)
        print("  bit-exact vs oracle: signatures, pairs, scores, top-K")


if __name__ == "__main__":
    main()
