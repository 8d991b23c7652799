#!/usr/bin/env python3
"""Bit-exact GPU-vs-oracle sweep over shapes the unit tests do not reach (development tool, run on the
GPU box): partition depths T = 8 .. 12, answer-set sizes from near-empty to large, cluster sizes that
make hot rows, wide tables, P/b combinations.  usage: python tools/sweep_check.py [--quick]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import qrlsh  # noqa: E402
from qrlsh import ops, pipeline  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)

CASES = [
    # nq,       D,      P,   b,  cluster, mean, p_replace
    (300_000, 32768, 128, 32, 8, 16.0, 0.15),
    (1_500_000, 32768, 128, 32, 8, 16.0, 0.15),      # T = 9: two-step partition
    (3_000_000, 32768, 128, 32, 8, 16.0, 0.15),      # T = 10
    (1_000_000, 32768, 128, 32, 64, 16.0, 0.05),     # big clusters: buckets of 64, long rows
    (1_000_000, 32768, 128, 32, 8, 1.5, 0.15),       # tiny answer sets, many empty
    (400_000, 32768, 128, 32, 8, 64.0, 0.15),        # large answer sets
    (1_000_000, 32768, 256, 64, 8, 16.0, 0.15),      # BASELINE shape 256/64
    (700_000, 100_000, 128, 32, 8, 16.0, 0.15),      # int32 table / signatures
    (500_000, 32768, 100, 20, 8, 16.0, 0.15),        # wide bands r = 5 (hashed ids + verification)
    (200_000, 32768, 96, 32, 8, 16.0, 0.15),         # r = 3
    (6_000_000, 32768, 128, 32, 64, 16.0, 0.05),     # T = 11: small-part finish (separate counters), buckets of 64, spills
    (12_000_000, 32768, 128, 32, 8, 16.0, 0.15),     # T = 12: packed-counter finish, scattered pair regions, run-form scoring
    (9_000_000, 32768, 256, 64, 16, 12.0, 0.10),     # the same with 256-value rows
]


def main():
    quick = "--quick" in sys.argv
    O.set_threads(int(os.environ.get('QRLSH_TEST_THREADS', '16')))
    bad = 0
    for (nq, D, P, b, cl, mean, pr) in (CASES[:3] if quick else CASES):
        K = pipeline.max_candidates(nq)
        off, rows = qrlsh.synth_csr(nq, D, seed=1, cluster=cl, mean=mean, p_replace=pr, device="cuda")
        perms = ops.legacy_permutations(P, D, seed=7)
        t0 = time.perf_counter()
        res = pipeline.query_similarities(off, rows, ops.perm_table(perms, "cuda"), b, K)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        ho, hr = off.cpu().numpy(), rows.cpu().numpy()
        sig = O.minhash(ho, hr, perms)
        if P // b <= 4:
            opairs = O.candidates(O.band_keys(sig, b), P // b)
        else:
            opairs = O.candidates_from_sig(sig, b)
        milli = O.score_pairs(sig, opairs, mode=1)
        s, d, v = O.topk(opairs, milli, K)
        tc = time.perf_counter() - t0
        ok = (np.array_equal(res.sig_int32().cpu().numpy(), sig)
              and np.array_equal(res.pairs.cpu().numpy().view(np.uint64), opairs)
              and np.array_equal(res.milli.cpu().numpy(), milli)
              and np.array_equal(res.src.cpu().numpy(), s) and np.array_equal(res.dst.cpu().numpy(), d)
              and np.array_equal(res.val.cpu().numpy(), v))
        bad += not ok
        print("%s nq=%d D=%d P=%d b=%d cluster=%d mean=%.1f: emitted=%d unique=%d kept=%d  bucket=%s dedup=%s  "
              "gpu(first call) %.1f ms, oracle %.1f s" % ("OK  " if ok else "FAIL", nq, D, P, b, cl, mean,
              res.stats["emitted_pairs"], res.pairs.numel(), res.src.numel(), res.stats.get("bucket_path"),
              res.stats.get("dedup_path"), tg * 1e3, tc), flush=True)
        del res, off, rows
        torch.cuda.empty_cache()
    print("sweep: %d failing case(s)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
