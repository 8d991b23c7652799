#!/usr/bin/env python3
"""Same-box A/B of the two scoring kernels (generic form / run form) on the candidate pairs of the standard workload:
results compared bit for bit, each form timed with events.  python tools/score_ab.py [--nq N] [--perm P --bands B]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import qrlsh  # noqa: E402
from qrlsh import ops, pipeline, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=10_000_000)
    ap.add_argument("--perm", type=int, default=128)
    ap.add_argument("--bands", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    lib = _lib.load()
    D = 32768
    off, rows = qrlsh.synth_csr(a.nq, D, seed=0, device=dev)
    table = ops.perm_table(ops.legacy_permutations(a.perm, D, seed=42), dev)
    K = pipeline.max_candidates(a.nq)
    res = pipeline.query_similarities(off, rows, table, a.bands, K)
    sig, norm2, pairs = res.sig, res.norm2, res.pairs
    ib = ops.id_bits_for(a.nq)
    n = pairs.numel()
    print("nq=%d P=%d pairs=%d" % (a.nq, a.perm, n))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    out = {}
    for form in (0, 1, 0, 1):
        lib.qrlsh_set_score_runs(form)
        m, rev = ops.score_pairs_rev(sig, norm2, pairs, ib)
        m2, _, edges = ops.score_pairs(sig, norm2, pairs[: n // 7], edge_id_bits=ib, wide=True)
        half = a.nq // 2    # two-piece table: rows below `half` in the first piece
        m3 = ops.score_pairs_split(sig[:half], norm2[:half], sig[half:], norm2[half:], pairs)
        ms = timed(lambda: ops.score_pairs_rev(sig, norm2, pairs, ib))
        ms_split = timed(lambda: ops.score_pairs_split(sig[:half], norm2[:half], sig[half:], norm2[half:], pairs))
        print("form %d (%s): score_pairs_rev %.3f ms = %.2f TB/s algorithmic (%d B per pair); split-table form %.3f ms"
              % (form, "runs" if form else "generic", ms, n * (4 * a.perm + 20) / ms / 1e9, 4 * a.perm + 20, ms_split), flush=True)
        cur = (m, rev if not isinstance(rev, tuple) else rev[0], m2, edges[0], edges[1], m3)
        if form in out:
            assert all(torch.equal(x, y) for x, y in zip(cur, out[form]))
        out[form] = cur
    assert all(torch.equal(x, y) for x, y in zip(out[0], out[1])), "the two forms differ"
    assert torch.equal(out[1][0], out[1][5]) and torch.equal(out[1][0], res.milli)
    print("forms agree bit for bit (scores, reverse words, wide edges, split-table scores)")


if __name__ == "__main__":
    main()
