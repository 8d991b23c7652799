#!/usr/bin/env python3
"""Development probe: what would a similarity-aware query order be worth?  Runs the hot path on the synthetic workload
as generated (cluster-mates nq/8 ids apart) and with the queries renumbered so that cluster-mates are neighbours
(oracle knowledge of the generator), and prints the per-kernel times of both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import qrlsh  # noqa: E402
from qrlsh import ops, pipeline, _lib  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = 32768
off, rows = qrlsh.synth_csr(nq, D, seed=0, device="cuda")
table = ops.perm_table(ops.legacy_permutations(128, D, seed=42), "cuda")
K = pipeline.max_candidates(nq)
nb = nq // 8
new = torch.arange(nq, device="cuda")
old_of_new = (new % 8) * nb + new // 8                  # position p holds old query (p % 8) * nb + p // 8
lens = (off[1:] - off[:-1])[old_of_new]
off2 = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
torch.cumsum(lens, 0, out=off2[1:])
idx = torch.repeat_interleave(off[:-1][old_of_new] - off2[:-1], lens) + torch.arange(int(off2[-1]), device="cuda")
rows2 = rows[idx].contiguous()
for name, (o, r) in (("generated order", (off, rows)), ("cluster-mates adjacent", (off2, rows2))):
    for _ in range(3):
        res = pipeline.query_similarities(o, r, table, 32, K, validate=False)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(4):
        res = pipeline.query_similarities(o, r, table, 32, K, validate=False)
    torch.cuda.synchronize()
    rep = _lib.prof_report()
    _lib.prof_enable(False)
    tot = sum(ms for _, ms in rep.values()) / 4
    print("== %s: %d pairs, sum of kernels %.2f ms" % (name, res.pairs.numel(), tot))
    for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:9]:
        print("   %-18s %.3f ms" % (k, ms / 4))
