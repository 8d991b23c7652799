#!/usr/bin/env python3
"""Per-kernel times of ONE emitter rank of configs[4] (100 M queries x 256 / 64, 8 bands over all 100 M ids in the
exchanged [rank][band][queries] layout): where the partition + finish of a giant-bucket workload spends its time.
python tools/config4_emit_probe.py [emitter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import qrlsh  # noqa: E402
from qrlsh import ops, _lib  # noqa: E402
from qrlsh import dist as qdist  # noqa: E402

e = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nq, D, P, b, world = 100_000_000, 32768, 256, 64, 8
r = P // b
dev = "cuda"
nql = nq // world
lo, hi = qdist.band_owner_ranges(b, world)[e]
nb = hi - lo
table = ops.perm_table(ops.legacy_permutations(P, D, seed=42), dev)
recv = torch.empty((world, nb, nql), dtype=torch.int64, device=dev)
for s_ in range(world):
    off, rows = qrlsh.synth_csr(nq, D, seed=0, q0=s_ * nql, nq_local=nql, device=dev)
    _, _, keys = ops.minhash(off, rows, table, b=b, compact=True, validate=False)
    recv[s_].copy_(keys[lo:hi])
    del off, rows, keys
be = qdist.HipBackend()
pairs = be.emit_pairs_chunked(recv.view(-1), world, nb, nql, r)
n = pairs.numel()
del pairs
torch.cuda.synchronize()
_lib.prof_enable(True)
pairs = be.emit_pairs_chunked(recv.view(-1), world, nb, nql, r)
torch.cuda.synchronize()
rep = _lib.prof_report()
_lib.prof_enable(False)
print("emitter %d: %d words emitted, path %s" % (e, n, be.stats["bucket_path"]))
for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print("%-18s launches %3d  ms %9.3f" % (k, c, ms))
import time
for ov in (1, 0):
    _lib.load().qrlsh_set_overlap(ov)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p2 = be.emit_pairs_chunked(recv.view(-1), world, nb, nql, r)
    torch.cuda.synchronize(); print("overlap %d: wall %.1f ms" % (ov, (time.perf_counter() - t0) * 1e3)); del p2
