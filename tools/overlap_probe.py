#!/usr/bin/env python3
"""Does MinHash (a per-CU miss-queue-bound gather) share the device with an HBM-streaming kernel?  MinHash alone, a
copy of `gb` GB alone, both at once on two streams.  Build with -DMH_WAVES=4 to cap MinHash's occupancy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import torch
import qrlsh
from qrlsh import ops
nq, D = 10_000_000, 32768
off, rows = qrlsh.synth_csr(nq, D, seed=0, device="cuda")
table = ops.perm_table(ops.legacy_permutations(128, D, seed=42), "cuda")
src = torch.empty((400_000_000,), dtype=torch.int64, device="cuda")   # 3.2 GB: read + write 6.4 GB, the first partition level's traffic
dst = torch.empty_like(src)
s2 = torch.cuda.Stream()
def mh():
    ops.minhash(off, rows, table, b=32, compact=True, validate=False)
def cp():
    with torch.cuda.stream(s2):
        dst.copy_(src)
def timed(fn, n=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def both():
    mh(); cp()
print("minhash alone %.3f ms, copy of 6.4 GB traffic alone %.3f ms, both at once %.3f ms" % (timed(mh), timed(cp), timed(both)))
