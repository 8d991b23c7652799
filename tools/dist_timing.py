#!/usr/bin/env python3
"""Where the sharded driver's wall time goes at world = 1 (development tool): wraps the backend
methods and the collectives with synchronised timestamps.  usage: python tools/dist_timing.py [nq]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
import qrlsh  # noqa: E402
from qrlsh import ops, dist as qd  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P, b, D, K = 128, 32, 32768, 34
off, rows = qrlsh.synth_csr(nq, D, seed=0, device=dev)
table = ops.perm_table(ops.legacy_permutations(P, D, seed=42), dev)
acc = {}
last = [0.0]


def mark(name):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - last[0])
    last[0] = t


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        mark("host/other")
        r = f(*a, **k)
        mark(label or name)
        return r
    setattr(obj, name, g)


be = qd.HipBackend()
for m in ("minhash", "emit_pairs", "sort_unique", "group_by_owner", "owner_sizes", "score_only", "topk"):
    wrap(be, m)
wrap(qd, "_all_to_all", "collective:all_to_all")
wrap(qd, "_all_gather", "collective:all_gather")
mode = sys.argv[2] if len(sys.argv) > 2 else "auto"
for it in range(4):
    acc.clear()
    torch.cuda.synchronize()
    t0 = last[0] = time.perf_counter()
    res = qd.query_similarities_sharded(off, rows, table, b, K, nq, backend=be, sig_exchange=mode)
    mark("host/other")
    total = time.perf_counter() - t0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-26s %8.3f ms" % (k, v * 1e3))
print("total (with the syncs this tool adds) %.3f ms; sig_exchange=%s" % (total * 1e3, res.stats["sig_exchange"]))
dist.destroy_process_group()
