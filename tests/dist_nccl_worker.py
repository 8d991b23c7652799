"""Worker for the RCCL rehearsal on a one-GPU box: ONE nccl rank that makes RCCL execute every collective shape the
N-rank sharded step issues (qrlsh.dist._all_to_all with split lists, the uint8 views of int16 rows, asynchronous
all-gathers on the second communicator, the size exchange, empty messages), checks the bytes that come back, and then
runs the sharded driver with force_collectives=True in every exchange / signature mode against the one-GPU pipeline.
Started as a child process by tests/test_gpu_parity.py (the test process itself never joins a process group)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import qrlsh  # noqa: E402
from qrlsh import ops, pipeline  # noqa: E402
from qrlsh import dist as qdist  # noqa: E402


def collective_shapes(dev):
    g = torch.Generator(device="cpu").manual_seed(5)
    world = dist.get_world_size()
    assert world == 1 and dist.get_backend() == "nccl"
    # variable all-to-all of 64-bit words with split lists (pairs, edge keys, row requests)
    a = torch.randint(-2 ** 62, 2 ** 62, (100_003,), dtype=torch.int64, generator=g).to(dev)
    out = torch.empty_like(a)
    qdist._all_to_all(out, a, [a.numel()], [a.numel()])
    assert torch.equal(out, a)
    # ... of the 32-bit payload that travels beside wide-id edge keys
    d = torch.randint(-2 ** 31, 2 ** 31 - 1, (100_003,), dtype=torch.int32, generator=g).to(dev)
    dout = torch.empty_like(d)
    qdist._all_to_all(dout, d, [d.numel()], [d.numel()])
    assert torch.equal(dout, d)
    # ... of compact signature rows: int16 [n, P] sent as a 2-D uint8 view, split by ROW counts
    rows = torch.randint(-2 ** 15, 2 ** 15 - 1, (4097, 128), dtype=torch.int16, generator=g).to(dev)
    rout = torch.empty_like(rows)
    qdist._all_to_all(rout, rows, [rows.shape[0]], [rows.shape[0]])
    assert torch.equal(rout, rows)
    # ... the band-partitioned bucket-id exchange: [b][nql] in, [world * nb][nql] out, split by bands
    keys = torch.randint(-2 ** 62, 2 ** 62, (32, 5000), dtype=torch.int64, generator=g).to(dev)
    kout = torch.empty_like(keys)
    qdist._all_to_all(kout, keys, [32], [32])
    assert torch.equal(kout, keys)
    # ... and with nothing to send (a rank whose pairs touch no remote query)
    e = torch.empty((0,), dtype=torch.int64, device=dev)
    eout = torch.empty((0,), dtype=torch.int64, device=dev)
    qdist._all_to_all(eout, e, [0], [0])
    e2 = torch.empty((0, 128), dtype=torch.int16, device=dev)
    qdist._all_to_all(torch.empty_like(e2), e2, [0], [0])
    # the size exchange: one small all-to-all + one read-back
    bounds = torch.tensor([0, 12345], dtype=torch.int64, device=dev)
    s, r = qdist._exchange_sizes(bounds)
    assert s == [12345] and r == [12345]
    # all-gathers: synchronous on the main communicator (the bucket-id all-gather north_star names) ...
    ag = torch.empty((world * 32, 5000), dtype=torch.int64, device=dev)
    qdist._all_gather(ag, keys)
    assert torch.equal(ag, keys)
    # ... asynchronous on the second communicator: int16 answer-set row ids as a (1, n) view, int32 offsets, int16 rows,
    #     int64 norms; the main communicator stays usable while they are in flight
    bg = qdist.background_group(None, force=True)
    assert bg is not None and bg is not dist.group.WORLD
    ids16 = torch.randint(-2 ** 15, 2 ** 15 - 1, (1_000_001,), dtype=torch.int16, generator=g).to(dev)
    off32 = torch.arange(50_001, dtype=torch.int32, device=dev)
    o_ids = torch.empty((world, ids16.numel()), dtype=torch.int16, device=dev)
    o_off = torch.empty((world, off32.numel()), dtype=torch.int32, device=dev)
    h1 = qdist._Pending(qdist._all_gather(o_ids, ids16.view(1, -1), bg, async_op=True), ids16)
    h2 = qdist._Pending(qdist._all_gather(o_off, off32.view(1, -1), bg, async_op=True), off32)
    qdist._all_to_all(out, a, [a.numel()], [a.numel()])       # the foreground communicator, meanwhile
    sa = torch.empty((world * rows.shape[0], 128), dtype=torch.int16, device=dev)
    na = torch.empty((world * 4097,), dtype=torch.int64, device=dev)
    nrm = torch.arange(4097, dtype=torch.int64, device=dev)
    h3 = qdist._all_gather(sa, rows, bg, async_op=True)
    h4 = qdist._all_gather(na, nrm, bg, async_op=True)
    for h in (h1, h2, h3, h4):
        h.wait()
    torch.cuda.synchronize()
    assert torch.equal(o_ids[0], ids16) and torch.equal(o_off[0], off32) and torch.equal(sa, rows) and torch.equal(na, nrm)
    print("collective shapes ok")


def forced_steps(dev, nq, D, P, b):
    K = pipeline.max_candidates(nq)
    table = ops.perm_table(ops.legacy_permutations(P, D, seed=42), dev)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=dev)
    ref = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    for exchange, sig_mode, dedup in (("all_to_all", "sets", None), ("all_gather", "fetch", True), ("all_to_all", "all_gather", None),
                                      ("all_to_all", "recompute", True), ("all_gather", "sets", None)):
        phases = {}
        res = qdist.query_similarities_sharded(off, rows, table, b, K, nq, exchange=exchange, sig_exchange=sig_mode,
                                               phases=phases, force_collectives=True, local_dedup=dedup)
        torch.cuda.synchronize()
        assert res.stats["sig_exchange"] == sig_mode
        assert res.stats["topk"].startswith("select")            # the N-rank top-K (received edges), not the one-rank one
        for name in ("sig", "norm2", "pairs", "milli", "src", "dst", "val"):
            assert torch.equal(getattr(res, name), getattr(ref, name)), (exchange, sig_mode, name)
        assert res.stats["emitted_pairs"] == ref.stats["emitted_pairs"]
        sent = sorted(k for k in phases if k.startswith("bytes:"))
        print("forced world-1 step ok: nq=%d %s / %s  collectives: %s" % (nq, exchange, sig_mode, ", ".join(sent)))


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    collective_shapes(dev)
    forced_steps(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 40000, 32768, 128, 32)
    if len(sys.argv) > 2:
        forced_steps(dev, int(sys.argv[2]), 32768, 128, 32)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
