"""Worker for the GPU dist test: W gloo ranks sharing cuda:0, real HIP kernels (HipBackend),
collectives staged through the host (qrlsh.dist._staged).  Exercises the sharded driver with
global ids / owned-band subsets on the device; RCCL itself is exercised by bench.py --gpus N."""
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


def main():
    out_dir, nq, D, P, b, mode, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    q0, n_real, nql = qdist.shard_range(nq, world, rank)
    K = pipeline.max_candidates(nq)
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, dev)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, q0=q0, nq_local=n_real, device=dev)
    sig_mode = sys.argv[8] if len(sys.argv) > 8 else "auto"
    mean = float(sys.argv[9]) if len(sys.argv) > 9 else 16.0
    if mean != 16.0:
        off, rows = qrlsh.synth_csr(nq, D, seed=0, mean=mean, q0=q0, nq_local=n_real, device=dev)
    if os.environ.get("QRLSH_TEST_TINY_EMIT_HINT"):
        class _Tiny(dict):          # every shape looks "seen before, 16 pairs": the one-pass emit must count, then retry
            def __contains__(self, key):
                return True

            def __missing__(self, key):
                return 16
        ops._EMIT_HINT = _Tiny()
    phases = {}
    res = qdist.query_similarities_sharded(off, rows, table, b, K, nq, exchange=mode, sig_exchange=sig_mode,
                                           phases=phases)
    assert phases["_steps"] == 1 and any(k.startswith("ms:") for k in phases)
    if os.environ.get("QRLSH_TEST_TINY_EMIT_HINT") and res.stats["emitted_pairs"] > 0:
        assert all(v > 16 for v in dict.values(ops._EMIT_HINT)) and len(ops._EMIT_HINT) > 0   # the retry recorded the real count
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), pairs=res.pairs.cpu().numpy(), milli=res.milli.cpu().numpy(),
             src=res.src.cpu().numpy(), dst=res.dst.cpu().numpy(), val=res.val.cpu().numpy(),
             emitted=res.stats["emitted_pairs"], sig_exchange=res.stats["sig_exchange"],
             fetched=res.stats.get("remote_rows_fetched", -1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
