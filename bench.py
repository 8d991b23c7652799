#!/usr/bin/env python3
"""bench.py -- the hot path of SURVEY.md section 8 on synthetic answer sets, on N GPUs of one node.

A step = one pass of the whole hot path (MinHash signatures + band keys -> bucket sort ->
pair emit -> sort/unique -> pair scoring -> per-query top-K) over one batch of queries whose
CSR answer sets and permutation table are already resident in HBM.

Workload at N=1: BASELINE.json configs[1] -- 1 M queries x 128-perm MinHash, 32 bands
(D = 32768 table rows, mean answer-set size 16, clusters of 8; K = round(log_1.5 nq) = 34).
For N > 1 every rank owns 1 M queries of an N x 1 M problem (weak scaling): signatures are
computed per shard, band keys are all-gathered over RCCL, bands are split across ranks for
the bucket sort, emitted pairs are exchanged to their owner (all-to-all), and owners score.

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nq", type=int, default=1_000_000, help="queries per GPU")
    ap.add_argument("--perm", type=int, default=128)
    ap.add_argument("--bands", type=int, default=32)
    ap.add_argument("--drows", type=int, default=32768)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="queries in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU baseline (box share: 16/GPU)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-every", type=int, default=5,
                    help="bracket the kernels of every Nth timed step with HIP events (1 = every step; the events "
                         "serialise back-to-back launches and cost ~10 %% of a step when recorded on all of them)")
    ap.add_argument("--exchange", default="all_to_all", choices=["all_to_all", "all_gather"],
                    help="bucket-id exchange of the sharded path (N > 1)")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded driver even with one rank (testing)")
    return ap.parse_args()


def algorithmic_bytes_per_step(w):
    """Algorithmic HBM bytes each kernel label moves in ONE step of workload w (DESIGN.md section 4:
    what the stage must read and write once, with this build's layouts -- compact uint16 signature
    rows when D <= 65535, uint64 keys, 8-byte pairs and edge keys).  Labels = the names the
    library's HIP-event profiler and tools/summarise_profile.py use."""
    nq, P, b, nnz = w["nq"], w["P"], w["b"], w["nnz"]
    em, un, kept = w["emitted"], w["unique"], w["kept"]
    sb = w["sig_bytes"]
    rec = b * w["nq_sorted"]                      # (band, query) records this rank buckets
    ib = max(1, (w["nq_total"] - 1).bit_length())
    g = ib % 8 if (ib > 8 and 0 < ib % 8 <= 4) else 0   # ops.row_group_bits: low bits of i the grouping sort skips
    pair_passes = -(-(ib - g) // 8)               # pairs are grouped by i >> g only; rows are finished in LDS
    edge_passes = -(-(ib + 11) // 8)
    out = {
        # CSR in (4 B/row id + 8 B offset); signature row, fused band keys and norm out
        "minhash": 4 * nnz + 8 * nq + (sb * P + 8 * b + 8) * nq,
        # partition pass of the bucket path: key in, key + id out
        "sort_scatter_kv": (8 + 12) * rec,
        "part_scatter": (8 + 12) * rec,               # one-kernel partition (atomic room reservation per tile and part)
        "bucket_count": 12 * rec,
        "bucket_fill": 12 * rec + 8 * em,
        "bucket_emit": 12 * rec + 8 * em,             # one-pass form (cursor-reserved output ranges)
        # keys-only LSD passes: pair words, then directed edge keys
        "sort_scatter_k": 16 * (pair_passes * em + edge_passes * 2 * un),
        "sort_hist": 8 * (pair_passes * em + edge_passes * 2 * un),
        # rows de-duplicated and ordered in LDS: emitted words in, distinct words out; then the gaps closed
        "row_unique": 8 * em + 8 * un,
        "row_unique_gather": 16 * un,
        # two signature rows + pair word in; score + two edge keys out
        "score_pairs": (2 * sb * P + 8 + 4 + 16) * un,
        "topk_count": 8 * 2 * un,
        "topk_fill": 8 * 2 * un + 12 * kept,
    }
    return out


def load_traffic():
    """PMC-derived HBM bytes per launch per label (profiles/*_hbm_traffic.json, written by
    tools/summarise_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return {}, None
    try:
        return json.load(open(files[-1])).get("by_label", {}), os.path.basename(files[-1])
    except Exception:
        return {}, None


def main():
    args = parse()
    # RCCL (and other native libraries) print banners on fd 1 when a communicator comes up; the
    # contract is ONE JSON line on stdout, so everything else is sent to stderr for the whole run
    # and the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import qrlsh
    from qrlsh import ops, pipeline, _lib
    _lib.load()  # fails loudly if the HIP extension is missing

    nq_local, P, b, D = args.nq, args.perm, args.bands, args.drows
    nq_total = nq_local * world
    K = pipeline.max_candidates(nq_total)
    q0 = rank * nq_local
    off, rows = qrlsh.synth_csr(nq_total, D, seed=0, q0=q0, nq_local=nq_local, device=dev)
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, dev)
    nnz = int(rows.numel())

    if world == 1 and not args.force_dist:
        def step():
            return pipeline.query_similarities(off, rows, table, b, K)
    else:
        from qrlsh import dist as qdist

        def step():
            return qdist.query_similarities_sharded(off, rows, table, b, K, nq_total, exchange=args.exchange)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    sync()
    every = max(1, args.prof_every)
    prof_steps = 0
    if not args.no_prof:
        _lib.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_prof:
            sampled = i % every == 0
            _lib.prof_pause(not sampled)
            prof_steps += sampled
        res = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = {}
    if not args.no_prof:
        prof = _lib.prof_report()
        _lib.prof_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([res.pairs.numel(), res.stats.get("emitted_pairs", 0)], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        unique_pairs, emitted = int(cnt[0].item()), int(cnt[1].item())
    else:
        unique_pairs, emitted = int(res.pairs.numel()), int(res.stats.get("emitted_pairs", 0))

    ms_per_step = elapsed / args.steps * 1e3
    value = nq_total * args.steps / elapsed

    out = None
    if rank == 0:
        w = dict(nq=nq_local, nq_total=nq_total, nq_sorted=nq_local,
                 P=P, b=b, nnz=nnz, emitted=int(res.stats.get("emitted_pairs", 0)), unique=int(res.pairs.numel()),
                 kept=int(res.src.numel()), sig_bytes=2 if res.sig.dtype == torch.int16 else 4)
        ab = algorithmic_bytes_per_step(w)
        sb_tab = 2 if D <= 65536 else 4
        traffic, traffic_src = load_traffic()
        kernels = {}
        for name, (cnt_, ms) in prof.items():   # summed over the prof_steps sampled steps of the timed region
            k = {"launches_per_step": cnt_ / prof_steps, "ms_per_step": ms / prof_steps, "avg_ms": ms / cnt_}
            if name in ab and ms > 0:
                k["algorithmic_bytes_per_step"] = ab[name]
                k["algorithmic_GBps"] = round(ab[name] / (ms / prof_steps * 1e-3) / 1e9, 1)
            kernels[name] = k
        roofline = None
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
            kd = kernels[dom]
            lps = kd["launches_per_step"]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": None, "avg_launch_ms": round(kd["avg_ms"], 4),
                        "launches_per_step": lps}
            if dom in ab:
                per_launch = ab[dom] / lps
                achieved = per_launch / (kd["avg_ms"] * 1e-3) / 1e9
                roofline.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_launch": int(per_launch)})
            if dom in traffic:
                roofline["traffic"] = int(traffic[dom]["hbm_bytes_per_launch"])
                roofline["traffic_source"] = traffic_src
            if dom == "minhash":
                gather = nnz * P * sb_tab
                roofline["cache_side"] = {"gather_bytes_per_launch": int(gather),
                                          "achieved_GBps": round((gather + per_launch) / (kd["avg_ms"] * 1e-3) / 1e9, 1),
                                          "l2_peak_GBps": 34500.0}
                roofline["note"] = ("the kernel's work is the gather of |A(q)| rows x 2P bytes per signature from the "
                                    "8 MB permutation table (%.2f GB per launch), which lives in L2 / Infinity Cache; "
                                    "FETCH_SIZE counts the L2 misses the Infinity Cache serves, hence traffic > "
                                    "algorithmic HBM bytes (DESIGN.md section 6)" % (nnz * P * sb_tab / 1e9))

        cpu_baseline = None
        recall = None
        if args.cpu_sample > 0 and world == 1:      # the CPU leg belongs to the one-GPU line only
            cpu_baseline, recall = cpu_leg(args.cpu_sample, D, P, b, dev, args.cpu_threads)

        out = {
            "metric": "MinHash signatures/sec through the whole hot path (signatures -> LSH candidates -> pair scoring -> top-K)",
            "value": round(value, 1),
            "unit": "signatures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16/int32 MinHash values, uint64 keys and pairs (integer); float64 cosine",
            "data": "synthetic (clustered answer sets, SURVEY 8d recipe; seed 0; permutation seed 42)",
            "config": {"workload": "configs[1]: %d queries/GPU x %d-perm MinHash, %d bands, D=%d, K=%d, mean |A(q)|=%.2f"
                       % (nq_local, P, b, D, K, nnz / nq_local),
                       "queries_total": nq_total, "parallelism": "query-sharded x%d" % world,
                       "bucket_id_exchange": (args.exchange if (world > 1 or args.force_dist) else "none (one GPU)")},
            "pairs_scored_per_sec": round(unique_pairs * args.steps / elapsed, 1),
            "unique_pairs": unique_pairs,
            "emitted_pairs": emitted,
            "recall_at_10": recall,
            "kernel_timing": ("HIP events around every kernel of %d of the %d timed steps (every %d%s)"
                              % (prof_steps, args.steps, every, "th" if every > 1 else "st")) if prof_steps else None,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def cpu_leg(nq_s, D, P, b, dev, threads):
    """CPU baseline (oracle = a C port of the reference's algorithm, OpenMP) on a bounded
    sample of the same workload, and recall@10 of the GPU path against it."""
    import qrlsh
    from qrlsh import ops, pipeline
    from oracle import oracle as O  # the checker / CPU baseline; never on the measured GPU path
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import recall_at_k

    K = pipeline.max_candidates(nq_s)
    perms = ops.legacy_permutations(P, D, seed=42)
    ho, hr = O.synth_csr(nq_s, D, seed=0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(threads, avail))
    O.set_threads(cores)
    O.query_similarities(ho[:2001], hr[:ho[2000]], D, P, b, K, 42)  # warm the library / threads
    t0 = time.perf_counter()
    sig = O.minhash(ho, hr, perms)
    t1 = time.perf_counter()
    keys = O.band_keys(sig, b)
    pairs = O.candidates(keys, P // b)
    t2 = time.perf_counter()
    milli = O.score_pairs(sig, pairs, mode=1)
    s, d, v = O.topk(pairs, milli, K)
    t3 = time.perf_counter()
    off = torch.from_numpy(ho).to(dev)
    rows = torch.from_numpy(hr).to(dev)
    res = pipeline.query_similarities(off, rows, ops.perm_table(perms, dev), b, K)
    torch.cuda.synchronize()
    exact = bool(np.array_equal(res.sig_int32().cpu().numpy(), sig)
                 and np.array_equal(res.pairs.cpu().numpy().view(np.uint64), pairs)
                 and np.array_equal(res.milli.cpu().numpy(), milli))
    recall = recall_at_k(s, d, v, res.src.cpu().numpy(), res.dst.cpu().numpy(), res.val.cpu().numpy(), 10)
    total = t3 - t0
    # the same port on ONE thread (the reference itself is single-threaded), on a smaller sample
    n1 = min(nq_s, 200_000)
    o1, r1 = O.synth_csr(n1, D, seed=0)
    O.set_threads(1)
    t4 = time.perf_counter()
    O.query_similarities(o1, r1, D, P, b, pipeline.max_candidates(n1), 42)
    t5 = time.perf_counter()
    O.set_threads(cores)
    base = {
        "value": round(nq_s / total, 1), "unit": "signatures/s", "cores": cores, "kind": "port",
        "single_thread": {"value": round(n1 / (t5 - t4), 1), "unit": "signatures/s", "cores": 1,
                          "sample": "whole hot path on nq=%d, 1 thread" % n1, "seconds": round(t5 - t4, 3)},
        "sample": "whole hot path on nq=%d queries of the same synthetic recipe (P=%d, b=%d, D=%d), oracle/qr_oracle.c with OpenMP, %d threads"
                  % (nq_s, P, b, D, cores),
        "seconds": round(total, 3),
        "phases_s": {"signatures": round(t1 - t0, 3), "candidates": round(t2 - t1, 3), "scoring_topk": round(t3 - t2, 3)},
        "minhash_signatures_per_s": round(nq_s / (t1 - t0), 1),
        "pairs_scored_per_s": round(len(pairs) / max(t3 - t2, 1e-9), 1),
        "gpu_bit_exact_on_sample": exact,
    }
    return base, round(recall, 6)


if __name__ == "__main__":
    main()
