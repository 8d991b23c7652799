#!/usr/bin/env python3
"""bench.py -- the hot path of SURVEY.md section 8 on synthetic answer sets, on N GPUs of one node.

A step = one pass of the whole hot path (MinHash signatures + band keys -> bucket partition ->
pair emit -> group/unique -> pair scoring -> per-query top-K) over one batch of queries whose
CSR answer sets and permutation table are already resident in HBM.

Workload (BASELINE.json):
  N = 1 : configs[2] -- 10 M queries x 128-perm MinHash, 32 bands on ONE MI355X
          (D = 32768 table rows, mean answer-set size 16, clusters of 8; K = round(log_1.5 nq) = 40).
  N > 1 : configs[3] -- the SAME 10 M queries sharded N ways (strong scaling, SURVEY 8e: efficiency =
          T1 / (N * TN) on identical total work): rank g owns ceil(nq/N) consecutive query ids,
          signatures are computed per shard, bucket ids are exchanged over RCCL (band-partitioned
          all-to-all by default, --exchange all_gather for the all-gather BASELINE names), every emitted
          pair goes to the rank that scores it (the owner of one of its two queries, picked by a hash bit:
          an even split for any data), scored edges go to the owner of their source query, every rank
          cuts its own top-K.
`--nq Q` switches to weak scaling (Q queries per GPU); `--nq-total T` picks another total size.
configs[1] (1 M queries) is timed as a secondary figure of the N = 1 line, next to the two "next" rows
of SURVEY 8f that feed / consume the path (answer-set construction N2, prediction loop N1).

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import hashlib
import glob
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
    ap.add_argument("--nq-total", type=int, default=10_000_000,
                    help="queries of the whole job, sharded over the GPUs (strong scaling; default configs[2]/[3])")
    ap.add_argument("--nq", type=int, default=0, help="queries PER GPU (weak scaling); overrides --nq-total")
    ap.add_argument("--perm", type=int, default=128)
    ap.add_argument("--bands", type=int, default=32)
    ap.add_argument("--drows", type=int, default=32768)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="queries in the CPU-baseline sample (-1 = the whole workload up to 10 M, 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the CPU baseline (0 = every host core this process may run on)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-every", type=int, default=5,
                    help="bracket the kernels of every Nth timed step with HIP events (1 = every step; the events "
                         "serialise back-to-back launches and cost ~10 %% of a step when recorded on all of them)")
    ap.add_argument("--exchange", default="all_to_all", choices=["all_to_all", "all_gather"],
                    help="bucket-id exchange of the sharded path (N > 1)")
    ap.add_argument("--sig-exchange", default="auto", choices=["auto", "fetch", "sets", "all_gather", "recompute"])
    ap.add_argument("--force-dist", action="store_true", help="run the sharded driver even with one rank (testing)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu: rehearse the N > 1 code path with all ranks on ONE GPU (testing; the "
                         "collectives are staged through the host, the timings mean nothing)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (testing, with --dist-backend gloo)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / N1 / N2 secondary timings")
    return ap.parse_args()


def workload_label(nq_total, P, b, D, world, K, mean_set):
    """Name the BASELINE.json entry the shape corresponds to (or say that it is none of them)."""
    shape = "%d queries x %d-perm MinHash, %d bands, D=%d, K=%d, mean |A(q)|=%.2f" % (nq_total, P, b, D, K, mean_set)
    if (P, b, D) == (128, 32, 32768):
        if nq_total == 1_000_000 and world == 1:
            return "configs[1]: " + shape + ", single MI355X"
        if nq_total == 10_000_000 and world == 1:
            return "configs[2]: " + shape + ", single MI355X"
        if nq_total == 10_000_000 and world > 1:
            tag = "configs[3]" if world == 8 else "configs[3] shape on %d GPUs" % world
            return "%s: %s, sharded %d-way (%d queries per rank)" % (tag, shape, world, -(-nq_total // world))
    if (P, b) == (256, 64) and nq_total == 100_000_000:
        return "configs[4]: " + shape + ", sharded %d-way" % world
    return "custom shape (none of BASELINE.json's configs): " + shape + ", %d GPU(s)" % world


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
    g = w.get("group_bits", 0)                    # low bits of i the grouping sort skips (ops.row_group_bits)
    pair_passes = -(-(ib - g) // 8)               # pairs are grouped by i >> g only; rows are finished in LDS
    group_levels = 0
    if "scattered" in w.get("dedup_path", ""):    # ... by the histogram-free fixed-region partition: no sort passes over them
        group_levels = 1 if ib - g <= 8 else 2
        pair_passes = 0
    sel = w.get("topk", "select") == "select"     # top-K by rank-in-list: only the n reverse words are sorted, on j
    edge_words = -(-ib // 8) * un if sel else -(-(ib + 11) // 8) * 2 * un   # key-passes of the top-K sort
    levels = 1 if w.get("part_bits", 8) <= 8 else 2
    out = {
        # CSR in (4 B/row id + 8 B offset); signature row, fused band keys and norm out
        "minhash": 4 * nnz + 8 * nq + (sb * P + 8 * b + 8) * nq,
        # partition of the bucket path: key in, key + id out (twice for partitions finer than 256 parts)
        "sort_scatter_kv": (8 + 12) * rec,
        "part_scatter": ((8 + 12) + (levels - 1) * (12 + 12)) * rec,
        "bucket_count": 12 * rec,
        "bucket_fill": 12 * rec + 8 * em,
        "bucket_emit": 12 * rec + 8 * em,             # one-pass form (cursor-reserved output ranges)
        # keys-only LSD passes: pair words, then directed edge keys
        "sort_scatter_k": 16 * (pair_passes * em + edge_words),
        "sort_hist": 8 * (pair_passes * em + edge_words),
        # rows de-duplicated and ordered in LDS: emitted words in, distinct words out; then the gaps closed
        "row_unique": 8 * em + 8 * un,
        "row_unique_gather": 16 * un,
        "region_unique": 8 * em + 8 * un,             # the same step, one workgroup per 2^g consecutive queries
        "pair_group": 16 * em * group_levels,         # emitted words dealt into fixed regions: read + written once per level
        "region_gather": 16 * un,
        # the pairs are sorted (i, j) and the kernel keeps / re-finds the first row while i does not change: the second
        # row of every pair + the first row once per RUN of equal i (w["first_rows"]; = un when unknown, i.e. two rows
        # per pair, SURVEY 8d's count) + pair word in; score + reverse word (or two edge keys) out
        "score_pairs": (sb * P + 8 + 4 + (8 if sel else 16)) * un + sb * P * w.get("first_rows", un),
        "topk_count": 8 * 2 * un,
        "topk_fill": 8 * 2 * un + 12 * kept,
        # select form: run starts of both lists (words in, nq + 1 starts out, twice); every edge reads its own
        # record once (pair + score, reverse word) and writes its output row if it is kept -- the walk over its
        # query's runs re-reads words its neighbours in the wave have just fetched (cache, not HBM)
        "topk_bounds": 2 * (8 * un + 4 * nq),
        "topk_select": (8 + 4 + 8) * un + 12 * kept + 16 * nq,
    }
    return out


def csrc_fingerprint():
    """sha256 over the kernel sources the loaded libqrlsh.so was built from: a PMC traffic file under
    profiles/ is only quoted when it was collected on the same kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "query-recommendation-system_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(nq_total, P, b):
    """PMC-derived HBM bytes per launch per label (profiles/*_hbm_traffic.json, written by
    tools/summarise_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS
    command).  -> (by_label, file name, why-not)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    fp = csrc_fingerprint()
    why = "no profiles/*_hbm_traffic.json"
    for f in reversed(files):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        wl = j.get("workload", {})
        if (wl.get("nq_total"), wl.get("P"), wl.get("b")) != (nq_total, P, b):
            if why.startswith("no profiles"):
                why = "profiles/ holds no PMC pass of this workload"
            continue
        if j.get("csrc_sha") != fp:
            why = "the PMC pass in profiles/%s was taken on other kernel sources (%s != %s)" % (
                os.path.basename(f), j.get("csrc_sha"), fp)
            continue
        return j.get("by_label", {}), os.path.basename(f), None
    return {}, None, why


PRIME_STEPS = 3   # untimed setup steps before the warmup (see main)
PHASE_STEPS = 2   # sharded runs: untimed steps after the timed ones that collect the per-phase table


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N
    bench.py <same arguments>` as a child process (one rank per GPU over RCCL) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:           # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: WORLD_SIZE unset, launching %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as CHILD processes, before this process has
        # made any GPU call (a process that touched the GPU must never be replaced by exec); rank 0's JSON line
        # reaches our stdout through the inherited descriptor and the launcher's return code becomes ours
        sys.exit(self_launch(args.gpus))
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
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: running with the launcher's %d rank(s)\n"
                         % (args.gpus, world, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    import qrlsh
    from qrlsh import ops, pipeline, _lib
    _lib.load()  # fails loudly if the HIP extension is missing

    P, b, D = args.perm, args.bands, args.drows
    weak = args.nq > 0
    nq_total = args.nq * world if weak else args.nq_total
    from qrlsh import dist as qdist
    q0, nq_local, _ = qdist.shard_range(nq_total, world, rank)
    K = pipeline.max_candidates(nq_total)
    off, rows = qrlsh.synth_csr(nq_total, D, seed=0, q0=q0, nq_local=nq_local, device=dev)
    perms = ops.legacy_permutations(P, D, seed=42)
    table = ops.perm_table(perms, dev)
    nnz = int(rows.numel())

    phases = {}
    if not sharded:
        def step():
            return pipeline.query_similarities(off, rows, table, b, K, validate=False)   # synth_csr's own output
    else:
        backend = qdist.HipBackend()
        backend.validate = False      # synth_csr's own output, as in the one-GPU step

        def step(ph=None):
            # the timed steps run without the per-phase events (collecting them ends every step with a device
            # synchronisation); the phase table comes from PHASE_STEPS extra, untimed steps afterwards
            return qdist.query_similarities_sharded(off, rows, table, b, K, nq_total, exchange=args.exchange,
                                                    backend=backend, sig_exchange=args.sig_exchange, phases=ph)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    # setup, before the W warmup steps: the first calls of a fresh process load the code objects, grow the caching
    # allocator to the step's working set (hipMalloc of multi-GB blocks) and settle the pair-buffer capacity guess
    # (ops._EMIT_HINT); on a fresh box that can spill past two warmup steps into the timed ones
    for _ in range(PRIME_STEPS):
        res = step()
    for _ in range(args.warmup):
        res = step()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step spread (diagnostic)
    every = max(1, args.prof_every)
    prof_steps = 0
    if not args.no_prof:
        _lib.prof_enable(True)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if not args.no_prof:
            sampled = i % every == 0
            _lib.prof_pause(not sampled)
            prof_steps += sampled
        res = step()
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if sharded:
        _lib.prof_pause(True)
        for _ in range(PHASE_STEPS):
            res = step(phases)
        sync()
    prof = {}
    if not args.no_prof:
        prof = _lib.prof_report()
        _lib.prof_enable(False)
    if dist is not None:
        rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([res.pairs.numel(), res.stats.get("emitted_pairs", 0), res.src.numel()], dtype=torch.int64,
                           device=rdev)
        dist.all_reduce(cnt)
        unique_pairs, emitted, kept_total = (int(x) for x in cnt.tolist())
    else:
        unique_pairs, emitted = int(res.pairs.numel()), int(res.stats.get("emitted_pairs", 0))
        kept_total = int(res.src.numel())

    ms_per_step = elapsed / args.steps * 1e3
    value = nq_total * args.steps / elapsed

    out = None
    if rank == 0:
        rec_q = nq_total if not sharded else qdist.shard_range(nq_total, world, 0)[2] * world
        w = dict(nq=nq_local, nq_total=nq_total, nq_sorted=(rec_q // world if sharded else nq_local),
                 P=P, b=b, nnz=nnz, emitted=int(res.stats.get("emitted_pairs", 0)), unique=int(res.pairs.numel()),
                 kept=int(res.src.numel()), sig_bytes=2 if res.sig.dtype == torch.int16 else 4,
                 group_bits=int(res.stats.get("group_bits", 0)), part_bits=int(res.stats.get("part_bits", 8)),
                 dedup_path=str(res.stats.get("dedup_path", "")),
                 topk=("select" if sharded and world > 1 else res.stats.get("topk", "select")),
                 first_rows=(int((torch.diff(res.pairs >> 32) != 0).sum().item()) + 1) if res.pairs.numel() else 0)
        ab = algorithmic_bytes_per_step(w)
        sb_tab = 2 if D <= 65536 else 4
        traffic, traffic_src, traffic_why = load_traffic(nq_total, P, b) if not sharded else ({}, None, "N > 1")
        kernels = {}
        for name, (cnt_, ms) in prof.items():   # summed over the prof_steps sampled steps of the timed region
            k = {"launches_per_step": cnt_ / prof_steps, "ms_per_step": ms / prof_steps, "avg_ms": ms / cnt_}
            if name in ab and ms > 0:
                k["algorithmic_bytes_per_step"] = ab[name]
                k["algorithmic_GBps"] = round(ab[name] / (ms / prof_steps * 1e-3) / 1e9, 1)
            if name in traffic:
                k["pmc_hbm_bytes_per_launch"] = int(traffic[name]["hbm_bytes_per_launch"])
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
                roofline["traffic_source"] = "profiles/" + traffic_src
            else:
                roofline["traffic_note"] = traffic_why or "kernel not in the PMC summary"
            if dom == "minhash":
                gather = nnz * P * sb_tab
                roofline["cache_side"] = {"gather_bytes_per_launch": int(gather),
                                          "achieved_GBps": round((gather + per_launch) / (kd["avg_ms"] * 1e-3) / 1e9, 1),
                                          "l2_peak_GBps": 34500.0,
                                          "l2_served_row_gather_GBps_per_guide": 18700.0}
                roofline["note"] = ("the kernel's work is the gather of |A(q)| rows x 2P bytes per signature from the "
                                    "8 MB permutation table (%.2f GB per launch), which lives in L2 / Infinity Cache; "
                                    "FETCH_SIZE counts the L2 misses the Infinity Cache serves, hence traffic > "
                                    "algorithmic HBM bytes.  Measured (DESIGN.md section 6): with the table shrunk until "
                                    "every gather is an L2 hit the same launch takes 3.12 ms of the 3.6, and it is indifferent "
                                    "to its own occupancy and loads in flight -- the floor is the L2's delivery rate for 256-byte "
                                    "row gathers (the guide: 29 - 32 B/clk/CU = ~18.7 TB/s), not HBM" % (nnz * P * sb_tab / 1e9))

        # the same figure for the three labels with the most time per step (the dominant one's is `roofline`)
        top_rooflines = []
        for name in sorted(kernels, key=lambda k: -kernels[k]["ms_per_step"])[:3]:
            kd = kernels[name]
            if name in ab:
                ach = ab[name] / kd["launches_per_step"] / (kd["avg_ms"] * 1e-3) / 1e9
                top_rooflines.append({"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                      "ms_per_step": round(kd["ms_per_step"], 4)})

        cpu_baseline = None
        recall = None
        if args.cpu_sample != 0 and world == 1 and not sharded:      # the CPU leg belongs to the one-GPU line only
            nq_s = min(nq_total, 10_000_000) if args.cpu_sample < 0 else min(args.cpu_sample, nq_total)
            cpu_baseline, recall = cpu_leg(nq_s, nq_total, D, P, b, dev, args.cpu_threads, res, table)
        secondary = None
        if world == 1 and not sharded and not args.no_secondary:
            del res
            torch.cuda.empty_cache()
            try:       # the secondary figures must never cost the run its headline line
                secondary = secondary_figures(dev, table, P, b, D)
            except Exception as exc:  # noqa: BLE001  (reported in the line itself)
                import traceback
                sys.stderr.write("bench.py: secondary figures failed:\n" + traceback.format_exc())
                secondary = {"secondary_error": "%s: %s" % (type(exc).__name__, exc)}
            res = None

        out = {
            "metric": "MinHash signatures/sec through the whole hot path (signatures -> LSH candidates -> pair scoring -> top-K)",
            "value": round(value, 1),
            "unit": "signatures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "u16/int32 MinHash values, uint64 keys and pairs (integer); float64 cosine",
            "data": "synthetic (clustered answer sets, SURVEY 8d recipe; seed 0; permutation seed 42)",
            "config": {"workload": workload_label(nq_total, P, b, D, world, K, nnz / max(nq_local, 1)),
                       "queries_total": nq_total, "queries_per_rank": nq_local,
                       "parallelism": "query-sharded x%d" % world,
                       "bucket_id_exchange": ((res.stats.get("bucket_id_exchange", args.exchange) if res is not None
                                               else args.exchange) if sharded else "none (one GPU)"),
                       "signature_exchange": (res.stats.get("sig_exchange") if sharded and res is not None else None)},
            "pairs_scored_per_sec": round(unique_pairs * args.steps / elapsed, 1),
            "unique_pairs": unique_pairs,
            "emitted_pairs": emitted,
            "kept_edges": kept_total,
            "recall_at_10": recall,
            "kernel_timing": ("HIP events around every kernel of %d of the %d timed steps (every %d%s)"
                              % (prof_steps, args.steps, every, "th" if every > 1 else "st")) if prof_steps else None,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "roofline": roofline,
            "roofline_top3": top_rooflines,
            "cpu_baseline": cpu_baseline,
            "step_ms": {"min": round(min(step_ms), 4), "median": round(sorted(step_ms)[len(step_ms) // 2], 4),
                        "max": round(max(step_ms), 4), "setup_steps_before_warmup": PRIME_STEPS},
        }
        if sharded:
            n = max(1, phases.get("_steps", 1))
            out["phases_rank0"] = {
                "note": "rank 0, mean over %d untimed steps run after the timed ones; ms from events on the compute stream "
                        "around each phase (a collective's ms includes waiting for the slowest peer); bytes = what this "
                        "rank sent" % PHASE_STEPS,
                "ms": {k[3:]: round(v / n, 4) for k, v in sorted(phases.items()) if k.startswith("ms:")},
                "bytes_sent": {k[6:]: int(v / n) for k, v in sorted(phases.items()) if k.startswith("bytes:")},
                "sig_exchange": res.stats.get("sig_exchange") if res is not None else None}
        if secondary:
            out.update(secondary)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def secondary_figures(dev, table, P, b, D):
    """Figures quoted beside the headline on the one-GPU line: configs[1] (1 M queries) through the same
    pipeline, and the two SURVEY 8f rows either side of the path -- N2 (answer sets of attribute=value
    queries -> the CSR the path consumes) and N1 (the prediction loop that consumes its top-K lists)."""
    import qrlsh
    from qrlsh import pipeline, _lib, answers, predict
    out = {}
    # configs[1]
    nq = 1_000_000
    K = pipeline.max_candidates(nq)
    off, rows = qrlsh.synth_csr(nq, D, seed=0, device=dev)
    for _ in range(3):
        res = pipeline.query_similarities(off, rows, table, b, K, validate=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        res = pipeline.query_similarities(off, rows, table, b, K, validate=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out["configs1_1M"] = {"workload": "configs[1]: 1000000 queries x %d-perm, %d bands, D=%d, K=%d" % (P, b, D, K),
                          "ms_per_step": round(dt * 1e3, 4), "signatures_per_s": round(nq / dt, 1),
                          "unique_pairs": int(res.pairs.numel()), "steps": steps}

    def timed(label, fn, reps=5):
        fn()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        rep = _lib.prof_report()
        _lib.prof_enable(False)
        return {k: ms / reps for k, (c, ms) in rep.items()}

    # N2: 1 M two-constraint queries on a 5-feature table of D rows (45 values per feature), answer sets ~ 16 rows
    rng = np.random.RandomState(7)
    nfeat, nval = 5, 45
    cols = [rng.randint(0, nval, size=D).astype(str) for _ in range(nfeat)]
    index = answers.build_answer_index(cols, dev)
    qn = 1_000_000
    qrows = np.full((qn, nfeat), -1, dtype=np.int32)
    f1 = rng.randint(0, nfeat, size=qn)
    f2 = (f1 + 1 + rng.randint(0, nfeat - 1, size=qn)) % nfeat
    for f, v in ((f1, rng.randint(0, nval, size=qn)), (f2, rng.randint(0, nval, size=qn))):
        base = np.array([index.value_rows[x][1] for x in range(nfeat)])[f]
        qrows[np.arange(qn), f] = base + np.minimum(v, np.array([len(index.value_rows[x][0]) for x in range(nfeat)])[f] - 1)
    qd = torch.from_numpy(qrows).to(dev)
    holder = {}

    def run_n2():
        holder["csr"] = answers.answer_sets(index, qd)
    ms = timed("answers", run_n2)
    nnz2 = int(holder["csr"][1].numel())
    wpr = index.wpr
    sweep_bytes = qn * (2 * wpr * 4 + nfeat * 4 + 4 + 64 * 4)     # two bitmap rows in (cache-resident table), slots out
    compact_bytes = qn * 8 + 2 * 4 * nnz2
    n2 = {"workload": "%d two-constraint queries, table D=%d, %d features x %d values, mean |A(q)|=%.2f"
                      % (qn, D, nfeat, nval, nnz2 / qn),
          "kernels_ms": {k: round(v, 4) for k, v in ms.items()},
          "queries_per_s": round(qn / (sum(ms.values()) * 1e-3), 1),
          "algorithmic_bytes": {"answers_sweep": sweep_bytes, "answers_compact": compact_bytes},
          "note": "the sweep reads 2 bitmap rows of D/8 bytes per query out of L2 (the bitmaps are %.1f MB): cache-side "
                  "traffic, not HBM" % (index.bitmaps.numel() * 4 / 1e6)}
    for lab, by in (("answers_sweep", sweep_bytes), ("answers_compact", compact_bytes)):
        if lab in ms and ms[lab] > 0:
            n2.setdefault("algorithmic_GBps", {})[lab] = round(by / (ms[lab] * 1e-3) / 1e9, 1)
    out["next_N2_answer_sets"] = n2

    # N1: 2000 users x 100 000 queries, 75 % of the cells to predict, K = 28 query neighbours, 19 user neighbours
    nu, nqq = 2000, 100_000
    Kq, Ku = pipeline.max_candidates(nqq), pipeline.max_candidates(nu)
    ratings = rng.randint(1, 101, size=(nu, nqq)).astype(np.int32)
    ratings[rng.rand(nu, nqq) < 0.75] = 0
    deg = rng.randint(0, Kq + 1, size=nqq)
    q_src = np.repeat(np.arange(nqq, dtype=np.int32), deg)
    q_dst = rng.randint(0, nqq, size=q_src.size).astype(np.int32)
    q_mil = np.sort(rng.randint(0, 1001, size=q_src.size).astype(np.int32))[::-1].copy()
    usims = {u: {"indexes": rng.randint(0, nu, size=Ku), "values": np.round(rng.rand(Ku), 3)} for u in range(nu)}
    rt = torch.from_numpy(ratings).to(dev)
    qs, qdst, qm = (torch.from_numpy(x).to(dev) for x in (q_src, q_dst, q_mil))

    def run_n1():
        holder["pred"] = predict.fill_predictions(rt, qs, qdst, qm, usims, device=dev)
    ms1 = timed("predict", run_n1, reps=3)
    zero = int((ratings == 0).sum())
    pb = nu * nqq * 8 + zero * ((deg.mean() + Ku) * 4) + q_src.size * 12       # matrix in/out + the gathered ratings + lists
    pk = sum(v for k, v in ms1.items() if k.startswith("predict"))   # byte transpose + sweep (+ the idle fallback launches)
    out["next_N1_prediction_loop"] = {
        "workload": "%d users x %d queries, %d cells to predict, <=%d query / %d user neighbours" % (nu, nqq, zero, Kq, Ku),
        "kernels_ms": {k: round(v, 4) for k, v in ms1.items()},
        "cells_per_s": round(zero / (pk * 1e-3), 1) if pk else None,
        "algorithmic_bytes": int(pb),
        "algorithmic_GBps": round(pb / (pk * 1e-3) / 1e9, 1) if pk else None,
        "note": "algorithmic bytes = matrix read + written once, one 4-byte rating gather per neighbour of a predicted "
                "cell, the neighbour lists once"}

    # N4 (device part: centring, pairs inside the clusters, cosine, per-user cut; the scikit-learn clustering that
    # precedes it is the reference's own library call and stays on the host): the same matrix, 40 clusters
    from qrlsh import users
    labels = torch.from_numpy(rng.randint(0, 40, size=nu).astype(np.int64)).to(dev)
    del holder["pred"]
    Kn = users.max_candidates(nu)

    def run_n4():
        holder["usim"] = users.user_similarities(rt, labels, K=Kn, device=dev)
    ms4 = timed("user_similarities", run_n4, reps=3)
    npairs = int(sum(c * (c - 1) // 2 for c in np.bincount(labels.cpu().numpy())))
    sc = ms4.get("score_pairs", 0.0)
    # the step of recommender.py:216-290 that precedes the pair kernels: the clustering (StandardScaler -> PCA(200) ->
    # BIRCH, :226-261).  Round 4: StandardScaler + PCA on the device (Gram matrix of the standardized ratings on the
    # matrix cores, eigh, U sqrt(lambda): qrlsh.users.pca_features), BIRCH on its 2000 x 200 features by the
    # reference's scikit-learn call on the host; the all-host form (the reference's own three calls) timed beside it
    from sklearn.cluster import Birch
    users.pca_features(rt, dev)                      # (first call: rocSOLVER / allocator warm-up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats = users.pca_features(rt, dev)
    torch.cuda.synchronize()
    dev_feat_s = time.perf_counter() - t0
    ms_g = timed("pca_features", lambda: users.standardized_gram(rt), reps=3)
    fh = feats.cpu().numpy()
    t0 = time.perf_counter()
    dlab = Birch(n_clusters=round(nu ** (1 / 1.3))).fit(fh).predict(fh)
    birch_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    hlab = users.cluster_labels(ratings)
    host_cluster_s = time.perf_counter() - t0
    try:
        host_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        host_cores = os.cpu_count() or 1
    gram_ms = ms_g.get("user_gram", 0.0)
    out["next_N4_user_similarity"] = {
        "host_clustering_s": round(dev_feat_s + birch_s, 3),
        "clustering": {"device_standardize_pca_s": round(dev_feat_s, 4), "host_birch_s": round(birch_s, 3),
                       "gram_kernels_ms": {k: round(v, 4) for k, v in ms_g.items()},
                       "gram_TFLOPs_f64": round(nu * (nu + 128) * nqq / (gram_ms * 1e-3) / 1e12, 1) if gram_ms else None,
                       "all_host_scikit_learn_s": round(host_cluster_s, 3)},
        "host_clustering": "StandardScaler + PCA(200) on the device (float64 MFMA Gram matrix of the standardized "
                           "ratings + eigh), BIRCH by scikit-learn on the host (recommender.py:226-261); the reference's "
                           "three scikit-learn calls on the host (%d cores available) take all_host_scikit_learn_s. "
                           "This random matrix has no cluster structure (%d / %d clusters found, largest %d users) -- "
                           "the device figures below use 40 synthetic clusters of ~50 users so that the pair kernels "
                           "have work" % (host_cores, len(np.unique(dlab)), len(np.unique(hlab)), int(np.bincount(hlab).max())),
        "total_s_host_plus_device": round(dev_feat_s + birch_s + sum(ms4.values()) * 1e-3, 3),
        "workload": "%d users x %d queries in 40 clusters: %d pairs of rows, K=%d" % (nu, nqq, npairs, Kn),
        "kernels_ms": {k: round(v, 4) for k, v in sorted(ms4.items(), key=lambda kv: -kv[1])[:6]},
        "device_ms_total": round(sum(ms4.values()), 4),
        "pairs_per_s": round(npairs / (sum(ms4.values()) * 1e-3), 1) if ms4 else None,
        "algorithmic_bytes": {"center_rows": int(nu * nqq * 8), "row_norms": int(nu * nqq * 4),
                              "score_pairs_rows_read": int(npairs * 2 * nqq * 4)},
        "cache_side_GBps": {"score_pairs": round(npairs * 2 * nqq * 4 / (sc * 1e-3) / 1e9, 1)} if sc else None,
        "note": "score_pairs reads both centred int32 rows of every pair (the reference builds the same Gram "
                "matrices with sklearn's cosine_similarity per cluster); a cluster's rows (50 x 400 KB) are shared "
                "by its 1225 pairs, so those reads are served by L2 / Infinity Cache: the figure is cache-side "
                "traffic, the HBM side is the matrix once"}
    return out


def host_cores():
    """(cores this process may be scheduled on, cgroup CPU quota in cores or None when unlimited)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            quota = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    return avail, quota


def cpu_leg(nq_s, nq_total, D, P, b, dev, threads, res_full, table):
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
    avail, quota = host_cores()
    usable = avail if quota is None else max(1, min(avail, int(quota + 0.5)))
    cores = usable if threads <= 0 else max(1, min(threads, avail))
    O.set_threads(cores)
    O.query_similarities(ho[:2001], hr[:ho[2000]], D, P, b, K, 42)  # warm the library / threads
    t0 = time.perf_counter()
    sig = O.minhash(ho, hr, perms)
    t1 = time.perf_counter()
    keys = O.band_keys(sig, b)
    pairs = O.candidates(keys, P // b)
    del keys
    t2 = time.perf_counter()
    milli = O.score_pairs(sig, pairs, mode=1)
    s, d, v = O.topk(pairs, milli, K)
    t3 = time.perf_counter()
    if nq_s == nq_total:
        res = res_full                       # the sample IS the benchmarked workload: check the timed run's own output
    else:
        off = torch.from_numpy(ho).to(dev)
        rows = torch.from_numpy(hr).to(dev)
        res = pipeline.query_similarities(off, rows, table, b, K)
    torch.cuda.synchronize()
    exact = bool(np.array_equal(res.sig_int32().cpu().numpy(), sig)
                 and np.array_equal(res.pairs.cpu().numpy().view(np.uint64), pairs)
                 and np.array_equal(res.milli.cpu().numpy(), milli))
    gs, gd, gv = res.src.cpu().numpy(), res.dst.cpu().numpy(), res.val.cpu().numpy()
    topk_exact = bool(np.array_equal(gs, s) and np.array_equal(gd, d) and np.array_equal(gv, v))
    # recall@10 on the lists of the first 200 000 queries (the tie-aware comparison is a Python loop)
    rq = min(nq_s, 200_000)
    ce, cg = int(np.searchsorted(s, rq)), int(np.searchsorted(gs, rq))
    recall = recall_at_k(s[:ce], d[:ce], v[:ce], gs[:cg], gd[:cg], gv[:cg], 10)
    total = t3 - t0
    del sig
    # the same port on ONE thread (the reference itself is single-threaded), on a smaller sample
    n1 = min(nq_s, 200_000)
    o1, r1 = O.synth_csr(n1, D, seed=0)
    O.set_threads(1)
    t4 = time.perf_counter()
    O.query_similarities(o1, r1, D, P, b, pipeline.max_candidates(n1), 42)
    t5 = time.perf_counter()
    O.set_threads(cores)
    base = {
        "value": round(nq_s / total, 1), "unit": "signatures/s", "cores": cores, "cores_available": avail,
        "cgroup_cpu_quota": quota, "kind": "port",
        "single_thread": {"value": round(n1 / (t5 - t4), 1), "unit": "signatures/s", "cores": 1,
                          "sample": "whole hot path on nq=%d, 1 thread" % n1, "seconds": round(t5 - t4, 3)},
        "sample": "whole hot path on nq=%d queries of the same synthetic recipe (P=%d, b=%d, D=%d)%s, oracle/qr_oracle.c "
                  "with OpenMP, %d threads" % (nq_s, P, b, D, " = the benchmarked workload" if nq_s == nq_total else "", cores),
        "seconds": round(total, 3),
        "phases_s": {"signatures": round(t1 - t0, 3), "candidates": round(t2 - t1, 3), "scoring_topk": round(t3 - t2, 3)},
        "minhash_signatures_per_s": round(nq_s / (t1 - t0), 1),
        "pairs_scored_per_s": round(len(pairs) / max(t3 - t2, 1e-9), 1),
        "gpu_bit_exact_on_sample": exact, "gpu_topk_exact_on_sample": topk_exact,
        "recall_sample": "tie-aware recall@10 over the lists of the first %d queries" % rq,
    }
    return base, round(recall, 6)


if __name__ == "__main__":
    main()
