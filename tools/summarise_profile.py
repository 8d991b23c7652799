#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes)
into the small files committed under profiles/.

usage: summarise_profile.py <tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir> [nq_total P b]
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_traffic.csv, profiles/<tag>_hbm_traffic.json
The JSON records the workload the passes ran (default: bench.py's default, 10 M x 128 / 32) and a hash of
the kernel sources at the time of the run: bench.py quotes roofline.traffic from it only when both match.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:60]
    base = m.group(1).split("::")[-1]
    return base + (m.group(2) or "")


# rocprof kernel name -> the label the library's HIP-event profiler (and bench.py) uses
LABELS = [
    (r"^minhash_", "minhash"), (r"^band_keys_kernel", "band_keys"), (r"^row_norms_kernel", "row_norms"),
    (r"^sort_hist_kernel", "sort_hist"), (r"^sort_rowscan_kernel", "sort_rowscan"),
    (r"^sort_scatter_kernel<\d+, false", "sort_scatter_k"), (r"^sort_scatter_kernel<\d+, true", "sort_scatter_kv"),
    (r"^bucket_finish_kernel<0[,>]", "bucket_count"), (r"^bucket_finish_kernel<1[,>]", "bucket_fill"),
    (r"^bucket_finish_kernel<2[,>]", "bucket_emit"), (r"^bucket_bounds_kernel", "bucket_bounds"),
    (r"^bucket_finish_big_kernel", "bucket_emit_big"), (r"^bucket_finish_packed_kernel", "bucket_emit"),
    (r"^pairs_count_kernel", "pairs_count"), (r"^pairs_fill_kernel", "pairs_fill"),
    (r"^row_unique_kernel", "row_unique"), (r"^row_unique_gather_kernel", "row_unique_gather"),
    (r"^sort_scatter_staged_kernel", "sort_scatter_k"), (r"^part_scatter_staged_kernel", "sort_scatter_kv"),
    (r"^part_scatter_atomic_kernel", "part_scatter"), (r"^row_unique_long_kernel", "row_unique_long"),
    (r"^region_unique_kernel", "region_unique"), (r"^region_gather_kernel", "region_gather"),
    (r"^region_bounds_kernel", "region_bounds"), (r"^region_unique_big_kernel", "region_unique_big"),
    (r"^edge_bounds_kernel", "topk_bounds"), (r"^edge_bounds_fix_kernel", "topk_bounds"), (r"^topk_len_kernel", "topk_len"),
    (r"^topk_select_short_kernel", "topk_select"), (r"^topk_select_medium_kernel", "topk_select_medium"),
    (r"^topk_select_long_kernel", "topk_select_long"), (r"^center_rows_kernel", "center_rows"),
    (r"^compact_count_kernel<0>", "unique_count"), (r"^compact_fill_kernel<0>", "unique_fill"),
    (r"^compact_count_kernel<1>", "topk_count"), (r"^compact_fill_kernel<1>", "topk_fill"),
    (r"^score_pairs_kernel", "score_pairs"), (r"^score_runs_kernel", "score_pairs"), (r"^scan_u64_kernel", "scan_blocks"),
    (r"^synth_kernel", "synth"), (r"^pair_group_scatter_kernel", "pair_group"), (r"^region_spans_kernel", "region_bounds"),
    (r"^bucket_big_gather_kernel", "bucket_emit_big"), (r"^user_gram_kernel", "user_gram"),
    (r"^column_stats_kernel", "user_colstats"), (r"^gram_reduce_kernel", "user_gram_reduce"),
]


def label_of(kname):
    for pat, lab in LABELS:
        if re.search(pat, kname):
            return lab
    return None


def pmc(dirname, counter):
    f = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(list)
    if not f:
        return acc
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    tag, trace, fdir, wdir = sys.argv[1:5]
    wl = [int(x) for x in sys.argv[5:8]] if len(sys.argv) >= 8 else [10_000_000, 128, 32]
    from bench import csrc_fingerprint
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(trace, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"],
                        "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])
    fetch, write = pmc(fdir, "FETCH_SIZE"), pmc(wdir, "WRITE_SIZE")
    traffic = {}
    with open(os.path.join(out, tag + "_hbm_traffic.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg_raw", "WRITE_SIZE_KB_avg_raw",
                    "read_bytes_avg_x2_gfx950", "write_bytes_avg", "hbm_bytes_avg"])
        for k in sorted(set(fetch) | set(write)):
            f = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
            wr = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
            # MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KB and reports exactly half of the bytes of a
            # wide coalesced read on gfx950 -> doubled; WRITE_SIZE reads exact.
            rb, wb = f * 1024 * 2, wr * 1024
            traffic[k] = {"launches": max(len(fetch.get(k, [])), len(write.get(k, []))),
                          "read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb}
            w.writerow([k, traffic[k]["launches"], "%.1f" % f, "%.1f" % wr, "%.0f" % rb, "%.0f" % wb, "%.0f" % (rb + wb)])
    by_label = {}
    for k, t in traffic.items():
        lab = label_of(k)
        if lab is None:
            continue
        e = by_label.setdefault(lab, {"launches": 0, "bytes": 0.0})
        e["launches"] += t["launches"]
        e["bytes"] += t["hbm_bytes"] * t["launches"]
    for lab, e in by_label.items():
        e["hbm_bytes_per_launch"] = e.pop("bytes") / max(e["launches"], 1)
    json.dump({"by_kernel": traffic, "by_label": by_label,
               "workload": {"nq_total": wl[0], "P": wl[1], "b": wl[2]}, "csrc_sha": csrc_fingerprint(),
               "note": "FETCH_SIZE(KB)*1024*2 (gfx950 half-count correction) + WRITE_SIZE(KB)*1024, averaged per launch; "
                       "separate rocprofv3 --pmc passes"},
              open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
