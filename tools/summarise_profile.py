#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC passes)
into the small files committed under profiles/.

usage: summarise_profile.py <tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir>
writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_traffic.csv, profiles/<tag>_hbm_traffic.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:60]
    base = m.group(1).split("::")[-1]
    return base + (m.group(2) or "")


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
    json.dump(traffic, open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
