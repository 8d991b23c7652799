#!/usr/bin/env python3
"""Condense the per-pass rocprofv3 --pmc outputs of tools/minhash_counters.sh into one small table:
usage: summarise_counters.py <tag> [kernel-name-substring] [stage]  ->  profiles/<tag>_<stage>_counters.csv
(stage "minhash" reads gpurun_out/<tag>_mh_*, any other stage gpurun_out/<tag>_<stage>_* of tools/stage_counters.sh)
Every row = one counter, averaged over the launches of the kernel in its pass, with the kernel's mean
duration in that pass (from the pass's own kernel trace) beside it."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "minhash"
    stage = sys.argv[3] if len(sys.argv) > 3 else "minhash"
    rows = []
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + ("_mh_*" if stage == "minhash" else "_%s_*" % stage)))):
        if not os.path.isdir(d):
            continue
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        dur = []
        if kt:
            for r in csv.DictReader(open(kt[0])):
                if want in r["Kernel_Name"]:
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        acc = defaultdict(list)
        if cc:
            for r in csv.DictReader(open(cc[0])):
                if want in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, vals in sorted(acc.items()):
            rows.append([os.path.basename(d), name, len(vals), "%.6g" % (sum(vals) / len(vals)),
                         "%.2f" % (sum(dur) / len(dur)) if dur else ""])
    out = os.path.join(ROOT, "profiles", "%s_%s_counters.csv" % (tag, stage))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["pass", "counter", "launches", "mean_value_per_launch", "kernel_mean_us_in_this_pass"])
        w.writerows(rows)
    print("wrote", out, "(%d rows)" % len(rows))


if __name__ == "__main__":
    main()
