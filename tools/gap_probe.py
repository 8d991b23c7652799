#!/usr/bin/env python3
"""How much of a step is the GPU idle?  Reads a rocprofv3 --kernel-trace CSV of tools/step_time.py (or bench.py) and
prints, for the steady-state steps, the union of kernel-busy time, the idle gaps and the largest gaps with the kernels
on either side.  usage: gap_probe.py <kernel_trace.csv> [steps_to_skip]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
# steps begin with the MinHash kernel
starts = [k for k, r in enumerate(rows) if "minhash_group_kernel" in r[2]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
starts = starts[skip:]
tot_busy = tot_span = 0
gaps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    cur_s, cur_e = seg[0][0], seg[0][1]
    busy = 0
    prev = seg[0]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev[2], n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = (s, e, n)
    busy += cur_e - cur_s
    gaps.append((t1 - cur_e, prev[2], "next step's minhash"))
    tot_busy += busy
    tot_span += t1 - t0
n = len(starts) - 1
print("steps %d: span %.3f ms, GPU busy (union of kernels) %.3f ms, idle %.3f ms per step" % (n, tot_span / n / 1e6, tot_busy / n / 1e6, (tot_span - tot_busy) / n / 1e6))
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg.setdefault(k, [0, 0])
    agg[k][0] += g
    agg[k][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%8.1f us per step (%4.1f us x %.1f)  after %-44s before %s" % (g / n / 1e3, g / c / 1e3, c / n, a[-44:], b[-44:]))
