"""What the GPU box's host offers the CPU baseline: cores by affinity, cgroup quota, and the oracle's candidate
stage (the part that was serial) timed at several OpenMP thread counts.  python tools/host_probe.py [nq]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cgroup cpu.max: n/a", e)
print("loadavg", open("/proc/loadavg").read().strip())
D, P, b = 32768, 128, 32
off, rows = O.synth_csr(nq, D, seed=0)
perms = O.legacy_permutations(42, P, D)
for th in (16, 32, 64, 128, len(os.sched_getaffinity(0))):
    if th > len(os.sched_getaffinity(0)):
        continue
    O.set_threads(th)
    t = time.time(); sig = O.minhash(off, rows, perms); t1 = time.time() - t
    keys = O.band_keys(sig, b)
    t = time.time(); pairs = O.candidates(keys, P // b); t2 = time.time() - t
    t = time.time(); milli = O.score_pairs(sig, pairs, mode=1); t3 = time.time() - t
    t = time.time(); O.topk(pairs, milli, O.max_candidates(nq)); t4 = time.time() - t
    print("threads %3d: minhash %.2f s  candidates %.2f s  score %.2f s  topk %.2f s  (%d pairs)" % (th, t1, t2, t3, t4, len(pairs)), flush=True)
