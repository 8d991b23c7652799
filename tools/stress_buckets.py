#!/usr/bin/env python3
"""Adversarial bucket structures through the partition + overflow pool + finish (all three forms) + block kernel, compared
with the oracle: band keys with planted multiplicities from pairs to tens of thousands of copies, several popular keys
landing in the same part, parts filled to exactly the image size, at partition depths 8 / 11 / 12 / 13.
python tools/stress_buckets.py [seeds]   (development tool, run on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "query-recommendation-system_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from qrlsh import ops, _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def planted_keys(rng, nq, b, groups):
    """[b][nq] keys: random distinct background + for every (size, count) in groups `count` keys with `size` copies per band"""
    keys = rng.integers(1, 1 << 62, size=(b, nq), dtype=np.int64)
    for band in range(b):
        perm = rng.permutation(nq)
        at = 0
        for size, count in groups:
            for _ in range(count):
                if at + size > nq:
                    break
                keys[band, perm[at:at + size]] = rng.integers(1, 1 << 62)
                at += size
    return keys


def np_mix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def same_part_keys(rng, nq, T, sizes, fill_to=None):
    """one band whose popular keys all hash into ONE part of the T-bit partition (mix64's top bits).
    fill_to: instead of sizes, ONE key with exactly as many copies as bring that part to fill_to records."""
    lib = _lib.load()
    keys = rng.integers(1, 1 << 62, size=(1, nq), dtype=np.int64)
    want = None
    perm = rng.permutation(nq)
    at = 0
    if fill_to is not None:
        k = int(rng.integers(1, 1 << 62))
        want = lib.qrlsh_mix64_host(k) >> (64 - T)
        parts = (np_mix64(keys[0].view(np.uint64)) >> np.uint64(64 - T)).astype(np.int64)
        inside = np.flatnonzero(parts == want)
        outside = np.flatnonzero(parts != want)
        size = fill_to - len(inside)
        assert size > 1
        keys[0, outside[:size]] = k          # background records of the part stay, `size` records from elsewhere join it
        return keys
    for size in sizes:
        while True:
            k = int(rng.integers(1, 1 << 62))
            part = lib.qrlsh_mix64_host(k) >> (64 - T)
            if want is None:
                want = part
            if part == want:
                break
        keys[0, perm[at:at + size]] = k
        at += size
    return keys


def check(name, keys, r=4):
    t0 = time.perf_counter()
    stats = {}
    emitted = ops.emit_pairs_any(torch.from_numpy(keys).cuda(), r, stats)
    torch.cuda.synchronize()
    kq = np.ascontiguousarray(keys.T).view(np.uint64)
    want = O.candidates(kq, r)
    n_want = O.emitted_pairs(kq, r)
    got = O.sort_unique(emitted.cpu().numpy().view(np.uint64))
    ok = emitted.numel() == n_want and np.array_equal(got, want)
    print("%s %-64s nq=%-9d b=%d emitted=%-11d unique=%-10d path=%s T=%d  %.1f s" % (
        "OK  " if ok else "FAIL", name, keys.shape[1], keys.shape[0], emitted.numel(), len(want), stats["bucket_path"],
        stats["part_bits"], time.perf_counter() - t0), flush=True)
    del emitted
    torch.cuda.empty_cache()
    return ok


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    O.set_threads(int(os.environ.get("QRLSH_TEST_THREADS", "16")))
    bad = 0
    for seed in range(seeds):
        rng = np.random.default_rng(100 + seed)
        mix = [(2, 20000), (3, 5000), (7, 2000), (40, 300), (300, 40), (1700, 6), (4100, 2), (9000, 1)]
        bad += not check("T=8 mixed multiplicities", planted_keys(rng, 900_000, 3, mix))
        bad += not check("T=11 (small form, separate counters)", planted_keys(rng, 6_000_000, 2, mix + [(25000, 1)]))
        bad += not check("T=12 (packed counters)", planted_keys(rng, 12_000_000, 2, mix + [(30000, 1)]))
        bad += not check("T=13 (packed counters)", planted_keys(rng, 20_000_000, 1, mix + [(12000, 2)]))
        # several popular keys in ONE part: region prefix + many spilled runs + several block pairs
        bad += not check("T=12, five popular keys in one part", same_part_keys(rng, 12_000_000, 12, [3000, 2500, 900, 5000, 1200]))
        bad += not check("T=11, four popular keys in one part", same_part_keys(rng, 6_000_000, 11, [2000, 2100, 1500, 7000]))
        bad += not check("T=8, popular keys in one part", same_part_keys(rng, 1_000_000, 8, [4000, 2500, 800]))
        # a part filled to exactly the image / one short of it / one beyond (the background records of the part are counted)
        for fill in (4094, 4095, 4096, 4097, 6143, 6144, 6145):
            bad += not check("T=12, a part of exactly %d records" % fill, same_part_keys(rng, 12_000_000, 12, None, fill_to=fill))
        for fill in (4095, 4096, 4097):
            bad += not check("T=11, a part of exactly %d records" % fill, same_part_keys(rng, 6_000_000, 11, None, fill_to=fill))
        for fill in (6143, 6144, 6145):
            bad += not check("T=8, a part of exactly %d records" % fill, same_part_keys(rng, 1_000_000, 8, None, fill_to=fill))
    print("stress: %d failing case(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
