"""The hot path end to end on one device (recommender.py:145-214 without its Python loops).

    answer sets (CSR) --minhash--> sig, norm2, band keys
                      --bucket sort + pair emit + sort/unique--> candidate pairs
                      --score--> milli + directed edge keys --sort + cut--> per-query top-K

Every stage is a libqrlsh kernel; this module only sequences them and sizes buffers.
"""
import math
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops


def max_candidates(nq):
    """K = round(log_1.5 nq)  (recommender.py:151)"""
    return round(math.log(nq, 1.5))


def select_bands(P, thresh=0.2):
    """Band-count rule of recommender.py:153-163: the largest b with P % b == 0, b % 10 == 0
    and round((1/b)**(1/r), 2) >= thresh.  The reference dies with UnboundLocalError when no
    b qualifies (e.g. P = 128, 256); here that is a ValueError."""
    for b in range(P, 0, -1):
        if P % b == 0 and b % 10 == 0:
            r = P / b
            if round((1 / b) ** (1 / r), 2) >= thresh:
                return b
    raise ValueError("no band count satisfies the reference rule for PERM=%d; pass b explicitly" % P)


@dataclass
class HotPathResult:
    sig: torch.Tensor          # int32 [nq,P], or compact uint16 rows (torch.int16) -- see sig_int32()
    norm2: torch.Tensor        # int64 [nq]
    pairs: torch.Tensor        # int64 [n] sorted unique i<<32|j
    milli: torch.Tensor        # int32 [n]  rint(1000*cos)
    src: torch.Tensor          # int32 [m]  top-K COO, sorted by (src, value desc, dst asc)
    dst: torch.Tensor
    val: torch.Tensor          # int32 [m]  milli of the kept neighbours
    K: int = 0
    b: int = 0
    stats: dict = field(default_factory=dict)

    def sig_int32(self):
        """the signature matrix with the reference's values (int32; -1 = empty answer set)"""
        return ops.sig_to_int32(self.sig)


def query_similarities(offsets, rows, table, b, K, timings=None, compact=None, wide_ids=None, topk="select",
                       validate=True):
    """Whole hot path for the queries described by (offsets, rows) on offsets.device.

    table: ops.PermTable (transposed permutations).  Returns HotPathResult; `timings`, when a
    dict, receives per-phase wall seconds (each phase synchronised -- diagnostic use only).
    validate: check the answer sets first (ops.check_csr: the MinHash kernel gathers by row id unchecked).
    """
    nq = offsets.numel() - 1
    P = table.P
    if P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))
    r = P // b
    stats = {}

    def tick(name, t0):
        if timings is not None:
            torch.cuda.synchronize()
            timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0)
            return time.perf_counter()
        return t0

    t0 = time.perf_counter() if timings is not None else 0.0
    if compact is None:
        compact = ops.can_compact(table)   # uint16 signature rows whenever they are lossless
    sig, norm2, keys = ops.minhash(offsets, rows, table, b=b, want_norm=True, compact=compact, validate=validate)
    t0 = tick("signatures", t0)
    pairs = ops.candidate_pairs(keys, r, stats, sig=sig)
    del keys
    t0 = tick("candidates", t0)
    ib = ops.id_bits_for(nq)
    if topk == "select" and (K > ops.SELECT_MAX_K or pairs.numel() >= (1 << 31)):
        topk = "sort"      # the select form's limits (csrc/pairs.hip: SEL_MAXK, 32-bit run starts)
    if topk == "select":     # reverse edges sorted on j alone, every edge ranks itself in its query's two runs
        milli, rev = ops.score_pairs_rev(sig, norm2, pairs, ib, wide=wide_ids)
        t0 = tick("scoring", t0)
        src, dst, val = ops.topk_select(pairs, milli, rev, K, ib, nq)
    else:                    # "sort": all 2n directed edge keys sorted on (src, value)
        milli, _, edges = ops.score_pairs(sig, norm2, pairs, edge_id_bits=ib, wide=wide_ids)
        t0 = tick("scoring", t0)
        src, dst, val = ops.topk_edges(edges, K, ib)
    tick("topk", t0)
    stats["topk"] = topk
    stats["unique_pairs"] = int(pairs.numel())
    stats["kept_edges"] = int(src.numel())
    return HotPathResult(sig, norm2, pairs, milli, src, dst, val, K, b, stats)


def sims_to_dict(src, dst, val):
    """COO top-K -> the dict recommender.py:206-214 returns:
    {q: {'indexes': int64[<=K], 'values': float64[<=K] descending}}; queries without
    candidates are absent (the consumer tests `j in querySimilarities`, :314)."""
    src, dst, val = ops.to_host(src), ops.to_host(dst), ops.to_host(val)   # (pinned staging: the link's rate)
    out = {}
    if len(src) == 0:
        return out
    cut = np.flatnonzero(np.diff(src)) + 1
    starts = np.concatenate(([0], cut))
    ends = np.concatenate((cut, [len(src)]))
    for s, e in zip(starts, ends):
        out[int(src[s])] = {"indexes": dst[s:e].astype(np.int64), "values": val[s:e].astype(np.float64) / 1000.0}
    return out
