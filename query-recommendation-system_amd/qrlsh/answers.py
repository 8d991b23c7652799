"""N2 -- answer sets of attribute=value queries on the device (replaces the pandas masks of
Recommender.compute_shingles, recommender.py:68-103).

Host side: dictionary-encode every table column once and pack one bitmap over the table rows
per (feature, value); a query becomes <= nfeat bitmap-row indices.  Device side:
qrlsh_answer_sets_count/fill AND the rows and emit the CSR that qrlsh_minhash consumes.
"""
import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream, _need


class AnswerIndex:
    """Per-(feature, value) bitmaps of a table, resident in HBM."""

    def __init__(self, bitmaps, D, wpr, value_rows, zero_row):
        self.bitmaps = bitmaps        # int32 tensor [nrows_total + 1, wpr] (last row all zero)
        self.D = D
        self.wpr = wpr
        self.value_rows = value_rows  # per feature: (sorted distinct values, first bitmap row)
        self.zero_row = zero_row
        self.nfeat = len(value_rows)


def build_answer_index(columns, device="cuda"):
    """columns: list of 1-D arrays (one per feature, the table's cells as the reference compares
    them: strings).  Returns an AnswerIndex."""
    D = len(columns[0])
    wpr = (D + 31) // 32
    value_rows, maps = [], []
    nrows = 0
    for col in columns:
        col = np.asarray(col)
        vals, inv = np.unique(col, return_inverse=True)
        value_rows.append((vals, nrows))                 # sorted distinct values, first bitmap row
        onehot = np.zeros((len(vals), wpr * 32), dtype=np.uint8)
        onehot[inv, np.arange(D)] = 1
        maps.append(np.packbits(onehot, axis=1, bitorder="little").view(np.uint32))
        nrows += len(vals)
    maps.append(np.zeros((1, wpr), dtype=np.uint32))
    bm = np.ascontiguousarray(np.concatenate(maps, axis=0))
    t = torch.from_numpy(bm.view(np.int32)).to(device)
    return AnswerIndex(t, D, wpr, value_rows, nrows)


def encode_queries(index, queries):
    """queries: (nq, nfeat) array of values, "" = unconstrained (recommender.py:86).
    -> int32 (nq, nfeat) bitmap rows (-1 unconstrained, zero row for values absent from the table)."""
    queries = np.asarray(queries)
    nq = queries.shape[0]
    out = np.full((nq, index.nfeat), -1, dtype=np.int32)
    for f in range(index.nfeat):
        vals, base = index.value_rows[f]
        col = queries[:, f].astype(vals.dtype if vals.dtype.kind in "US" else object)
        want = col != ""
        if not want.any():
            continue
        c = col[want]
        pos = np.searchsorted(vals, c)
        pos_c = np.minimum(pos, len(vals) - 1)
        found = vals[pos_c] == c
        out[want, f] = np.where(found, base + pos_c, index.zero_row).astype(np.int32)
    return out


SLOT = 64   # row ids parked per query by the one-sweep form (AS_SLOT in csrc/answers.hip)


def answer_sets(index, qrows, one_sweep=True):
    """-> (offsets int64 [nq+1], rows int32 [nnz]) on the device; rows ascending per query.

    one_sweep: AND the bitmaps once, parking up to 64 row ids per query, then compact; only when
    some answer set is larger does the exact fill pass run as well (it rewrites every query)."""
    lib = _lib.load()
    dev = index.bitmaps.device
    if not isinstance(qrows, torch.Tensor):
        qrows = torch.from_numpy(np.ascontiguousarray(qrows, dtype=np.int32)).to(dev)
    _need(qrows, torch.int32, "qrows", 2)
    nq, nfeat = qrows.shape
    sizes = torch.empty((nq,), dtype=torch.int32, device=dev)
    offsets = torch.zeros((nq + 1,), dtype=torch.int64, device=dev)
    slots = None
    if one_sweep and nq:
        slots = torch.empty((nq, SLOT), dtype=torch.int32, device=dev)
        _lib.check(lib.qrlsh_answer_sets_sweep(_ptr(index.bitmaps), index.wpr, index.D, _ptr(qrows), nq, nfeat,
                                               _ptr(sizes), _ptr(slots), _stream()))
    else:
        _lib.check(lib.qrlsh_answer_sets_count(_ptr(index.bitmaps), index.wpr, index.D, _ptr(qrows), nq, nfeat,
                                               _ptr(sizes), _stream()))
    torch.cumsum(sizes, dim=0, out=offsets[1:])
    if nq:
        nnz, biggest = torch.stack([offsets[-1], sizes.max().to(torch.int64)]).tolist()
    else:
        nnz, biggest = 0, 0
    rows = torch.empty((nnz,), dtype=torch.int32, device=dev)
    if slots is not None and biggest <= SLOT:
        _lib.check(lib.qrlsh_answer_sets_compact(_ptr(slots), _ptr(offsets), nq, _ptr(rows), _stream()))
    elif nq:
        _lib.check(lib.qrlsh_answer_sets_fill(_ptr(index.bitmaps), index.wpr, index.D, _ptr(qrows), nq, nfeat,
                                              _ptr(offsets), _ptr(rows), _stream()))
    return offsets, rows
