"""Synthetic answer sets generated on the device (bench / test input).

Twin of oracle.synth_csr (bit-identical; tests/test_gpu_parity.py): a pure function of
(seed, query index), so any shard [q0, q0+nq_local) can be generated independently.
"""
import math

import numpy as np
import torch

from . import _lib
from .ops import _ptr, _stream


def poisson_cdf_u32(mean, n=64):
    """32-bit fixed-point CDF thresholds of Poisson(mean): size = #{k : u >= cdf[k]}."""
    out = np.empty(n, dtype=np.uint32)
    term = math.exp(-mean)
    acc = 0.0
    for k in range(n):
        acc += term
        out[k] = min(int(acc * 4294967296.0), 4294967295)
        term *= mean / (k + 1)
    return out


def synth_csr(nq, D, seed=0, cluster=8, mean=16.0, p_replace=0.15, q0=0, nq_local=None, device="cuda"):
    """-> (offsets int64 [nq_local+1], rows int32 [nnz]) on the device."""
    lib = _lib.load()
    if nq_local is None:
        nq_local = nq - q0
    cdf_h = poisson_cdf_u32(mean)
    cdf = torch.from_numpy(cdf_h.view(np.int32)).to(device)
    thr = int(p_replace * (1 << 24))
    sizes = torch.empty((nq_local,), dtype=torch.int32, device=device)
    _lib.check(lib.qrlsh_synth_sizes(seed, q0, nq_local, nq, cluster, D, _ptr(cdf), len(cdf_h), thr, _ptr(sizes),
                                     _stream()))
    offsets = torch.zeros((nq_local + 1,), dtype=torch.int64, device=device)
    torch.cumsum(sizes, dim=0, out=offsets[1:])
    nnz = int(offsets[-1].item()) if nq_local else 0
    rows = torch.empty((nnz,), dtype=torch.int32, device=device)
    _lib.check(lib.qrlsh_synth_fill(seed, q0, nq_local, nq, cluster, D, _ptr(cdf), len(cdf_h), thr, _ptr(offsets),
                                    _ptr(rows), _stream()))
    return offsets, rows
