"""Drop-in for the reference's lsh.py: same class, same call surface, GPU underneath.

    from lsh import LSH
    lsh = LSH(b)
    for sig in signatures: lsh.compute_buckets(sig)      # lsh.py:31 of the reference
    candidates = lsh.get_candidates(signatures)          # lsh.py:40 -> set[(i, j)], i < j

Differences from the reference, all deliberate:
  * `buckets` / `counter` are per instance.  In the reference they are class attributes
    (lsh.py:9-10) and a second LSH() in one process keeps appending to the first one's
    dicts; that leak is not reproduced.
  * compute_buckets only buffers the signature; the GPU work (band keys -> radix sort ->
    segment -> pair emit -> unique) runs once, in get_candidates.  Batch twins
    compute_buckets_batch / get_candidates_array avoid per-query Python calls and Python
    sets at scale.
  * `buckets` (list of b dicts "v0,v1,.." -> [ids]) is materialised lazily on first access,
    for code that inspects it; the hot path never builds it.
"""
import numpy as np
import torch

import qrlsh
from qrlsh import ops


class LSH:

    def __init__(self, b, device="cuda"):
        self.b = b
        self.counter = 0
        self.device = device
        self._host_sigs = []      # buffered compute_buckets() rows
        self._dev_sigs = []       # buffered compute_buckets_batch() tensors
        self._buckets = None

    # -- reference surface ---------------------------------------------------
    def make_subvecs(self, signature):
        """(b, r) int16 view of one signature (lsh.py:17-28)."""
        signature = np.asarray(signature)
        l = len(signature)
        assert l % self.b == 0
        r = l // self.b
        return signature.reshape(self.b, r).astype('int16')

    def compute_buckets(self, signature):
        """Register one signature under the next query id (lsh.py:31-38)."""
        signature = np.asarray(signature)
        assert len(signature) % self.b == 0
        self._host_sigs.append(signature)
        self.counter += 1
        self._buckets = None

    def get_candidates(self, signatures=None):
        """set of (i, j), i < j, sharing at least one non-empty band (lsh.py:40-55).
        `signatures` is accepted and ignored, as in the reference."""
        pairs = ops.to_host(self.get_candidates_array())
        i = (pairs >> 32).astype(np.int64)
        j = (pairs & 0xFFFFFFFF).astype(np.int64)
        return set(zip(i.tolist(), j.tolist()))

    @property
    def buckets(self):
        if self._buckets is None:
            self._buckets = self._materialise_buckets()
        return self._buckets

    # -- batch twins -----------------------------------------------------------
    def compute_buckets_batch(self, signatures):
        """Register many signatures at once: int tensor / array [n, P]; ids continue from
        `counter`."""
        if isinstance(signatures, torch.Tensor):
            t = signatures
            if t.dtype != torch.int32:
                t = t.to(torch.int64).bitwise_and(0xFFFFFFFF).to(torch.int32) if t.dtype == torch.int64 else t.to(torch.int32)
            t = t.to(self.device).contiguous()
        else:
            t = self._to_device(np.asarray(signatures))
        assert t.shape[1] % self.b == 0
        self._flush_host()
        self._dev_sigs.append(t)
        self.counter += t.shape[0]
        self._buckets = None

    def signatures_tensor(self):
        """All registered signatures as one int32 [n, P] device tensor (low 32 bits)."""
        self._flush_host()
        if not self._dev_sigs:
            return torch.empty((0, self.b), dtype=torch.int32, device=self.device)
        if len(self._dev_sigs) > 1:
            self._dev_sigs = [torch.cat(self._dev_sigs, dim=0)]
        return self._dev_sigs[0]

    def get_candidates_array(self, stats=None):
        """Sorted unique int64 device tensor of i << 32 | j (i < j)."""
        sig = self.signatures_tensor()
        n, P = sig.shape
        if n == 0:
            return torch.empty((0,), dtype=torch.int64, device=self.device)
        keys = ops.band_keys(sig, self.b)
        return ops.candidate_pairs(keys, P // self.b, stats, sig=sig)

    # -- internals ---------------------------------------------------------------
    def _to_device(self, a):
        if a.ndim == 1:
            a = a[None, :]
        # only the low 16 bits of every value take part in a bucket key (lsh.py:28); keep 32
        a32 = (a.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        return torch.from_numpy(np.ascontiguousarray(a32)).to(self.device)

    def _flush_host(self):
        if self._host_sigs:
            self._dev_sigs.append(self._to_device(np.stack(self._host_sigs)))
            self._host_sigs = []

    def _materialise_buckets(self):
        """The reference's `buckets` attribute (list of b dicts "v0,v1,.." -> [ids]), rebuilt on
        request from the registered signatures; insertion order = first appearance by id."""
        sig = ops.to_host(ops.sig_to_int32(self.signatures_tensor()))
        out = [dict() for _ in range(self.b)]
        n, P = sig.shape
        if n == 0:
            return out
        r = P // self.b
        sub = (sig.astype(np.int64) & 0xFFFF).astype(np.uint16).view(np.int16).reshape(n, self.b, r)
        for band in range(self.b):
            d = out[band]
            for q in range(n):
                d.setdefault(",".join(str(int(v)) for v in sub[q, band]), []).append(q)
        return out
