"""Thin torch-tensor wrappers over the C ABI (include/qrlsh.h).

Tensors are device buffers only: every function checks device / dtype / contiguity, passes
raw pointers and the current HIP stream to libqrlsh, and returns tensors.  64-bit unsigned
words (band keys, pairs, edge keys) are carried in torch.int64 tensors (bit patterns).
"""
import ctypes

import numpy as np
import torch

from . import _lib

_vp = ctypes.c_void_p


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    return _vp(t.data_ptr())


def _need(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dimensions" % (name, ndim))
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def to_host(t):
    """device tensor -> numpy array through PINNED host memory (torch's caching host allocator: the first copy of a
    size pays for the pinning, later ones reuse the block once the array it backed is gone).  A plain `.cpu()` lands
    in pageable memory at ~6 GB/s -- 139 ms for the 786 MB of top-K rows of a 10 M-query step, 7 x the step itself;
    the pinned copy runs at the link's rate.  The array owns its buffer (nothing is shared between calls)."""
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()


def id_bits_for(nq):
    """bits needed for a query id in [0, nq)"""
    return max(1, int(nq - 1).bit_length()) if nq > 1 else 1


# ---------------------------------------------------------------------------
# permutation table (host side; recommender.py:120)
# ---------------------------------------------------------------------------
def legacy_permutations(P, D, seed=None, rng=None):
    """The P consecutive np.random.permutation(D) draws of recommender.py:120.

    seed given -> RandomState(seed) (the same stream np.random.seed(seed) selects);
    otherwise `rng` (default: the global legacy np.random, consumed exactly like the
    reference consumes it).  Returns int32 [P][D].
    """
    if seed is not None:
        rng = np.random.RandomState(seed)
    elif rng is None:
        rng = np.random
    out = np.empty((P, D), dtype=np.int32)
    for p in range(P):
        out[p] = rng.permutation(D)
    return out


class PermTable:
    """Transposed permutation table resident in HBM: tab[d][0..P) = perm_0[d] .. perm_{P-1}[d]."""

    def __init__(self, tab, code, P, P_stride, D):
        self.tab, self.code, self.P, self.P_stride, self.D = tab, code, P, P_stride, D


def perm_table(perms, device="cuda", force_i32=False):
    """Upload int [P][D] permutations as the transposed [D][P_stride] table the MinHash
    kernel gathers from: uint16 when D <= 65536 (half the gather bytes), else int32."""
    perms = np.asarray(perms)
    P, D = perms.shape
    if D <= 65536 and not force_i32:
        stride = (P + 7) // 8 * 8
        tab = np.full((D, stride), 0xFFFF, dtype=np.uint16)
        tab[:, :P] = perms.T
        t = torch.from_numpy(tab.view(np.int16)).to(device)
        code = _lib.PERM_U16
    else:
        stride = (P + 3) // 4 * 4
        tab = np.full((D, stride), 0x7FFFFFFF, dtype=np.int32)
        tab[:, :P] = perms.T
        t = torch.from_numpy(tab).to(device)
        code = _lib.PERM_I32
    return PermTable(t, code, P, stride, D)


# ---------------------------------------------------------------------------
# a1 / a2
# ---------------------------------------------------------------------------
def can_compact(table):
    """compact uint16 signature rows are lossless iff every value < 65535 (0xFFFF encodes -1)"""
    return table.D <= 65535 and table.code == _lib.PERM_U16


def sig_to_int32(sig):
    """signature tensor (int32, or compact uint16 rows carried as torch.int16) -> int32"""
    if sig.dtype == torch.int32:
        return sig
    v = sig.to(torch.int32) & 0xFFFF
    return torch.where(v == 0xFFFF, torch.full_like(v, -1), v)


def check_csr(offsets, rows, D):
    """raise ValueError unless (offsets, rows) is a well-formed CSR of row ids in [0, D): the kernel gathers
    table rows by these ids without a bounds check (one small kernel + one read-back)"""
    lib = _lib.load()
    nq = offsets.numel() - 1
    flags = torch.zeros((1,), dtype=torch.int32, device=offsets.device)
    _lib.check(lib.qrlsh_check_csr(_ptr(offsets), _ptr(rows), nq, rows.numel(), int(D), _ptr(flags), _stream()))
    f = int(flags.item())
    if f:
        what = [m for bit, m in ((1, "offsets[0] != 0 or offsets[-1] != len(rows)"), (2, "offsets decrease"),
                                 (4, "row id outside [0, %d)" % D)) if f & bit]
        raise ValueError("malformed answer sets: " + "; ".join(what))


def minhash(offsets, rows, table, b=None, want_norm=True, compact=False, validate=True, out=None):
    """sig[q][p] = min over the answer set of perm_p (recommender.py:105-143), -1 if empty.
    Returns (sig [nq,P], norm2 int64 [nq] | None, keys int64 [b,nq] | None).  sig is int32, or
    with compact=True (needs can_compact(table)) the uint16 rows (torch.int16 bit patterns,
    0xFFFF = -1) that qrlsh_score_pairs reads at half the bytes; see sig_to_int32.
    validate: check the CSR first (check_csr); callers that built it with the library's own kernels
    (answer_sets, synth_csr) may pass False.
    out=(sig, norm2, keys): write into these contiguous tensors (slices of larger buffers) instead of new ones."""
    lib = _lib.load()
    _need(offsets, torch.int64, "offsets", 1)
    _need(rows, torch.int32, "rows", 1)
    nq = offsets.numel() - 1
    if nq < 0:
        raise ValueError("offsets must have nq+1 entries")
    if validate and nq > 0:
        check_csr(offsets, rows, table.D)
    P = table.P
    dev = offsets.device
    if b is not None and P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))  # lsh.py:20
    if compact and not can_compact(table):
        raise ValueError("compact signatures need D <= 65535")
    if out is not None:
        sig, norm2, keys = out
        _need(sig, torch.int16 if compact else torch.int32, "out sig", 2)
        if tuple(sig.shape) != (nq, P) or (norm2 is not None and norm2.numel() != nq) or \
                (keys is not None and (b is None or tuple(keys.shape) != (b, nq))):
            raise ValueError("out buffers do not match (nq=%d, P=%d, b=%s)" % (nq, P, b))
        if norm2 is not None:
            _need(norm2, torch.int64, "out norm2", 1)
        if keys is not None:
            _need(keys, torch.int64, "out keys", 2)
        if (b is not None) != (keys is not None) or want_norm != (norm2 is not None):
            raise ValueError("out must hold exactly the outputs asked for")
    else:
        sig = torch.empty((nq, P), dtype=torch.int16 if compact else torch.int32, device=dev)
        norm2 = torch.empty((nq,), dtype=torch.int64, device=dev) if want_norm else None
        keys = torch.empty((b, nq), dtype=torch.int64, device=dev) if b is not None else None
    _lib.check(lib.qrlsh_minhash(_ptr(offsets), _ptr(rows), nq, _ptr(table.tab), table.code, P, table.P_stride,
                                 table.D, None if compact else _ptr(sig), _ptr(sig) if compact else None,
                                 _ptr(norm2), _ptr(keys), b if b is not None else 0, _stream()))
    return sig, norm2, keys


def band_keys(sig, b, want_norm=False):
    """Band keys of an int32 signature matrix (lsh.py:17-38) -> int64 [b,nq] (+ norm2)."""
    lib = _lib.load()
    _need(sig, torch.int32, "sig", 2)
    nq, P = sig.shape
    if b <= 0 or P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))  # lsh.py:20
    keys = torch.empty((b, nq), dtype=torch.int64, device=sig.device)
    norm2 = torch.empty((nq,), dtype=torch.int64, device=sig.device) if want_norm else None
    _lib.check(lib.qrlsh_band_keys(_ptr(sig), nq, P, b, _ptr(keys), _ptr(norm2), _stream()))
    return (keys, norm2) if want_norm else keys


def row_norms(sig):
    lib = _lib.load()
    _need(sig, torch.int32, "sig", 2)
    nq, P = sig.shape
    out = torch.empty((nq,), dtype=torch.int64, device=sig.device)
    _lib.check(lib.qrlsh_row_norms(_ptr(sig), nq, P, _ptr(out), _stream()))
    return out


# ---------------------------------------------------------------------------
# radix sort
# ---------------------------------------------------------------------------
def sort_u64(keys, vals=None, bit_lo=0, bit_hi=64, mix=False, iota=False, fold=0, owner_shard=0, host_shard=0):
    """Stable LSD radix sort of each row of keys (int64 bit patterns, unsigned order) over
    bits [bit_lo, bit_hi) (of mix64(key) when mix).  `keys` (and vals) are consumed as one
    of the two ping-pong buffers.  Returns (sorted_keys, sorted_vals | None)."""
    lib = _lib.load()
    _need(keys, torch.int64, "keys")
    k2 = keys if keys.dim() == 2 else keys.view(1, -1)
    nbatch, n = k2.shape
    if iota and vals is None:
        vals = torch.empty((nbatch, n), dtype=torch.int32, device=keys.device)
    if vals is not None:
        _need(vals, torch.int32, "vals")
        if vals.numel() != k2.numel():
            raise ValueError("vals must match keys")
    kb = torch.empty_like(k2)
    vb = torch.empty_like(vals) if vals is not None else None
    nbytes = lib.qrlsh_sort_workspace_bytes(n, nbatch)
    ws = _ws(nbytes, keys.device)
    flags = (_lib.SORT_MIX if mix else 0) | (_lib.SORT_IOTA if iota else 0)
    aux = 0
    if fold:
        flags |= _lib.SORT_FOLD
        aux = int(fold)
    if owner_shard:   # one pass: digit = (key >> bit_lo) // owner_shard  (rank owning that id)
        flags |= _lib.SORT_OWNER
        aux = int(owner_shard)
        bit_hi = bit_lo + 1
    if host_shard:    # one pass over pair words: digit = the rank that scores the pair (common.h: qr_pair_host)
        flags |= _lib.SORT_HOST
        aux = int(host_shard)
        bit_lo, bit_hi = 0, 1
    rc = _lib.check(lib.qrlsh_sort_u64(_ptr(k2), _ptr(kb), _ptr(vals), _ptr(vb), n, nbatch, bit_lo, bit_hi, flags, aux,
                                       _ptr(ws), ws.numel(), _stream()))
    ko = kb if rc == 1 else k2
    vo = (vb if rc == 1 else vals) if vals is not None else None
    if keys.dim() == 1:
        ko = ko.view(-1)
        vo = vo.view(-1) if vo is not None else None
    elif vo is not None:
        vo = vo.view(nbatch, n)
    return ko, vo


def hash_bits_for(n):
    """Bits of mix64(key) the grouping sort orders by: enough that a run of equal hash bits
    holds ~1/8 foreign keys on average (n / 2^bits <= 1/8), in whole 8-bit passes."""
    need = max(1, int(n - 1).bit_length()) + 3 if n > 1 else 8
    return min(32, max(8, (need + 7) // 8 * 8))


def owner_bounds(words, bit_lo, shard, world):
    """split points (device int64 [world + 1]) of words already grouped by owner = (word >> bit_lo) // shard;
    bit_lo = -1: pair words grouped by their scoring rank (sort_u64(host_shard=...))"""
    lib = _lib.load()
    _need(words, torch.int64, "words", 1)
    bounds = torch.empty((world + 1,), dtype=torch.int64, device=words.device)
    _lib.check(lib.qrlsh_owner_bounds(_ptr(words), words.numel(), bit_lo, shard, world, _ptr(bounds), _stream()))
    return bounds


def owner_sizes(words, bit_lo, shard, world):
    """per-rank counts of words already grouped by owner (a host list: one read-back)"""
    bl = owner_bounds(words, bit_lo, shard, world).tolist()
    return [bl[g + 1] - bl[g] for g in range(world)]


def bucket_sort(keys, hash_bits=None):
    """Group equal band keys (replaces the dict buckets of lsh.py:31-38): per band, sort
    (key, query id) by the top hash_bits bits of mix64(key).  keys int64 [b,nq] (consumed).
    Returns (sorted_keys [b,nq], sorted_ids int32 [b,nq])."""
    _need(keys, torch.int64, "keys", 2)
    hb = hash_bits if hash_bits is not None else hash_bits_for(keys.shape[1])
    return sort_u64(keys, None, 64 - hb, 64, mix=True, iota=True)


# ---------------------------------------------------------------------------
# a3: pairs
# ---------------------------------------------------------------------------
def emit_pairs(sorted_keys, sorted_ids, r, hash_bits=None):
    """All (i<j) pairs of every non-empty bucket with > 1 member, every band
    (lsh.py:42-53), as int64 i<<32|j, duplicates across bands included."""
    lib = _lib.load()
    _need(sorted_keys, torch.int64, "sorted_keys", 2)
    _need(sorted_ids, torch.int32, "sorted_ids", 2)
    b, nq = sorted_keys.shape
    dev = sorted_keys.device
    hb = hash_bits if hash_bits is not None else hash_bits_for(nq)
    ws = _ws(lib.qrlsh_pairs_workspace_bytes(nq, b), dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_pairs_count(_ptr(sorted_keys), nq, b, r, hb, _ptr(ws), ws.numel(), _ptr(total), _stream()))
    n = int(total.item())
    pairs = torch.empty((n,), dtype=torch.int64, device=dev)
    if n:
        _lib.check(lib.qrlsh_pairs_fill(_ptr(sorted_keys), _ptr(sorted_ids), nq, b, r, hb, _ptr(ws), _ptr(pairs),
                                        _stream()))
    return pairs


def unique_sorted(a):
    lib = _lib.load()
    _need(a, torch.int64, "a", 1)
    n = a.numel()
    ws = _ws(lib.qrlsh_compact_workspace_bytes(n), a.device)
    total = torch.zeros(1, dtype=torch.int64, device=a.device)
    _lib.check(lib.qrlsh_unique_count(_ptr(a), n, _ptr(ws), ws.numel(), _ptr(total), _stream()))
    m = int(total.item())
    out = torch.empty((m,), dtype=torch.int64, device=a.device)
    if m:
        _lib.check(lib.qrlsh_unique_fill(_ptr(a), n, _ptr(ws), _ptr(out), _stream()))
    return out


def sort_pairs(pairs, nq):
    """sort i<<32|j words: LSD over j's bits, then i's bits"""
    ib = id_bits_for(nq)
    p, _ = sort_u64(pairs, None, 0, 2 * ib, fold=ib)   # digits of i << ib | j: ceil(2 ib / 8) passes
    return p


def row_group_bits(id_bits, words_per_row=0.0):
    """low bits of i the grouping sort may ignore (rows of 2^g consecutive i, finished in LDS) when that
    saves its last 8-bit pass: the id bits left over above a multiple of 8, if at most 4 -- and only while
    the grouped rows stay short (words_per_row = emitted pairs per query: rows of 2^g queries beyond ~1000
    words would all go through the long-row kernel, or overflow it)"""
    g = id_bits % 8
    if not (id_bits > 8 and 0 < g <= 4):
        return 0
    return g if words_per_row * (1 << g) <= 1024 else 0


def group_pairs_by_i(pairs, nq, group_bits=0):
    """pairs i<<32|j ordered by i >> group_bits only (arbitrary order inside a row)"""
    p, _ = sort_u64(pairs, None, 32 + group_bits, 32 + id_bits_for(nq))
    return p


def row_unique(grouped, group_bits=0, id_bits=32):
    """Sorted unique pairs from pairs grouped by i >> group_bits (group_pairs_by_i); None when a row is
    too long even for the one-row-per-workgroup kernel (heavily skewed data) -- use
    unique_sorted(sort_pairs(...)) then."""
    lib = _lib.load()
    _need(grouped, torch.int64, "grouped", 1)
    n = grouped.numel()
    if n == 0:
        return grouped
    dev = grouped.device
    tmp = torch.empty_like(grouped)
    ws = _ws(lib.qrlsh_row_unique_workspace_bytes(n), dev)
    tot = torch.empty(2, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_row_unique_count(_ptr(grouped), n, int(group_bits), int(id_bits), _ptr(tmp), _ptr(ws),
                                          ws.numel(), _ptr(tot), _stream()))
    total, overflow = tot.tolist()
    if overflow:
        return None
    out = torch.empty((total,), dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_row_unique_fill(_ptr(tmp), n, _ptr(ws), _ptr(out), _stream()))
    return out


def region_group_bits(id_bits, nids, words_per_query=0.0):
    """group bits of the region form of the de-duplication (regions of 2^g consecutive queries finished by one
    workgroup each): as many as fit the 32-bit (i's low bits, j) value, at most 8, fewer while a region would
    hold more than ~16 K words; None when fewer than 3 remain (single-i rows: use row_unique)"""
    g = min(8, 32 - id_bits, id_bits)    # (no more than the id has: everything is one region then)
    if g + id_bits == 32 and nids >= (1 << id_bits):
        g -= 1
    while g > 0 and words_per_query * (1 << g) > 16384:
        g -= 1
    return g if g >= 3 else None


def region_unique(grouped, group_bits, id_bits, nids):
    """Sorted unique pairs from pairs grouped by i >> group_bits (group_pairs_by_i), one workgroup per region of
    2^group_bits queries; None when a region holds more than ~11 K distinct pairs (use the general path)."""
    lib = _lib.load()
    _need(grouped, torch.int64, "grouped", 1)
    n = grouped.numel()
    if n == 0:
        return grouped
    dev = grouped.device
    tmp = torch.empty_like(grouped)
    ws = _ws(lib.qrlsh_region_unique_workspace_bytes(nids, group_bits), dev)
    tot = torch.empty(2, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_region_unique_count(_ptr(grouped), n, int(group_bits), int(id_bits), int(nids), _ptr(tmp), _ptr(ws),
                                             ws.numel(), _ptr(tot), _stream()))
    total, overflow = tot.tolist()
    if overflow:
        return None
    out = torch.empty((total,), dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_region_unique_fill(_ptr(tmp), n, int(group_bits), int(nids), _ptr(ws), _ptr(out), _stream()))
    return out


REGION_MIN_GROUP_BITS = 6               # below: the general sort + unique instead of the region form (see unique_pairs)
REGION_SCATTER_MAX_BYTES = 48 << 30    # the fixed-region grouping is skipped when its buffers would exceed this


def region_unique_scattered(emitted, group_bits, id_bits, nids, words_per_query=0.0):
    """Sorted unique pairs straight from the emitted words: dealt into fixed regions of 2^group_bits queries by two
    histogram-free partition steps (qrlsh_pair_regions_scatter: nothing inside a region is ordered -- the region
    finish does not need it), then the region finish on those regions.  -> (pairs, "") or (None, why): "cap" when a
    region outgrew its capacity or the buffers would be too large (the caller then groups by sorting), "distinct" when
    a region holds more distinct pairs than the finish's LDS set (sorting by region would meet the same: the caller
    sorts everything)."""
    lib = _lib.load()
    _need(emitted, torch.int64, "emitted", 1)
    n = emitted.numel()
    if n == 0:
        return emitted, ""
    dev = emitted.device
    wpq = float(words_per_query)
    words = lib.qrlsh_pair_regions_words(n, nids, group_bits, wpq)
    twords = lib.qrlsh_pair_regions_tmp_words(n, nids, group_bits, wpq)
    if words == 0 or n >= (1 << 32) or (2 * words + twords) * 8 > REGION_SCATTER_MAX_BYTES:
        return None, "cap"          # (words == 0: more than 65536 regions -- two levels of 256 digits do not reach)
    cap = lib.qrlsh_pair_regions_cap(n, nids, group_bits, wpq)
    nreg = lib.qrlsh_pair_regions_count(n, nids, group_bits, wpq)
    try:      # fixed regions trade memory for passes: when the device is short of it, group by sorting instead
        regions = torch.empty((words,), dtype=torch.int64, device=dev)
        tmpr = torch.empty((twords,), dtype=torch.int64, device=dev) if twords else None
        tmp = torch.empty_like(regions)
    except torch.cuda.OutOfMemoryError:
        regions = tmpr = tmp = None
        torch.cuda.empty_cache()
        return None, "cap"
    counts = torch.empty((nreg + 256,), dtype=torch.int32, device=dev)
    ovf = torch.empty((1,), dtype=torch.int32, device=dev)
    _lib.check(lib.qrlsh_pair_regions_scatter(_ptr(emitted), n, int(group_bits), int(nids), wpq, _ptr(tmpr), _ptr(regions),
                                              _ptr(counts), _ptr(ovf), _stream()))
    del tmpr
    ws = _ws(lib.qrlsh_region_unique_workspace_bytes(nids, group_bits), dev)
    tot = torch.empty(2, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_region_unique_count_regions(_ptr(regions), _ptr(counts), cap, n, int(group_bits), int(id_bits),
                                                     int(nids), _ptr(tmp), _ptr(ws), ws.numel(), _ptr(tot), _stream()))
    total, overflow = tot.tolist()          # (the stream is in order: the scatter's flag is final by now as well)
    if int(ovf.item()):
        return None, "cap"
    if overflow:
        return None, "distinct"
    del regions
    out = torch.empty((total,), dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_region_unique_fill(_ptr(tmp), n, int(group_bits), int(nids), _ptr(ws), _ptr(out), _stream()))
    return out, ""


def unique_pairs(emitted, nq, stats=None, words_per_query=None, scatter=True):
    """the Python set of lsh.py:41,53: sorted unique words of the emitted pairs (consumed).
    words_per_query: emitted words per query id of the id space (default: emitted / nq).
    scatter: group by the histogram-free fixed-region partition (default); False: by the stable radix sort."""
    ib = id_bits_for(nq)
    wpq = emitted.numel() / max(nq, 1) if words_per_query is None else words_per_query
    g = region_group_bits(ib, nq, wpq)
    pairs = None
    if g is not None and g < REGION_MIN_GROUP_BITS:
        # ids of 27 bits and more leave at most 5 group bits: regions of 32 queries, a workgroup per ~2 000 words, three
        # grouping passes before it -- the plain 7-pass sort + unique is faster there (configs[4], one rank: 4 ms against
        # 9.6 for 93 M words, 39 against 70.6 for the 897 M it really receives, where the regions would give up anyway)
        grouped, g = emitted, 0
    elif g is not None:
        path, why = "regions-in-lds", "cap"
        if scatter:
            pairs, why = region_unique_scattered(emitted, g, ib, nq, wpq)
            path = "regions-in-lds (scattered)"
        if pairs is None and why == "distinct":
            grouped = emitted                       # straight to the full sort below
        elif pairs is None:
            grouped = group_pairs_by_i(emitted, nq, g)
            pairs = region_unique(grouped, g, ib, nq)
            path = "regions-in-lds"
    else:
        g = row_group_bits(ib, wpq)
        grouped = group_pairs_by_i(emitted, nq, g)
        pairs = row_unique(grouped, g, ib)
        path = "rows-in-lds"
    if stats is not None:
        stats["dedup_path"] = path if pairs is not None else "full-sort"
        stats["group_bits"] = g
    if pairs is None:
        pairs = unique_sorted(sort_pairs(grouped, nq))
    return pairs


def part_bits_for(n):
    """T = bits of the hash partition of the fast bucket path: parts of <= ~4400 records on
    average (the LDS image holds 6144; the room above the mean is for popular keys -- with parts of ~4900
    on average the 10 M-query bench workload overflows a part), at least 8, at most 16."""
    t = 8
    while t < 16 and n > 4400 * (1 << t):
        t += 1
    return t


_EMIT_HINT = {}   # (nq, b, r) -> pairs emitted by the last call of that shape: the next call's capacity guess


def emit_pairs_fast(keys, r, part_bits=None, one_pass=True, capacity=None, chunks=None):
    """emit_pairs for unsorted band-major keys through the partition + LDS-finish path.
    Returns the pairs tensor, or None when a part overflowed the LDS image (skewed data).
    one_pass: the parts reserve their output ranges on a device cursor (qrlsh_bucket_pairs_emit), so
    no count pass runs; the output buffer is sized by `capacity` (default: 1.25 x what the last call
    of this shape emitted, else 24 pairs per query) and the call is repeated once, exactly sized,
    if that was too small.  one_pass=False is the count-then-fill form.
    chunks=(world, nb, nql[, stride]): `keys` is the [world][nb][nql] buffer a band-partitioned all-to-all delivers
    (band t, query q at [q // nql][t][q % nql]); it is read in place (one-pass form).  stride (default nb * nql):
    words between two ranks' chunks -- nb of each rank's b bands, read out of a [world][b][nql] buffer."""
    lib = _lib.load()
    if chunks is not None:
        world, b, nql = chunks[:3]
        stride = chunks[3] if len(chunks) > 3 else b * nql     # words between the chunks of two ranks
        nq = world * nql
        _need(keys, torch.int64, "keys")
        if stride < b * nql or keys.numel() < (world - 1) * stride + b * nql or not one_pass:
            raise ValueError("chunked keys: need (world - 1) * stride + nb * nql words and the one-pass form")
        layout = (nql, stride, nql)
    else:
        _need(keys, torch.int64, "keys", 2)
        b, nq = keys.shape
        layout = (0, 0, 0)
    dev = keys.device
    T = part_bits if part_bits is not None else part_bits_for(nq)
    if nq > 6144 * (1 << T):
        return None  # cannot fit even with the finest partition
    words = lib.qrlsh_bucket_part_words(nq, b, T) if one_pass else b * nq
    pk = torch.empty((words,), dtype=torch.int64, device=dev)
    pid = torch.empty((words,), dtype=torch.int32, device=dev)
    twords = (lib.qrlsh_bucket_tmp_words(nq, b, T) if one_pass else b * nq) if T > 8 else 0
    tk = torch.empty((twords,), dtype=torch.int64, device=dev) if T > 8 else None
    tid = torch.empty((twords,), dtype=torch.int32, device=dev) if T > 8 else None
    ws = _ws(lib.qrlsh_bucket_workspace_bytes(nq, b, T), dev)
    tot = torch.zeros(2, dtype=torch.int64, device=dev)
    if one_pass:
        shape = (nq, b, r)
        if capacity is None:
            capacity = _EMIT_HINT[shape] * 5 // 4 + 1024 if shape in _EMIT_HINT else 24 * nq + 1024
        while True:
            pairs = torch.empty((capacity,), dtype=torch.int64, device=dev)
            _lib.check(lib.qrlsh_bucket_pairs_emit_chunked(_ptr(keys), layout[0], layout[1], layout[2], _ptr(pk),
                                                           _ptr(pid), _ptr(tk), _ptr(tid), nq, b, r, T, _ptr(ws),
                                                           ws.numel(), _ptr(pairs), capacity, _ptr(tot), _stream()))
            n, overflow = tot.tolist()
            if overflow:
                return None
            _EMIT_HINT[shape] = n
            if n <= capacity:
                return pairs[:n]
            del pairs
            capacity = n       # the cursor counted everything: the second run fits exactly
    _lib.check(lib.qrlsh_bucket_pairs_count(_ptr(keys), _ptr(pk), _ptr(pid), _ptr(tk), _ptr(tid), nq, b, r, T,
                                            _ptr(ws), ws.numel(), _ptr(tot), _stream()))
    n, overflow = tot.tolist()
    if overflow:
        return None
    del tk, tid
    pairs = torch.empty((n,), dtype=torch.int64, device=dev)
    if n:
        _lib.check(lib.qrlsh_bucket_pairs_fill(_ptr(pk), _ptr(pid), nq, b, r, T, _ptr(ws), _ptr(pairs), _stream()))
    return pairs


def emit_pairs_any(keys, r, stats=None):
    """All (i<j) pairs of every non-empty bucket, every band, from unsorted band-major keys
    [b,nq] (keys may be consumed): fast path, or the general sort path when it overflows."""
    emitted = emit_pairs_fast(keys, r)
    if stats is not None:
        stats["bucket_path"] = "partition+lds" if emitted is not None else "general-sort"
        stats["part_bits"] = part_bits_for(keys.shape[1])
    if emitted is None:
        sk, sid = bucket_sort(keys)
        emitted = emit_pairs(sk, sid, r)
    return emitted


def verify_pairs(sig, b, pairs):
    """Exact candidate test of lsh.py:31-53 per pair (shares a non-empty band): uint8 flags.
    Needed only for wide bands (r = P / b > 4), whose bucket ids are hashes."""
    lib = _lib.load()
    if not isinstance(sig, torch.Tensor) or sig.dtype not in (torch.int32, torch.int16):
        raise TypeError("sig must be an int32 or int16 (compact) tensor")
    _need(sig, sig.dtype, "sig", 2)
    _need(pairs, torch.int64, "pairs", 1)
    n = pairs.numel()
    flags = torch.empty((n,), dtype=torch.uint8, device=sig.device)
    code = _lib.SIG_U16 if sig.dtype == torch.int16 else _lib.SIG_I32
    _lib.check(lib.qrlsh_verify_pairs(_ptr(sig), code, sig.shape[1], b, _ptr(pairs), n, _ptr(flags), _stream()))
    return flags


def drop_unverified(sig, b, pairs, stats=None):
    """wide bands: keep only the pairs that really share a band (hash collisions of the bucket
    ids, astronomically rare, are removed here so the candidate set stays exact)"""
    if pairs.numel() == 0:
        return pairs
    flags = verify_pairs(sig, b, pairs)
    bad = int((flags == 0).sum().item())
    if stats is not None:
        stats["hash_collision_pairs_dropped"] = bad
    return pairs if bad == 0 else pairs[flags.bool()]


def candidate_pairs(keys, r, stats=None, sig=None):
    """get_candidates (lsh.py:40-55) on band-major keys [b,nq] (consumed): sorted unique
    int64 array of i<<32|j, i<j.  For wide bands (r > 4) pass the signature matrix `sig`: the
    hashed bucket ids are verified against it."""
    b, nq = keys.shape
    if r > 4 and sig is None:
        raise ValueError("band width r=%d > 4 needs the signatures for exact verification" % r)
    emitted = emit_pairs_any(keys, r, stats)
    if stats is not None:
        stats["emitted_pairs"] = int(emitted.numel())
    if emitted.numel() == 0:
        return emitted
    pairs = unique_pairs(emitted, nq, stats)
    if r > 4:
        pairs = drop_unverified(sig, b, pairs, stats)
    return pairs


# ---------------------------------------------------------------------------
# a5: scoring and top-K
# ---------------------------------------------------------------------------
def wide_ids(id_bits):
    """two ids + 11 score bits do not fit one 64-bit top-K key: use key + payload edges"""
    return 2 * id_bits + 11 > 64


def score_pairs(sig, norm2, pairs, want_cos=False, edge_id_bits=None, wide=None):
    """milli = rint(1000 * cosine(sig_i, sig_j)) per pair (recommender.py:203-204).  sig: int32
    rows or compact uint16 rows (torch.int16).
    Returns (milli int32, cos float64 | None, edges | None); edges is the int64 [2n] packed key
    tensor, or with wide ids (forced by wide=True) the tuple (keys int64 [2n], dst int32 [2n])."""
    lib = _lib.load()
    if not isinstance(sig, torch.Tensor) or sig.dtype not in (torch.int32, torch.int16):
        raise TypeError("sig must be an int32 or int16 (compact) tensor")
    _need(sig, sig.dtype, "sig", 2)
    if norm2 is not None:       # None: the kernel sums the norms from the rows itself (exact too, but slower)
        _need(norm2, torch.int64, "norm2", 1)
    _need(pairs, torch.int64, "pairs", 1)
    n = pairs.numel()
    dev = sig.device
    milli = torch.empty((n,), dtype=torch.int32, device=dev)
    cosv = torch.empty((n,), dtype=torch.float64, device=dev) if want_cos else None
    edges = torch.empty((2 * n,), dtype=torch.int64, device=dev) if edge_id_bits is not None else None
    if wide is None:
        wide = edge_id_bits is not None and wide_ids(edge_id_bits)
    edst = torch.empty((2 * n,), dtype=torch.int32, device=dev) if (wide and edges is not None) else None
    code = _lib.SIG_U16 if sig.dtype == torch.int16 else _lib.SIG_I32
    _lib.check(lib.qrlsh_score_pairs(_ptr(sig), code, _ptr(norm2), sig.shape[1], _ptr(pairs), n, _ptr(milli),
                                     _ptr(cosv), _ptr(edges), edge_id_bits if edge_id_bits is not None else 0,
                                     _ptr(edst), _stream()))
    return milli, cosv, ((edges, edst) if edst is not None else edges)


class RemoteIds:
    """the set of remote query ids a rank's pairs touch (bitmap + rank structure on the device)"""

    def __init__(self, ws, nids, q0, nql, bounds):
        self.ws, self.nids, self.q0, self.nql, self.bounds = ws, nids, q0, nql, bounds


def remote_ids(pairs, q0, nql, nids, world):
    """-> RemoteIds of the endpoints of `pairs` outside [q0, q0 + nql); .bounds (device int64 [world + 1]) =
    number of such ids below g * nql, i.e. the per-owner request sizes, [-1] = the total"""
    lib = _lib.load()
    _need(pairs, torch.int64, "pairs", 1)
    dev = pairs.device
    ws = _ws(lib.qrlsh_idset_workspace_bytes(nids), dev)
    bounds = torch.empty((world + 1,), dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_idset_build(_ptr(pairs), pairs.numel(), int(q0), int(nql), int(nids), int(nql), world,
                                     _ptr(ws), ws.numel(), _ptr(bounds), _stream()))
    return RemoteIds(ws, nids, q0, nql, bounds)


def remote_id_list(rid, total):
    """the ids of the set, ascending (int64 [total])"""
    lib = _lib.load()
    out = torch.empty((total,), dtype=torch.int64, device=rid.ws.device)
    if total:
        _lib.check(lib.qrlsh_idset_list(_ptr(rid.ws), rid.nids, _ptr(out), _stream()))
    return out


def remap_pairs_ids(pairs, rid):
    """pairs (global ids) -> slot(i) << 32 | slot(j) over the row table [local rows | rows of rid's ids]"""
    lib = _lib.load()
    _need(pairs, torch.int64, "pairs", 1)
    out = torch.empty_like(pairs)
    _lib.check(lib.qrlsh_idset_remap(_ptr(pairs), pairs.numel(), int(rid.q0), int(rid.nql), _ptr(rid.ws), rid.nids,
                                     _ptr(out), _stream()))
    return out


def gather_rows(sig, norm2, ids, q0):
    """signature rows + norms of the global ids `ids` (all owned by this rank: row = id - q0)"""
    lib = _lib.load()
    _need(ids, torch.int64, "ids", 1)
    n = ids.numel()
    rows = torch.empty((n, sig.shape[1]), dtype=sig.dtype, device=sig.device)
    norms = torch.empty((n,), dtype=torch.int64, device=sig.device)
    _lib.check(lib.qrlsh_gather_rows(_ptr(sig), sig.shape[1] * sig.element_size(), _ptr(norm2), _ptr(ids), n, int(q0),
                                     _ptr(rows), _ptr(norms), _stream()))
    return rows, norms


def gather_sets(ids, offs_all, rows_all, nql):
    """CSR (offsets int64 [n + 1], rows int32) of the answer sets of the global query ids `ids` (int64, ascending or
    not) out of the replicated per-shard arrays: offs_all [world, nql + 1] int32 / int64, rows_all [world, max_nnz]
    int16 (unsigned 16-bit row ids) / int32 (one read-back: the number of row ids)"""
    lib = _lib.load()
    _need(ids, torch.int64, "ids", 1)
    if offs_all.dtype not in (torch.int32, torch.int64) or rows_all.dtype not in (torch.int16, torch.int32):
        raise TypeError("offs_all must be int32 / int64, rows_all int16 / int32")
    _need(offs_all, offs_all.dtype, "offs_all", 2)
    _need(rows_all, rows_all.dtype, "rows_all", 2)
    world, max_nnz = rows_all.shape
    if tuple(offs_all.shape) != (world, nql + 1):
        raise ValueError("offs_all must be [world, nql + 1]")
    n, dev = ids.numel(), ids.device
    off = torch.empty((n + 1,), dtype=torch.int64, device=dev)
    ws = _ws(lib.qrlsh_gather_sets_workspace_bytes(n), dev)
    _lib.check(lib.qrlsh_gather_sets_count(_ptr(ids), n, _ptr(offs_all), offs_all.element_size(), int(nql), world, _ptr(off),
                                           _ptr(ws), ws.numel(), _stream()))
    total = int(off[-1].item())
    rows = torch.empty((total,), dtype=torch.int32, device=dev)
    if total:
        _lib.check(lib.qrlsh_gather_sets_fill(_ptr(ids), n, _ptr(offs_all), offs_all.element_size(), _ptr(rows_all),
                                              rows_all.element_size(), int(nql), max_nnz, _ptr(off), _ptr(rows), _stream()))
    return off, rows


def score_pairs_split(sig, norm2, sig_b, norm2_b, pairs):
    """milli of pairs whose halves index the two-piece row table [sig | sig_b]"""
    lib = _lib.load()
    _need(pairs, torch.int64, "pairs", 1)
    n = pairs.numel()
    milli = torch.empty((n,), dtype=torch.int32, device=pairs.device)
    code = _lib.SIG_U16 if sig.dtype == torch.int16 else _lib.SIG_I32
    _lib.check(lib.qrlsh_score_pairs_split(_ptr(sig), _ptr(norm2), sig.shape[0], _ptr(sig_b), _ptr(norm2_b), code,
                                           sig.shape[1], _ptr(pairs), n, _ptr(milli), _stream()))
    return milli


def pair_edges_interleaved(pairs, milli, id_bits, wide=False):
    """both directed edges of every scored pair, [2t] = i -> j, [2t + 1] = j -> i (what score_pairs writes):
    packed int64 [2n], or with wide ids (keys int64 [2n], dst int32 [2n])"""
    lib = _lib.load()
    n, dev = pairs.numel(), pairs.device
    e = torch.empty((2 * n,), dtype=torch.int64, device=dev)
    d = torch.empty((2 * n,), dtype=torch.int32, device=dev) if wide else None
    _lib.check(lib.qrlsh_pair_edges(_ptr(pairs), _ptr(milli), n, 0 if wide else int(id_bits), _ptr(e), None, _ptr(d),
                                    None, _stream()))
    return (e, d) if wide else e


def local_bits_for(nql):
    return max(1, int(nql - 1).bit_length()) if nql > 1 else 1


def topk_edges_local(edges, edst, K, id_bits, q0, nql, select=True):
    """Per-query top-K from edges that arrived from several scoring ranks (no useful order): re-base the
    src field to this rank's id range, sort the whole (src, 1000 - milli, dst) key, cut.  edst: the dst
    payload of key + payload edges, or None for packed ones."""
    lib = _lib.load()
    dev = edges.device
    n = edges.numel()
    if n == 0:
        z = torch.empty((0,), dtype=torch.int32, device=dev)
        return z, z.clone(), z.clone()
    lb = local_bits_for(nql)
    if lb + 11 + id_bits > 64:
        raise NotImplementedError("%d local + %d global id bits do not fit the 64-bit top-K key" % (lb, id_bits))
    loc = torch.empty_like(edges)
    _lib.check(lib.qrlsh_edges_localize(_ptr(edges), _ptr(edst), n, id_bits, int(q0), int(nql), _ptr(loc), _stream()))
    if select and K <= SELECT_MAX_K and n < (1 << 31):
        # select form: the re-based keys ARE reverse words (src = a local query); sorted on the src bits alone
        # (ceil(lb / 8) passes instead of ceil((lb + 11 + id_bits) / 8)), every edge then ranks itself in its query's run
        src, dst, val = topk_select(None, None, loc, K, id_bits, int(nql))
        src += int(q0)
        return src, dst, val
    se, _ = sort_u64(loc, None, 0, lb + 11 + id_bits)
    ws = _ws(lib.qrlsh_compact_workspace_bytes(n), dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_topk_count(_ptr(se), n, K, id_bits, _ptr(ws), ws.numel(), _ptr(total), _stream()))
    m = int(total.item())
    src = torch.empty((m,), dtype=torch.int32, device=dev)
    dst = torch.empty((m,), dtype=torch.int32, device=dev)
    val = torch.empty((m,), dtype=torch.int32, device=dev)
    _lib.check(lib.qrlsh_topk_fill_based(_ptr(se), None, n, K, id_bits, int(q0), _ptr(ws), _ptr(src), _ptr(dst),
                                         _ptr(val), _stream()))
    return src, dst, val


def score_pairs_rev(sig, norm2, pairs, id_bits, wide=None):
    """milli per pair + the reverse edge word of every pair (src = j): -> (milli int32, rev) with rev the packed
    int64 [n] words, or with wide ids the tuple (keys int64 [n], dst int32 [n]).  Input of topk_select."""
    lib = _lib.load()
    _need(sig, sig.dtype, "sig", 2)
    if norm2 is not None:
        _need(norm2, torch.int64, "norm2", 1)
    _need(pairs, torch.int64, "pairs", 1)
    n, dev = pairs.numel(), sig.device
    if wide is None:
        wide = wide_ids(id_bits)
    milli = torch.empty((n,), dtype=torch.int32, device=dev)
    rev = torch.empty((n,), dtype=torch.int64, device=dev)
    rdst = torch.empty((n,), dtype=torch.int32, device=dev) if wide else None
    code = _lib.SIG_U16 if sig.dtype == torch.int16 else _lib.SIG_I32
    _lib.check(lib.qrlsh_score_pairs_rev(_ptr(sig), code, _ptr(norm2), sig.shape[1], _ptr(pairs), n, _ptr(milli), _ptr(rev),
                                         0 if wide else int(id_bits), _ptr(rdst), _stream()))
    return milli, ((rev, rdst) if wide else rev)


SELECT_MAX_K = 256   # SEL_MAXK in csrc/pairs.hip


def topk_select(pairs, milli, rev, K, id_bits, nq):
    """Per-query top-K (recommender.py:206-210) from the sorted scored pairs and their reverse words
    (score_pairs_rev): the reverse words are sorted on j's bits only, then every directed edge ranks itself
    inside its query's two runs.  -> (src, dst, milli) int32, the same COO topk_edges returns.
    pairs = milli = None: the lists are made of the reverse words alone (packed words src << (id_bits + 11) |
    inv << id_bits | neighbour with src < nq: topk_edges_local)."""
    lib = _lib.load()
    rdst = None
    if isinstance(rev, tuple):
        rev, rdst = rev
    dev = rev.device
    n = rev.numel()
    if n == 0:
        z = torch.empty((0,), dtype=torch.int32, device=dev)
        return z, z.clone(), z.clone()
    if K > SELECT_MAX_K:
        raise ValueError("topk_select handles K <= %d; use topk_edges (the sort form) beyond" % SELECT_MAX_K)
    if rdst is None:
        src_bits = id_bits if pairs is not None else local_bits_for(nq)
        rs, rd = sort_u64(rev, None, id_bits + 11, id_bits + 11 + src_bits)
    else:
        rs, rd = sort_u64(rev, rdst, 11, 11 + id_bits)
    ws = _ws(lib.qrlsh_topk_select_workspace_bytes(nq), dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_topk_select_count(_ptr(pairs), n, _ptr(rs), _ptr(rd), int(nq), K, int(id_bits), _ptr(ws), ws.numel(),
                                           _ptr(total), _stream()))
    m = int(total.item())
    src = torch.empty((m,), dtype=torch.int32, device=dev)
    dst = torch.empty((m,), dtype=torch.int32, device=dev)
    val = torch.empty((m,), dtype=torch.int32, device=dev)
    _lib.check(lib.qrlsh_topk_select_fill(_ptr(pairs), _ptr(milli), n, _ptr(rs), _ptr(rd), int(nq), K, int(id_bits), _ptr(ws),
                                          _ptr(src), _ptr(dst), _ptr(val), _stream()))
    return src, dst, val


def remap_pairs(pairs, q0, nql, need):
    """pairs (i local to [q0, q0+nql), j anywhere) -> (i - q0) << 32 | slot(j) over the row table
    [local rows | rows of the ascending global ids `need`] (multi-GPU scoring, qrlsh/dist.py)"""
    lib = _lib.load()
    _need(pairs, torch.int64, "pairs", 1)
    _need(need, torch.int64, "need", 1)
    out = torch.empty_like(pairs)
    _lib.check(lib.qrlsh_remap_pairs(_ptr(pairs), pairs.numel(), int(q0), int(nql), _ptr(need), need.numel(),
                                     _ptr(out), _stream()))
    return out


def pair_edges(pairs, milli, id_bits, wide=False):
    """forward / reverse directed edge keys of scored pairs: (fwd, rev) int64, or with wide ids
    ((fwd_keys, fwd_dst), (rev_keys, rev_dst))"""
    lib = _lib.load()
    _need(pairs, torch.int64, "pairs", 1)
    _need(milli, torch.int32, "milli", 1)
    n, dev = pairs.numel(), pairs.device
    fwd = torch.empty((n,), dtype=torch.int64, device=dev)
    rev = torch.empty((n,), dtype=torch.int64, device=dev)
    fd = torch.empty((n,), dtype=torch.int32, device=dev) if wide else None
    rd = torch.empty((n,), dtype=torch.int32, device=dev) if wide else None
    _lib.check(lib.qrlsh_pair_edges(_ptr(pairs), _ptr(milli), n, 0 if wide else int(id_bits), _ptr(fwd), _ptr(rev),
                                    _ptr(fd), _ptr(rd), _stream()))
    return ((fwd, fd), (rev, rd)) if wide else (fwd, rev)


def topk_edges(edges, K, id_bits):
    """Per-query top-K (recommender.py:206-210) from the directed edges written by score_pairs
    (packed int64 keys, or the (keys, dst) tuple of the wide-id format):
    -> (src, dst, milli) int32, sorted by src, value desc, dst asc; <= K per src."""
    lib = _lib.load()
    edst = None
    if isinstance(edges, tuple):
        edges, edst = edges
    _need(edges, torch.int64, "edges", 1)
    dev = edges.device
    n = edges.numel()
    if n == 0:
        z = torch.empty((0,), dtype=torch.int32, device=dev)
        return z, z.clone(), z.clone()
    # Sorting on (src, 1000-milli) alone is enough: the sort is stable and, per src, the edges
    # arrive with dst ascending (pairs are sorted by (i, j); edge 2t is i->j, 2t+1 is j->i, so a
    # src first meets its smaller neighbours, then its larger ones) -- 4 passes instead of 7.
    if edst is None:
        se, sd = sort_u64(edges, None, id_bits, 2 * id_bits + 11)
        kb = id_bits
    else:
        se, sd = sort_u64(edges, edst, 0, id_bits + 11)
        kb = 0
    ws = _ws(lib.qrlsh_compact_workspace_bytes(n), dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(lib.qrlsh_topk_count(_ptr(se), n, K, kb, _ptr(ws), ws.numel(), _ptr(total), _stream()))
    m = int(total.item())
    src = torch.empty((m,), dtype=torch.int32, device=dev)
    dst = torch.empty((m,), dtype=torch.int32, device=dev)
    val = torch.empty((m,), dtype=torch.int32, device=dev)
    _lib.check(lib.qrlsh_topk_fill(_ptr(se), _ptr(sd), n, K, kb, _ptr(ws), _ptr(src), _ptr(dst), _ptr(val), _stream()))
    return src, dst, val
