"""Query-sharded hot path over one process per GPU (torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Rank g owns the contiguous query range [g*nql, (g+1)*nql) and bands {k : k // ceil(b/W) == g}.

  1. MinHash + band keys + norms on the local shard (no communication).
  2. bucket-id exchange so that cross-shard candidates are found.  Two modes with identical
     results:
       "all_to_all" (default): band-partitioned -- a rank receives only the bands it owns,
                     b/W * nq_total * 8 bytes instead of b * nq_total * 8;
       "all_gather": every rank receives every band key (the exchange BASELINE.json's
                     north_star names), then keeps its bands.
  3. signatures for scoring.  An owner scores pairs (i local, j anywhere), so it needs the signature
     rows of the j that live on other ranks.  Two ways, identical results:
       "fetch" : after step 5 the owner asks the owning ranks for exactly the distinct remote j of
                 its pairs (ids out, rows + norms back: two all-to-alls).  With p pairs per rank
                 that is at most p rows however large the world is -- a third of the gather's
                 volume at 8 ranks on the bench workload, most of it at 2;
       "all_gather": every rank receives every row ((W-1)/W * nq_total * 2P bytes per rank) on a
                 second communicator, asynchronously, beside steps 4-5.
     "auto" (default): all_gather below 4 ranks (one or two peers: the volumes are close and the
     gather hides behind steps 4-5), fetch from 4 ranks on (p rows against 3 - 7 shards).
  4. per owned band: bucket partition + pair emission over ALL queries; sorted by i only.
  5. pairs go to the owner of their smaller query id (variable-size all-to-all); the owner
     sorts + uniques what it received -> its share of the global candidate set.
  6. owners score their pairs against the gathered signatures; the reverse edge (j -> i) of
     every scored pair goes to the owner of j (variable-size all-to-all).
  7. per-query top-K on the local edges.

The compute steps go through a small backend object so that the host logic above can be
exercised on CPU (gloo) with the oracle standing in for the kernels (tests only); the
default backend is the HIP library and nothing else ships.
"""
import torch
import torch.distributed as dist

from . import ops
from .pipeline import HotPathResult


class HipBackend:
    """libqrlsh kernels (the product path)."""

    rows_hint = 0   # queries this rank owns (set by the driver): sizes the rows of the pair de-dup

    def minhash(self, offsets, rows, table, b):
        return ops.minhash(offsets, rows, table, b=b, want_norm=True, compact=ops.can_compact(table))

    def emit_pairs(self, keys, r):
        return ops.emit_pairs_any(keys, r)

    def emit_pairs_chunked(self, recv, world, nb, nql, r):
        """emit_pairs on the [world][nb][nql] buffer of the band-partitioned exchange, read in place;
        the transposing copy to [nb][world * nql] is made only if the general path is needed"""
        if world * nql <= (1 << 24):
            pairs = ops.emit_pairs_fast(recv, r, chunks=(world, nb, nql))
            if pairs is not None:
                return pairs
        return ops.emit_pairs_any(_owned_bands(recv, world, nb, nql), r)

    def sort_unique(self, words, bit_ranges):
        # bit_ranges = [(0, ib), (32, 32 + ib)]: pair words i << 32 | j
        ib = bit_ranges[0][1]
        g = ops.row_group_bits(ib, words.numel() / max(1, self.rows_hint or 1))
        grouped, _ = ops.sort_u64(words, None, 32 + g, 32 + ib)  # by i >> g only; rows are finished in LDS
        pairs = ops.row_unique(grouped, g, ib)
        if pairs is None:                                        # a row too long for the LDS image
            words, _ = ops.sort_u64(grouped, None, 0, 2 * ib, fold=ib)
            pairs = ops.unique_sorted(words)
        return pairs

    def sort_words(self, words, lo, hi):
        return ops.sort_u64(words, None, lo, hi)[0]

    def owner_sizes(self, words, lo, shard, world):
        """per-destination counts of words already grouped by owner = (word >> lo) // shard"""
        return ops.owner_sizes(words, lo, shard, world) if words.numel() else [0] * world

    def group_by_owner(self, words, lo, shard, vals=None):
        """one stable pass that orders words (and vals) by (word >> lo) // shard"""
        return ops.sort_u64(words, vals, lo, lo + 1, owner_shard=shard)

    def remap_pairs(self, pairs, q0, nql, need):
        return ops.remap_pairs(pairs, q0, nql, need)

    def pair_edges(self, pairs, milli, id_bits, wide):
        return ops.pair_edges(pairs, milli, id_bits, wide)

    def score_only(self, sig_rows, norm_rows, pairs):
        """milli of pairs whose two halves index rows of sig_rows"""
        return ops.score_pairs(sig_rows, norm_rows, pairs)[0]

    def verify_flags(self, sig_rows, b, pairs):
        return ops.verify_pairs(sig_rows, b, pairs)

    def sort_words_kv(self, words, vals, lo, hi):
        return ops.sort_u64(words, vals, lo, hi)

    def topk(self, edges, K, id_bits):
        return ops.topk_edges(edges, K, id_bits)


def _owned_bands(recv, world, nb, nql):
    """[world][nb][nql] (as received) -> band-major [nb][world * nql]"""
    return recv.view(world, nb, nql).permute(1, 0, 2).reshape(nb, world * nql).contiguous()


_BG_GROUPS = {}


def background_group(group=None):
    """A second communicator over the same ranks for the long signature all-gather, so that it
    runs beside the short exchanges instead of ahead of them (collectives of ONE communicator
    execute in issue order).  Created collectively on first use, then cached."""
    world = dist.get_world_size(group)
    if world == 1:
        return group
    key = id(group) if group is not None else 0
    if key not in _BG_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(world))
        _BG_GROUPS[key] = dist.new_group(ranks=ranks)
    return _BG_GROUPS[key]


def band_owner_ranges(b, world):
    """contiguous band blocks: rank g owns [lo[g], hi[g])"""
    per = (b + world - 1) // world
    return [(min(g * per, b), min((g + 1) * per, b)) for g in range(world)]


class _Done:
    def wait(self):
        return None


def _staged(t, group):
    """gloo cannot move device tensors through every collective: stage them through the host.
    Only used by tests that run several gloo ranks on one GPU; RCCL never takes this path."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(out, inp, group=None, async_op=False):
    if inp.dtype == torch.int16:  # compact signature rows: RCCL has no int16, move them as bytes
        out, inp = out.view(torch.uint8), inp.view(torch.uint8)
    if _staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    h = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return h if async_op else _Done()


def _all_to_all(out, inp, osplit=None, isplit=None, group=None):
    if _staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=osplit, input_split_sizes=isplit, group=group)
        out.copy_(o)
        return
    dist.all_to_all_single(out, inp, output_split_sizes=osplit, input_split_sizes=isplit, group=group)


def _exchange_var(chunks_sizes, send, group=None, want_sizes=False):
    """variable-size all-to-all of a 1-D int64 tensor already ordered by destination.
    chunks_sizes: python list of per-destination element counts."""
    world = dist.get_world_size(group)
    dev = send.device
    sizes = torch.tensor(chunks_sizes, dtype=torch.int64, device=dev)
    rsizes = torch.empty(world, dtype=torch.int64, device=dev)
    _all_to_all(rsizes, sizes, group=group)
    rs = rsizes.tolist()
    recv = torch.empty(int(sum(rs)), dtype=send.dtype, device=dev)
    _all_to_all(recv, send, rs, list(chunks_sizes), group)
    return (recv, rs) if want_sizes else recv


def _fetch_rows(sig, norm2, need, nql, group=None):
    """Signature rows + norms of the global query ids `need` (int64, ascending, none of them local):
    ids go to the ranks that own them, rows and norms come back in the same order.  One payload per
    row: its signature bytes followed by the 8 bytes of its norm."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = sig.device
    bounds = torch.arange(world + 1, dtype=torch.int64, device=dev) * nql
    cuts = torch.searchsorted(need, bounds).tolist()
    sizes = [cuts[g + 1] - cuts[g] for g in range(world)]
    req, rs = _exchange_var(sizes, need, group, want_sizes=True)
    local = req - rank * nql
    rowbytes = sig.shape[1] * sig.element_size()
    out = torch.empty((local.numel(), rowbytes + 8), dtype=torch.uint8, device=dev)
    out[:, :rowbytes] = sig.index_select(0, local).view(torch.uint8).view(local.numel(), rowbytes)
    out[:, rowbytes:] = norm2.index_select(0, local).view(torch.uint8).view(local.numel(), 8)
    got = torch.empty((need.numel(), rowbytes + 8), dtype=torch.uint8, device=dev)
    _all_to_all(got, out, sizes, rs, group)
    rows = got[:, :rowbytes].contiguous().view(sig.dtype).view(need.numel(), sig.shape[1])
    norms = got[:, rowbytes:].contiguous().view(torch.int64).view(need.numel())
    return rows, norms


def query_similarities_sharded(offsets, rows, table, b, K, nq_total, exchange="all_to_all", backend=None,
                               group=None, wide_ids=None, sig_exchange="auto"):
    """Hot path for this rank's query shard; collective over `group`.  Every rank must hold
    the same number of queries (nq_total % world == 0).  Returns a HotPathResult whose pairs /
    top-K rows are this rank's share (global query ids); concatenated over ranks in rank
    order they equal the single-GPU result."""
    be = backend if backend is not None else HipBackend()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nql = offsets.numel() - 1
    be.rows_hint = nql
    if nql * world != nq_total:
        raise ValueError("every rank must own nq_total / world queries (got %d x %d != %d)" % (nql, world, nq_total))
    P = table.P
    if P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))
    r = P // b
    ib = ops.id_bits_for(nq_total)
    wide = ops.wide_ids(ib) if wide_ids is None else wide_ids
    dev = offsets.device
    stats = {}

    # 1. local signatures
    sig, norm2, keys = be.minhash(offsets, rows, table, b)

    # 2. bucket-id exchange (short, needed at once: issued before the long gather)
    ranges = band_owner_ranges(b, world)
    lo, hi = ranges[rank]
    nb = hi - lo
    if exchange == "all_gather":
        allk = torch.empty((world * b, nql), dtype=torch.int64, device=dev)
        _all_gather(allk, keys, group)
        owned = allk.view(world, b, nql)[:, lo:hi, :].permute(1, 0, 2).reshape(nb, nq_total).contiguous()
        del allk
    elif exchange == "all_to_all":
        in_split = [h - l for (l, h) in ranges]
        recv = torch.empty((world * nb, nql), dtype=torch.int64, device=dev)
        _all_to_all(recv, keys, [nb] * world, in_split, group)
        owned = None            # read in place by emit_pairs_chunked
    else:
        raise ValueError("exchange must be 'all_to_all' or 'all_gather'")
    del keys

    # 3. "all_gather": async gather of the signature rows + norms on the background communicator
    #    (consumed in step 6; overlaps steps 4-5).  "fetch" / "auto": nothing yet, see step 6.
    if sig_exchange not in ("auto", "fetch", "all_gather"):
        raise ValueError("sig_exchange must be 'auto', 'fetch' or 'all_gather'")

    def start_gather():
        bg = background_group(group)
        sa = torch.empty((nq_total, P), dtype=sig.dtype, device=dev)
        na = torch.empty((nq_total,), dtype=torch.int64, device=dev)
        return sa, na, _all_gather(sa, sig, bg, async_op=True), _all_gather(na, norm2, bg, async_op=True)

    if sig_exchange == "auto":
        sig_exchange = "all_gather" if world in (2, 3) else "fetch"
    gathered = start_gather() if sig_exchange == "all_gather" else None

    # 4. candidates of the owned bands over all queries
    pair_bits = [(0, ib), (32, 32 + ib)]
    if nb > 0:
        emitted = be.emit_pairs(owned, r) if owned is not None else be.emit_pairs_chunked(recv, world, nb, nql, r)
        stats["emitted_pairs"] = int(emitted.numel())
        # only order by i (so the list splits by owner); duplicates across this rank's few bands are
        # rare and the owner de-duplicates anyway, so the local unique is not worth its passes
        mine = be.group_by_owner(emitted, 32, nql)[0] if emitted.numel() else emitted
    else:
        stats["emitted_pairs"] = 0
        mine = torch.empty((0,), dtype=torch.int64, device=dev)
    owned = recv = None

    # 5. pairs -> owner of i
    got = _exchange_var(be.owner_sizes(mine, 32, nql, world), mine, group)
    pairs = be.sort_unique(got, pair_bits) if got.numel() else got

    # 6. score on the owner; reverse edges -> owner of j
    q0 = rank * nql
    pj = pairs & 0xFFFFFFFF
    if gathered is None:
        # which rows of other ranks do my pairs touch?  (i is local by construction)
        remote = (pj < q0) | (pj >= q0 + nql)
        need = torch.unique(pj[remote])                     # ascending
    stats["sig_exchange"] = "all_gather" if gathered is not None else "fetch"
    if gathered is not None:
        sig_rows, norm_rows, h_sig, h_nrm = gathered
        h_sig.wait()
        h_nrm.wait()
        local_pairs = pairs                                  # row index == global query id
    else:
        if world > 1:
            rrows, rnorms = _fetch_rows(sig, norm2, need, nql, group)
        stats["remote_rows_fetched"] = int(need.numel())
        sig_rows = torch.cat([sig, rrows]) if need.numel() else sig
        norm_rows = torch.cat([norm2, rnorms]) if need.numel() else norm2
        local_pairs = be.remap_pairs(pairs, q0, nql, need)    # both halves index rows of sig_rows
    if r > 4 and pairs.numel():   # wide bands: hashed bucket ids -> exact verification on the owner
        keep = be.verify_flags(sig_rows, b, local_pairs).bool()
        if not bool(keep.all()):
            pairs, local_pairs = pairs[keep], local_pairs[keep]
    milli = be.score_only(sig_rows, norm_rows, local_pairs)
    if wide:
        # key + payload edges (src << 11 | inv, dst): ids of any width
        (fwd_k, fwd_d), (rev_k, rev_d) = be.pair_edges(pairs, milli, ib, True)
        if pairs.numel():
            rev_k, rev_d = be.group_by_owner(rev_k, 11, nql, rev_d)
        sizes = be.owner_sizes(rev_k, 11, nql, world)
        rk_in = _exchange_var(sizes, rev_k, group)
        rd_in = _exchange_var(sizes, rev_d.view(torch.int32), group)
        edges_local = (torch.cat([rk_in, fwd_k]), torch.cat([rd_in, fwd_d]))
    else:
        fwd, rev = be.pair_edges(pairs, milli, ib, False)
        if pairs.numel():
            rev = be.group_by_owner(rev, ib + 11, nql)[0]
        rev_in = _exchange_var(be.owner_sizes(rev, ib + 11, nql, world), rev, group)
        edges_local = torch.cat([rev_in, fwd])

    # 7. local top-K.  Order matters for the stable top-K sort: per src, reverse edges (dst < src,
    # ascending by sender rank and pair order) come before forward edges (dst > src, ascending)
    src, dst, val = be.topk(edges_local, K, ib)
    stats["unique_pairs"] = int(pairs.numel())
    stats["kept_edges"] = int(src.numel())
    return HotPathResult(sig, norm2, pairs, milli, src, dst, val, K, b, stats)
