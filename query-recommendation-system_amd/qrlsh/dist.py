"""Query-sharded hot path over one process per GPU (torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Rank g owns the contiguous query range [g*nql, (g+1)*nql), nql = ceil(nq_total / W) (the last shards are
padded with empty answer sets, which are never candidates), and bands {k : k // ceil(b/W) == g}.

With `sig_exchange` = "fetch" or "all_gather" (shards stay shards):

  1. MinHash + band keys + norms on the local shard (no communication).
  2. bucket-id exchange so that cross-shard candidates are found.  Two modes with identical
     results:
       "all_to_all" (default): band-partitioned -- a rank receives only the bands it owns,
                     b/W * nq_total * 8 bytes instead of b * nq_total * 8;
       "all_gather": every rank receives every band key (the exchange BASELINE.json's
                     north_star names), then keeps its bands.
  3. per owned band: bucket partition + pair emission over ALL queries.
  4. every emitted pair goes to the rank that scores it (variable-size all-to-all): the owner of ONE of
     its two queries, picked by a bit of mix64(pair) (csrc/common.h: qr_pair_host).  "Always the owner
     of the smaller id" would give rank g a share proportional to the ids above its shard -- twice the
     mean on rank 0, nothing on the last rank -- whatever the data; the coin splits evenly.  The scoring
     rank sorts + uniques what it received -> its share of the global candidate set (the shares are
     disjoint; their union is the single-GPU candidate list).
  5. signatures for scoring.  At most one row of a pair is remote.
       "fetch" : the scoring rank asks the owners for exactly the distinct remote ids of its pairs (ids
                 out, rows + norms back);
       "all_gather": every rank receives every row ((W-1)/W * nq_total * 2P bytes per rank) on a
                 second communicator, asynchronously, beside steps 3-4.
  6. score; both directed edges of every scored pair go to the owner of their src (variable-size
     all-to-all; the local share passes through it as a device copy).
  7. per-query top-K on the edges received: re-based to the local id range, sorted on the whole
     (src, value, dst) key (edges of one query come from several scoring ranks in no useful order).

With `sig_exchange` = "recompute" (answer sets replicated instead of signatures exchanged):

  0. all-gather of the ANSWER SETS (row ids padded to the largest shard + the offsets, 64 + 8 bytes per
     query) on the second communicator, beside the rank's own MinHash;
  1. every rank computes the signatures, norms and band keys of ALL queries, its own shard first, each
     shard's block written straight into its place of the replicated tables;
  2. -- no bucket-id exchange: the keys of the owned bands are read in place from the replicated
     [rank][band][queries] key buffer -- and no step 5: every row is local.  Steps 3, 4, 6, 7 as above.

With `sig_exchange` = "sets": steps 1 - 4 and 6 - 7 as in "fetch", the answer-set gather of step 0 runs in the
background from the start of the step, and at step 5 the rank computes the signatures of exactly the remote queries
its pairs touch from the replicated answer sets (one MinHash launch; no requests, no rows on the wire).

"auto" (default) = "recompute" for 2 to 4 ranks -- few ranks = few xGMI links, and everything a rank sends to one
peer crosses ONE of them: 64 B of answer set per query against 256 B of signature row per scored pair plus 8 B of
bucket id per band -- "sets" from five ranks on (seven links to spread the bucket ids over; the rows of "fetch" would
be 70 % of a rank's incoming bytes), "fetch" for one rank.

Host round trips per step: the emitted-pair count, one per size exchange (pairs, row requests, edges),
the unique-pair count and the top-K count.  Everything else is a libqrlsh kernel or a collective.

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
    validate = True  # check the answer sets before MinHash gathers by their row ids (ops.check_csr)

    def __init__(self):
        self.stats = {}   # which paths the kernels took (bucket_path, dedup_path, group_bits, part_bits)

    def minhash(self, offsets, rows, table, b, out=None, validate=None, keys=True):
        """(sig, norm2, band keys | None when keys is False)"""
        return ops.minhash(offsets, rows, table, b=b if keys else None, want_norm=True, compact=ops.can_compact(table),
                           validate=self.validate if validate is None else validate, out=out)

    def sig_dtype(self, table):
        return torch.int16 if ops.can_compact(table) else torch.int32

    def emit_pairs_bands(self, keys_all, lo, hi, r):
        """emit_pairs for the bands [lo, hi) of every rank's [b][nql] key block in keys_all [world][b][nql],
        read in place"""
        world, b, nql = keys_all.shape
        nb = hi - lo
        pairs = ops.emit_pairs_fast(keys_all.view(-1)[lo * nql:], r, chunks=(world, nb, nql, b * nql))
        if pairs is not None:
            self.stats["bucket_path"] = "partition+lds"
            self.stats["part_bits"] = ops.part_bits_for(world * nql)
            return pairs
        return ops.emit_pairs_any(_band_major(keys_all, lo, hi), r, self.stats)

    def emit_pairs(self, keys, r):
        return ops.emit_pairs_any(keys, r, self.stats)

    def emit_pairs_chunked(self, recv, world, nb, nql, r):
        """emit_pairs on the [world][nb][nql] buffer of the band-partitioned exchange, read in place;
        the transposing copy to [nb][world * nql] is made only if the general path is needed"""
        pairs = ops.emit_pairs_fast(recv, r, chunks=(world, nb, nql))
        if pairs is not None:
            self.stats["bucket_path"] = "partition+lds"
            self.stats["part_bits"] = ops.part_bits_for(world * nql)
            return pairs
        return ops.emit_pairs_any(_owned_bands(recv, world, nb, nql), r, self.stats)

    def group_pairs_by_host(self, pairs, nql, world):
        """pairs ordered by the rank that scores them -> (grouped, bounds int64 [world + 1] on the device)"""
        if pairs.numel() == 0:
            return pairs, torch.zeros((world + 1,), dtype=torch.int64, device=pairs.device)
        g, _ = ops.sort_u64(pairs, None, host_shard=nql)
        return g, ops.owner_bounds(g, -1, nql, world)

    def sort_unique(self, words, nids, words_per_query=None):
        # the words a rank receives touch ~2 x its own share of the ids (the other endpoint of half its pairs)
        if words_per_query is None:
            words_per_query = words.numel() / max(1, 2 * (self.rows_hint or 1))
        return ops.unique_pairs(words, nids, self.stats, words_per_query=words_per_query)

    def remote_ids(self, pairs, q0, nql, nids, world):
        return ops.remote_ids(pairs, q0, nql, nids, world)

    def remote_id_list(self, rid, total):
        return ops.remote_id_list(rid, total)

    def remap_pairs(self, pairs, rid):
        return ops.remap_pairs_ids(pairs, rid)

    def gather_rows(self, sig, norm2, ids, q0):
        return ops.gather_rows(sig, norm2, ids, q0)

    def gather_sets(self, ids, offs_all, rows_all, nql):
        """CSR of the answer sets of `ids` out of the replicated per-shard arrays (qrlsh_gather_sets_*)"""
        return ops.gather_sets(ids, offs_all, rows_all, nql)

    def score(self, sig, norm2, sig_b, norm2_b, pairs):
        """milli of pairs whose halves index the row table [sig | sig_b]"""
        if sig_b is None:
            sig_b, norm2_b = sig[:0], norm2[:0]
        return ops.score_pairs_split(sig, norm2, sig_b, norm2_b, pairs)

    def verify_flags(self, sig_rows, b, pairs):
        return ops.verify_pairs(sig_rows, b, pairs)

    def edges(self, pairs, milli, ib, wide):
        """-> (keys int64 [2n], dst int32 [2n] | None), edge 2t = i -> j, 2t + 1 = j -> i"""
        e = ops.pair_edges_interleaved(pairs, milli, ib, wide)
        return e if wide else (e, None)

    def group_edges_by_owner(self, keys, dst, lo, nql, world):
        if keys.numel() == 0:
            return keys, dst, torch.zeros((world + 1,), dtype=torch.int64, device=keys.device)
        k, d = ops.sort_u64(keys, dst, lo, lo + 1, owner_shard=nql)
        return k, d, ops.owner_bounds(k, lo, nql, world)

    def topk(self, edges, K, id_bits):
        return ops.topk_edges(edges, K, id_bits)

    def topk_local(self, keys, dst, K, ib, q0, nql):
        return ops.topk_edges_local(keys, dst, K, ib, q0, nql)


def _band_major(keys_all, lo, hi):
    """bands [lo, hi) of [world][b][nql] -> band-major [hi - lo][world * nql]"""
    world, _, nql = keys_all.shape
    return keys_all[:, lo:hi, :].permute(1, 0, 2).reshape(hi - lo, world * nql).contiguous()


def _owned_bands(recv, world, nb, nql):
    """[world][nb][nql] (as received) -> band-major [nb][world * nql]"""
    return recv.view(world, nb, nql).permute(1, 0, 2).reshape(nb, world * nql).contiguous()


_BG_GROUPS = {}


def background_group(group=None, force=False):
    """A second communicator over the same ranks for the long signature all-gather, so that it
    runs beside the short exchanges instead of ahead of them (collectives of ONE communicator
    execute in issue order).  Created collectively on first use, then cached.
    force: make the second communicator even for one rank (the RCCL rehearsal of force_collectives)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return group
    key = id(group) if group is not None else 0
    if key not in _BG_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(world))
        _BG_GROUPS[key] = dist.new_group(ranks=ranks)
    return _BG_GROUPS[key]


def shard_range(nq_total, world, rank):
    """Rank `rank` owns the query ids [q0, q0 + n_real) of nq_total; every rank's id space is nql =
    ceil(nq_total / world) wide (the last shards are padded with empty answer sets: their signatures are
    all -1, their band keys the empty key, so they are never candidates).  -> (q0, n_real, nql)"""
    nql = -(-nq_total // world)
    q0 = min(rank * nql, nq_total)
    return q0, min(nql, nq_total - q0), nql


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


def _bytes_view(out, inp):
    """compact signature rows are torch.int16, which RCCL does not move: send them as bytes (2-D [rows, bytes],
    so row-count splits still apply)"""
    if inp.dtype == torch.int16:
        return out.view(torch.uint8), inp.view(torch.uint8)
    return out, inp


def _all_gather(out, inp, group=None, async_op=False):
    out, inp = _bytes_view(out, inp)
    if _staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()
    h = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return h if async_op else _Done()


def _all_to_all(out, inp, osplit=None, isplit=None, group=None):
    out, inp = _bytes_view(out, inp)
    if _staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=osplit, input_split_sizes=isplit, group=group)
        out.copy_(o)
        return
    dist.all_to_all_single(out, inp, output_split_sizes=osplit, input_split_sizes=isplit, group=group)


def _exchange_sizes(bounds, group=None):
    """bounds: int64 [world + 1] split points (on the compute device) of a buffer ordered by destination.
    One small all-to-all and ONE read-back -> (send sizes, receive sizes) as host lists."""
    world = dist.get_world_size(group)
    both = torch.empty((2, world), dtype=torch.int64, device=bounds.device)
    torch.sub(bounds[1:], bounds[:-1], out=both[0])
    _all_to_all(both[1], both[0], group=group)
    s, r = both.tolist()
    return s, r


class _Phases:
    """per-phase milliseconds (events on the compute stream) and bytes sent, accumulated into a caller's dict"""

    def __init__(self, sink, dev):
        self.sink = sink
        self.on = sink is not None and dev.type == "cuda"
        self.marks = []
        if self.on:
            self._mark(None)

    def _mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def done(self, name):
        if self.on:
            self._mark(name)

    def sent(self, name, nbytes):
        if self.sink is not None:
            self.sink["bytes:" + name] = self.sink.get("bytes:" + name, 0) + int(nbytes)

    def close(self):
        if self.sink is None:
            return
        self.sink["_steps"] = self.sink.get("_steps", 0) + 1
        if not self.on:
            return
        torch.cuda.synchronize()
        for (_, a), (name, e) in zip(self.marks[:-1], self.marks[1:]):
            self.sink["ms:" + name] = self.sink.get("ms:" + name, 0.0) + a.elapsed_time(e)


class _Pending:
    """handle of an asynchronous collective that keeps its SEND buffer referenced until wait() returns: the buffers
    are temporaries of the issuing function, and nothing but ProcessGroupNCCL's own bookkeeping would otherwise keep
    the allocator from handing their memory to the next kernel while the collective still reads it"""

    def __init__(self, handle, *buffers):
        self.handle, self.buffers = handle, buffers

    def wait(self):
        if self.handle is not None:
            self.handle.wait()
        self.handle, self.buffers = None, ()


def _gather_answer_sets(offsets, rows, table, nql, world, group, ph, force=False):
    """asynchronous all-gather (second communicator) of every shard's answer sets: row ids padded to the largest
    shard + the nql + 1 offsets.  On the wire the row ids are 16-bit words when the table has at most 65536 rows and
    the offsets 32-bit ones (a shard holds fewer than 2^31 row ids): 34 instead of 72 bytes per query of mean size
    16.  -> (rows [world, max_nnz], offsets [world, nql + 1], narrow, small_off, handle, handle); the handles hold the
    send buffers until they are waited for."""
    dev = offsets.device
    cnt = torch.empty((world,), dtype=torch.int64, device=dev)
    _all_gather(cnt, torch.tensor([rows.numel()], dtype=torch.int64, device=dev), group)
    max_nnz = max(1, int(cnt.max().item()))
    rows_pad = rows if rows.numel() == max_nnz else torch.cat([rows, rows.new_zeros(max_nnz - rows.numel())])
    bg = background_group(group, force)
    narrow = table.D <= 65536
    small_off = max_nnz < (1 << 31)
    rows_w = rows_pad.to(torch.int16) if narrow else rows_pad
    off_w = offsets.to(torch.int32) if small_off else offsets
    ra_w = torch.empty((world, max_nnz), dtype=rows_w.dtype, device=dev)
    oa_w = torch.empty((world, nql + 1), dtype=off_w.dtype, device=dev)
    h_r = _Pending(_all_gather(ra_w, rows_w.view(1, -1), bg, async_op=True), rows_w, rows_pad)
    h_o = _Pending(_all_gather(oa_w, off_w.view(1, -1), bg, async_op=True), off_w)
    ph.sent("0_answer_sets", (rows_w.numel() * rows_w.element_size() + off_w.numel() * off_w.element_size()) * (world - 1))
    return ra_w, oa_w, narrow, small_off, h_r, h_o


def query_similarities_sharded(offsets, rows, table, b, K, nq_total, exchange="all_to_all", backend=None,
                               group=None, wide_ids=None, sig_exchange="auto", phases=None, local_dedup=None,
                               force_collectives=False):
    """Hot path for this rank's query shard (the queries shard_range(nq_total, world, rank) names);
    collective over `group`.  Returns a HotPathResult: sig / norm2 / top-K rows of this rank's queries
    (global ids; concatenated over ranks in rank order they equal the single-GPU result) and the
    candidate pairs this rank scored, sorted (disjoint over ranks; their union is the single-GPU list).
    phases: a dict that accumulates "ms:<phase>" / "bytes:<collective>" over calls (diagnostics).
    local_dedup: de-duplicate the rank's own emissions before the pair exchange (None: up to four ranks).
    force_collectives: with ONE rank, go through every exchange step (collectives that send a rank's data to
    itself, the second communicator, the remote-row machinery over an empty remote set) instead of the one-rank
    short cuts -- a one-GPU box then makes RCCL execute every collective shape the N-rank step issues."""
    be = backend if backend is not None else HipBackend()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    _, n_real, nql = shard_range(nq_total, world, rank)
    if offsets.numel() - 1 != n_real:
        raise ValueError("rank %d of %d owns %d of the %d queries, got %d answer sets"
                         % (rank, world, n_real, nq_total, offsets.numel() - 1))
    if n_real < nql:     # pad the shard with empty answer sets
        offsets = torch.cat([offsets, offsets[-1:].expand(nql - n_real)])
    q0 = rank * nql
    nids = nql * world                       # padded global id space
    be.rows_hint = nql
    P = table.P
    if P % b != 0:
        raise AssertionError("signature length %d not divisible by b=%d" % (P, b))
    if exchange not in ("all_to_all", "all_gather"):
        raise ValueError("exchange must be 'all_to_all' or 'all_gather'")
    if sig_exchange not in ("auto", "fetch", "all_gather", "recompute", "sets"):
        raise ValueError("sig_exchange must be 'auto', 'fetch', 'sets', 'all_gather' or 'recompute'")
    r = P // b
    ib = ops.id_bits_for(nids)
    wide = ops.wide_ids(ib) if wide_ids is None else wide_ids
    dev = offsets.device
    stats = {}
    ph = _Phases(phases, dev)
    multi = world > 1 or bool(force_collectives)      # take the exchange steps

    if sig_exchange == "auto":
        # few ranks = few links, and everything a rank sends to one peer crosses ONE of them: up to four ranks the
        # answer sets are replicated instead (64 B per query, against 256 B of signature row per scored pair plus
        # 8 B per band of bucket ids) and every rank computes all signatures; beyond, shards stay shards
        # (... and from five ranks on the answer sets are still replicated -- in the background, 34 B per query -- but
        # only the signatures a rank's pairs need are computed from them: no row fetch)
        sig_exchange = "recompute" if 2 <= world <= 4 else ("sets" if world > 4 else "fetch")
    ranges = band_owner_ranges(b, world)
    lo, hi = ranges[rank]
    nb = hi - lo
    keys_all = None
    if sig_exchange == "recompute" and multi:
        # 0. answer sets of every shard: the offsets (nql + 1 each) and the row ids, padded to the largest shard
        ra_w, oa_w, narrow, small_off, h_r, h_o = _gather_answer_sets(offsets, rows, table, nql, world, group, ph,
                                                                      force_collectives)
        # 1. signatures of ALL queries, own shard first (it runs beside the gather); every block goes straight to
        #    its place in the replicated tables when the row blocks keep the kernel's 16-byte alignment
        sdt = be.sig_dtype(table)
        esz = 2 if sdt == torch.int16 else 4
        in_place = (nql * P * esz) % 16 == 0
        sa = torch.empty((nids, P), dtype=sdt, device=dev)
        na = torch.empty((nids,), dtype=torch.int64, device=dev)
        keys_all = torch.empty((world, b, nql), dtype=torch.int64, device=dev)

        def shard_signatures(g, off_g, rows_g, validate):
            blk = slice(g * nql, (g + 1) * nql)
            if in_place:
                be.minhash(off_g, rows_g, table, b, out=(sa[blk], na[blk], keys_all[g]), validate=validate)
            else:
                s_g, n_g, k_g = be.minhash(off_g, rows_g, table, b, validate=validate)
                sa[blk].copy_(s_g)
                na[blk].copy_(n_g)
                keys_all[g].copy_(k_g)
        shard_signatures(rank, offsets, rows, None)
        h_r.wait()
        h_o.wait()
        for g in range(world):
            if g != rank:
                ra_g = (ra_w[g].to(torch.int32) & 0xFFFF) if narrow else ra_w[g]
                oa_g = oa_w[g].to(torch.int64) if small_off else oa_w[g]
                shard_signatures(g, oa_g, ra_g, False)      # validated by their owner
                del ra_g, oa_g
        del ra_w, oa_w
        sig, norm2 = sa[q0:q0 + nql], na[q0:q0 + nql]
        keys = None
        stats["bucket_id_exchange"] = "none (answer sets replicated)"
    else:
        sets = None
        if sig_exchange == "sets" and multi:
            # 0'. the same gather, in the background: what it brings is first needed at step 5
            sets = _gather_answer_sets(offsets, rows, table, nql, world, group, ph, force_collectives)
        # 1. local signatures
        sig, norm2, keys = be.minhash(offsets, rows, table, b)
    ph.done("1_minhash")

    # 2. bucket-id exchange (short, needed at once: issued before the long gather)
    if keys_all is not None:
        recv = owned = None
    elif not multi:
        recv, owned = keys, None  # nothing to exchange: the local keys are the [1][b][nql] buffer
    elif exchange == "all_gather":
        allk = torch.empty((world * b, nql), dtype=torch.int64, device=dev)
        _all_gather(allk, keys, group)
        owned = allk.view(world, b, nql)[:, lo:hi, :].permute(1, 0, 2).reshape(nb, nids).contiguous()
        del allk
        ph.sent("2_bucket_ids", keys.numel() * 8 * (world - 1))
    else:
        in_split = [h - l for (l, h) in ranges]
        recv = torch.empty((world * nb, nql), dtype=torch.int64, device=dev)
        _all_to_all(recv, keys, [nb] * world, in_split, group)
        owned = None            # read in place by emit_pairs_chunked
        ph.sent("2_bucket_ids", (b - nb) * nql * 8)
    del keys
    ph.done("2_bucket_id_exchange")

    # "all_gather" of the signature rows + norms: asynchronous, on the background communicator
    gathered = None
    if keys_all is not None:
        gathered = (sa, na, _Done(), _Done())
    elif sig_exchange == "all_gather" and multi:
        bg = background_group(group, force_collectives)
        sa = torch.empty((nids, P), dtype=sig.dtype, device=dev)
        na = torch.empty((nids,), dtype=torch.int64, device=dev)
        gathered = (sa, na, _Pending(_all_gather(sa, sig, bg, async_op=True), sig),
                    _Pending(_all_gather(na, norm2, bg, async_op=True), norm2))
        ph.sent("5_signature_rows", (sig.numel() * sig.element_size() + norm2.numel() * 8) * (world - 1))
    stats["sig_exchange"] = sig_exchange

    # 3. candidates of the owned bands over all queries
    if nb > 0 and keys_all is not None:
        emitted = be.emit_pairs_bands(keys_all, lo, hi, r)
    elif nb > 0:
        emitted = be.emit_pairs(owned, r) if owned is not None else be.emit_pairs_chunked(recv, world, nb, nql, r)
    else:
        emitted = torch.empty((0,), dtype=torch.int64, device=dev)
    stats["emitted_pairs"] = int(emitted.numel())
    owned = recv = keys_all = None
    ph.done("3_bucket_pairs")

    # 4. pairs -> the rank that scores them.  With many ranks (few bands each) the duplicates across this rank's own
    #    bands are left in -- the scoring rank de-duplicates anyway and a local unique would cost more passes than the
    #    bytes it saves; with up to four ranks (8+ of 32 bands each, ONE to three links to push everything through) a
    #    pair that collides in several of the rank's bands is sent once: the rank's emissions are de-duplicated first
    #    (375 -> ~140 MB out per rank and step at two ranks, 10 M queries)
    if local_dedup is None:
        local_dedup = 1 < world <= 4
    if multi and local_dedup and emitted.numel():
        emitted = be.sort_unique(emitted, nids, words_per_query=emitted.numel() / max(1, nids))
        stats["local_unique_pairs"] = int(emitted.numel())
    if multi:
        mine, bounds = be.group_pairs_by_host(emitted, nql, world)
        ssz, rsz = _exchange_sizes(bounds, group)
        got = torch.empty((sum(rsz),), dtype=torch.int64, device=dev)
        _all_to_all(got, mine, rsz, ssz, group)
        ph.sent("4_pairs", (sum(ssz) - ssz[rank]) * 8)
        del mine
    else:
        got = emitted
    del emitted
    ph.done("4_pair_exchange")
    pairs = be.sort_unique(got, nids) if got.numel() else got
    del got
    ph.done("4_pair_unique")

    # 5. the rows the pairs need
    sig_b = norm_b = None
    if gathered is not None:
        sa, na, h_sig, h_nrm = gathered
        h_sig.wait()
        h_nrm.wait()
        score_sig, score_norm, local_pairs = sa, na, pairs          # row index == global query id
    elif not multi:
        score_sig, score_norm, local_pairs = sig, norm2, pairs
        stats["remote_rows_fetched"] = 0
    elif sig_exchange == "sets":
        # the remote queries this rank's pairs touch: their answer sets are here already (step 0'), their signatures
        # are computed now -- one MinHash over exactly the rows a fetch would have asked the owners for
        ra_w, oa_w, narrow, _, h_r, h_o = sets
        rid = be.remote_ids(pairs, q0, nql, nids, world)
        need = be.remote_id_list(rid, int(rid.bounds[-1].item()))
        h_r.wait()
        h_o.wait()
        off_b, rows_b = be.gather_sets(need, oa_w, ra_w, nql)
        if need.numel():
            sig_b, norm_b, _ = be.minhash(off_b, rows_b, table, b, validate=False, keys=False)
        else:
            sig_b, norm_b = sig[:0], norm2[:0]
        stats["remote_signatures_computed"] = int(need.numel())
        score_sig, score_norm = sig, norm2
        local_pairs = be.remap_pairs(pairs, rid)
        del rid, need, off_b, rows_b, ra_w, oa_w, sets
    else:
        rid = be.remote_ids(pairs, q0, nql, nids, world)
        ssz, rsz = _exchange_sizes(rid.bounds, group)                # ids I request / ids requested from me
        need = be.remote_id_list(rid, sum(ssz))
        req = torch.empty((sum(rsz),), dtype=torch.int64, device=dev)
        _all_to_all(req, need, rsz, ssz, group)
        out_rows, out_norms = be.gather_rows(sig, norm2, req, q0)
        sig_b = torch.empty((need.numel(), P), dtype=sig.dtype, device=dev)
        norm_b = torch.empty((need.numel(),), dtype=torch.int64, device=dev)
        _all_to_all(sig_b, out_rows, ssz, rsz, group)
        _all_to_all(norm_b, out_norms, ssz, rsz, group)
        ph.sent("5_row_requests", sum(ssz) * 8)
        ph.sent("5_signature_rows", sum(rsz) * (P * sig.element_size() + 8))
        stats["remote_rows_fetched"] = int(need.numel())
        score_sig, score_norm = sig, norm2
        local_pairs = be.remap_pairs(pairs, rid)                     # both halves index [local rows | fetched rows]
        del rid, need, req, out_rows, out_norms
    ph.done("5_signature_rows")

    # 6. score; edges -> owner of their src
    if r > 4 and pairs.numel():   # wide bands: hashed bucket ids -> exact verification on the scoring rank
        table_rows = score_sig if sig_b is None else torch.cat([score_sig, sig_b])
        keep = be.verify_flags(table_rows, b, local_pairs).bool()
        del table_rows
        if not bool(keep.all()):
            pairs, local_pairs = pairs[keep], local_pairs[keep]
    milli = be.score(score_sig, score_norm, sig_b, norm_b, local_pairs)
    del local_pairs, sig_b, norm_b, gathered
    ph.done("6_score")
    ek, ed = be.edges(pairs, milli, ib, wide)
    if multi:
        klo = 11 if wide else ib + 11
        ek, ed, bounds = be.group_edges_by_owner(ek, ed, klo, nql, world)
        ssz, rsz = _exchange_sizes(bounds, group)
        ein = torch.empty((sum(rsz),), dtype=torch.int64, device=dev)
        _all_to_all(ein, ek, rsz, ssz, group)
        din = None
        if ed is not None:
            din = torch.empty((sum(rsz),), dtype=torch.int32, device=dev)
            _all_to_all(din, ed, rsz, ssz, group)
        ph.sent("6_edges", (sum(ssz) - ssz[rank]) * (12 if wide else 8))
        ph.done("6_edge_exchange")
        # 7. edges of one query now come from several scoring ranks: full-key sort on re-based keys
        src, dst, val = be.topk_local(ein, din, K, ib, q0, nql)
    else:
        # one scoring rank: the edges are in pair order, the stable (src, value) sort of the one-GPU path applies
        ph.done("6_edge_exchange")
        src, dst, val = be.topk((ek, ed) if wide else ek, K, ib)
    ph.done("7_topk")
    stats["unique_pairs"] = int(pairs.numel())
    stats["kept_edges"] = int(src.numel())
    stats["topk"] = "select (received edges = reverse words, sorted on their src bits)" if multi else "sort-stable"
    stats.update(getattr(be, "stats", {}))
    ph.close()
    return HotPathResult(sig[:n_real], norm2[:n_real], pairs, milli, src, dst, val, K, b, stats)
