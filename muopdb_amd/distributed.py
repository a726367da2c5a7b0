"""Multi-GPU search: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm).

Partitioning
  * IVF / SPANN / multi-user SPANN: every rank loads the same files with (shard_rank, shard_world);
    posting list l of every user of a multi-user collection is owned by rank l % world, the lists of a single IVF
    index (C5) are dealt size-balanced (`balanced_owners`: longest first to the least loaded rank), centroids / graphs / doc-id tables are
    replicated, so probe selection is identical on all ranks.
  * HNSW: the traversal does not partition (replicas only): ranks split the batch, no collective.
  * flat: row-range shards; ids are global rows, so the (distance, row) merge of the per-shard rows is exact.

The step of a list-sharded index is EXACT (PointsGather):
  1. every rank runs search_with_centroids (rs/index/src/ivf/block_based/index.rs:250-286) over the lists it owns and writes its
     k smallest (distance, POINT id) rows — not remapped — straight into its POINTS block (mdb_*_search_shard;
     { u32 point_ids[b][k]; f32 scores[b][k]; u32 counts[b]; u8 found[b] }: (8 k + 5) bytes per query per rank);
  2. ONE all-gather of the preallocated blocks: latency-bound, far below a single xGMI link's bandwidth, so a direct
     all-gather (not a ring pipeline) is the right shape;
  3. every rank merges on the device (mdb_*_merge_shards): the k smallest of the union by (distance, point id) — which is the
     reference's heap over all probed lists, since each list lives on one rank — and only THEN doc ids and the IdWithScore
     (score, doc id) order (search_with_centroids_and_remap :298-332).
  Row for row the unsharded result, including score ties at rank k under doc ids that are not monotone in point ids
  (reindexed segments) and duplicate PQ codes (tests/test_gpu_traversal.py::test_sharded_merge_is_exact_under_ties).
  A host without torch issues step 2 through the C ABI: mdb_allgather_blocks(ctx, ncclComm_t, ...).
  The role replaced is the aggregator's fan-out (rs/aggregator/src/aggregator.rs:80-135).

PackedTopkGather / mdb_merge_shards (IdWithScore merge of already remapped rows) remain for rows of DIFFERENT indexes —
flat row shards and the segments of a snapshot (Snapshot::search_for_users, rs/index/src/collection/snapshot.rs:60-63) — where
(score, doc id) IS the reference's merge rule.
"""
import ctypes as C

import torch
import torch.distributed as dist


class StepTimer:
    """Device time of the exchange steps of a sharded search (bench.py --gpus N: the all-gathers' own time and the merge kernels'):
    event pairs on the current stream around each collective / merge call, resolved once at the end (`ms()` synchronises).
    Off (the default) it costs nothing: `TIMER = None`."""

    def __init__(self):
        self.pairs = []

    def span(self, kind):
        return _Span(self, kind)

    def ms(self):
        torch.cuda.synchronize()
        out = {}
        for kind, e0, e1 in self.pairs:
            tot, cnt = out.get(kind, (0.0, 0))
            out[kind] = (tot + e0.elapsed_time(e1), cnt + 1)
        self.pairs = []
        return {kind: dict(total_ms=t, calls=c, ms_per_call=t / max(c, 1)) for kind, (t, c) in out.items()}


class _Span:
    def __init__(self, timer, kind):
        self.t, self.kind = timer, kind

    def __enter__(self):
        if self.t is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.t.pairs.append((self.kind, self.e0, e1))
        return False


TIMER = None   # a StepTimer while bench.py measures the exchange steps


def _span(kind):
    return _Span(TIMER, kind)


def shard_of_list(list_index, world):
    """Owner rank of posting list `list_index` of every user of a MULTI-USER collection (mdb_multi_spann_load)."""
    return list_index % world


def balanced_owners(list_sizes, world):
    """Owner rank of every posting list of ONE index (mdb_ivf_load with shard_world > 1): size-balanced greedy — lists
    longest first (ties: lower index), each to the least loaded rank (ties: lower rank).  Same rule as the library, so a
    host can tell which rank holds a list without asking."""
    order = sorted(range(len(list_sizes)), key=lambda l: (-int(list_sizes[l]), l))
    load = [0] * world
    owner = [0] * len(list_sizes)
    for l in order:
        r = min(range(world), key=lambda i: (load[i], i))
        owner[l] = r
        load[r] += int(list_sizes[l])
    return owner


def split_batch(b, rank, world):
    """Contiguous slice [lo, hi) of a batch of b queries for replica-parallel search."""
    return rank * b // world, (rank + 1) * b // world


def block_bytes(b, k):
    """bytes of one rank's packed result block (mdb_shard_block_bytes): ids [b][k] u128 | scores [b][k] f32 | counts [b] u32 | pad 16"""
    return (b * k * 20 + b * 4 + 15) // 16 * 16


def block_views(block, b, k):
    """typed views INTO a uint8 block tensor: (doc ids int64 [b,k,2] (lo, hi), scores f32 [b,k], counts int32 [b])"""
    ids = block[:b * k * 16].view(torch.int64).view(b, k, 2)
    scores = block[b * k * 16:b * k * 20].view(torch.float32).view(b, k)
    counts = block[b * k * 20:b * k * 20 + b * 4].view(torch.int32)
    return ids, scores, counts


class PackedTopkGather:
    """Exchange + IdWithScore merge of already remapped rows (flat row shards, segments of a snapshot — NOT list shards of one
    index: PointsGather): ONE all-gather per batch of one preallocated packed block per rank, then the device merge.  The search writes its outputs straight into this rank's send block (`ids`, `scores`,
    `counts` are views of it), so the step allocates nothing and repacks nothing:

        g = PackedTopkGather(ctx, b, k, "cuda")
        mdb_*_search(..., g.ids.data_ptr(), g.scores.data_ptr(), g.counts.data_ptr(), ...)
        docs, scores, counts = g.gather_merge()          # [b,k,2], [b,k], [b] on every rank

    `ctx` None (CPU / gloo plumbing tests): gather only, the caller merges `recv_views()` itself."""

    def __init__(self, ctx, b, k, device, group=None):
        self.ctx, self.b, self.k, self.group = ctx, b, k, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        nb = block_bytes(b, k)
        self.send = torch.zeros(nb, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * nb, dtype=torch.uint8, device=device)
        self.ids, self.scores, self.counts = block_views(self.send, b, k)
        self.out_docs = torch.zeros((b, k, 2), dtype=torch.int64, device=device)
        self.out_scores = torch.zeros((b, k), dtype=torch.float32, device=device)
        self.out_counts = torch.zeros(b, dtype=torch.int32, device=device)

    def gather(self):
        if self.world == 1:
            self.recv.copy_(self.send)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)  # rank-major blocks
        return self.recv

    def recv_views(self):
        nb = block_bytes(self.b, self.k)
        return [block_views(self.recv[w * nb:(w + 1) * nb], self.b, self.k) for w in range(self.world)]

    def gather_merge(self):
        self.gather()
        c = self.ctx
        c.check(c.lib.mdb_merge_shards_packed(c.h, C.c_void_p(self.recv.data_ptr()), C.c_size_t(self.world), C.c_size_t(self.b),
                                              C.c_size_t(self.k), C.c_void_p(self.out_docs.data_ptr()),
                                              C.c_void_p(self.out_scores.data_ptr()), C.c_void_p(self.out_counts.data_ptr())))
        return self.out_docs, self.out_scores, self.out_counts



def points_block_bytes(b, k):
    """bytes of one rank's POINTS block (mdb_points_block_bytes): point ids [b][k] u32 | scores [b][k] f32 | counts [b] u32 | found [b] u8 | pad 16"""
    return (b * k * 8 + b * 4 + b + 15) // 16 * 16


def points_block_views(block, b, k):
    """typed views INTO a uint8 points block: (point ids int32 [b,k], scores f32 [b,k], counts int32 [b], found uint8 [b])"""
    pids = block[:b * k * 4].view(torch.int32).view(b, k)
    scores = block[b * k * 4:b * k * 8].view(torch.float32).view(b, k)
    counts = block[b * k * 8:b * k * 8 + b * 4].view(torch.int32)
    found = block[b * k * 8 + b * 4:b * k * 8 + b * 5]
    return pids, scores, counts, found


class PointsGather:
    """The EXACT sharded step (module docstring): search_shard into `send`, ONE all-gather, merge on the device.

        g = PointsGather(ctx, b, k, "cuda")
        mdb_ivf_search_shard(ivf, ..., MDB_MEM_DEVICE, ..., g.send.data_ptr())
        docs, scores, counts = g.gather_merge_ivf(ivf)                   # [b,k,2], [b,k], [b] on every rank
      (SPANN: mdb_spann_search_shard / g.gather_merge_spann(spann); multi-user: g.gather_merge_multi(ms, user_ids))

    `ctx` None (CPU / gloo plumbing tests): gather only, the caller merges `recv_views()` itself."""

    def __init__(self, ctx, b, k, device, group=None):
        self.ctx, self.b, self.k, self.group = ctx, b, k, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.nb = points_block_bytes(b, k)
        self.send = torch.zeros(self.nb, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * self.nb, dtype=torch.uint8, device=device)
        self.out_docs = torch.zeros((b, max(k, 1), 2), dtype=torch.int64, device=device)
        self.out_scores = torch.zeros((b, max(k, 1)), dtype=torch.float32, device=device)
        self.out_counts = torch.zeros(b, dtype=torch.int32, device=device)
        self.out_found = torch.ones(b, dtype=torch.uint8, device=device)

    def gather(self):
        if self.world == 1:
            self.recv.copy_(self.send)
        else:
            with _span("points_allgather"):
                dist.all_gather_into_tensor(self.recv, self.send, group=self.group)  # rank-major blocks
        return self.recv

    def recv_views(self):
        return [points_block_views(self.recv[w * self.nb:(w + 1) * self.nb], self.b, self.k) for w in range(self.world)]

    def _tail(self, with_found):
        a = [C.c_void_p(self.recv.data_ptr()), C.c_size_t(self.world), C.c_size_t(self.b), C.c_size_t(self.k),
             C.c_void_p(self.out_docs.data_ptr()), C.c_void_p(self.out_scores.data_ptr()), C.c_void_p(self.out_counts.data_ptr())]
        if with_found:
            a.append(C.c_void_p(self.out_found.data_ptr()))
        return a

    def gather_merge_ivf(self, ivf):
        self.gather()
        with _span("merge_points"):
            self.ctx.check(self.ctx.lib.mdb_ivf_merge_shards(ivf.h, *self._tail(False)))
        return self.out_docs, self.out_scores, self.out_counts

    def gather_merge_spann(self, spann):
        self.gather()
        with _span("merge_points"):
            self.ctx.check(self.ctx.lib.mdb_spann_merge_shards(spann.h, *self._tail(True)))
        return self.out_docs, self.out_scores, self.out_counts

    def gather_merge_multi(self, ms, user_ids_c):
        """user_ids_c: the same ctypes U128 array the search was called with"""
        self.gather()
        with _span("merge_points"):
            self.ctx.check(self.ctx.lib.mdb_multi_spann_merge_shards(ms.h, user_ids_c, *self._tail(True)))
        return self.out_docs, self.out_scores, self.out_counts

class ProbeRowsShare:
    """List-sharded multi-user SPANN with the centroid stage NOT replicated: rank r runs `centroids.ann_search` + the ratio filter
    (spann/index.rs:211-246) for ITS slice of the batch only, the probe rows meet in one all-gather, every rank scans its lists for
    the whole batch from the gathered table (include/muopdb_hip.h: mdb_multi_spann_probes / mdb_multi_spann_search_shard_probes).

        sh = ProbeRowsShare(ctx, b, row_words, "cuda")           # row_words = mdb_spann_probe_row_words(params)
        lo, hi = sh.slice
        mdb_multi_spann_probes(ms, user_ids[lo:hi], queries[lo:hi], hi - lo, params, MDB_MEM_DEVICE, sh.send.data_ptr())
        rows = sh.gather()                                        # [b][row_words] on every rank, identical
        mdb_multi_spann_search_shard_probes(ms, user_ids, queries, b, params, MDB_MEM_DEVICE, rows.data_ptr(), ..., g.send.data_ptr())

    Slices are `per` = ceil(b / world) rows each (the last ranks' may be short or empty): rank-major blocks of an all-gather are
    then the batch's table in batch order, no permutation; the padding rows of a short slice stay zero (count 0, found 0) and lie
    beyond row b.  When it pays: the closure kernel is one wave per pair, so its time is flat up to a few hundred pairs and grows
    with the batch beyond; replicating it on every rank costs nothing at batch 128 and most of the step at batch 1024 x 8 ranks."""

    def __init__(self, ctx, b, row_words, device, group=None):
        self.ctx, self.b, self.row_words, self.group = ctx, b, row_words, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.per = (b + self.world - 1) // self.world
        lo = min(self.rank * self.per, b)
        self.slice = (lo, min(lo + self.per, b))
        self.send = torch.zeros((max(self.per, 1), row_words), dtype=torch.int32, device=device)
        self.rows = torch.zeros((self.world * max(self.per, 1), row_words), dtype=torch.int32, device=device)

    def gather(self):
        if self.world == 1:
            self.rows.copy_(self.send)
        else:
            with _span("probes_allgather"):
                dist.all_gather_into_tensor(self.rows, self.send, group=self.group)  # rank-major slices == batch order
        return self.rows


# ------------------------------------------------------------------------------------------ query partitionings (no merge)
# SURVEY.md §8e asks for BOTH partitionings to be measured: posting-list shards (above: every rank sees every query, one all-gather
# of points blocks + an exact merge) and QUERY partitionings, where a rank answers a disjoint subset of the batch whole and the only
# exchange is the finished result rows:
#   * by user  (multi-user collections): user slot u (its record in the user table) lives on rank u % world — indexes, graphs and
#     lists of that user only; a (user, query) pair is routed to its owner (the aggregator's role: rs/aggregator/src/aggregator.rs:80-135
#     fans a request out to the nodes that hold the shard);
#   * by batch (an index that fits one GPU many times over, C3 / C5): replicas, rank r takes the contiguous slice split_batch(b, r, world).
# Either way the rows a rank produced ARE the reference's rows for those queries (no cross-rank merge exists to get wrong).

def user_owner(user_slot, world):
    """rank that holds user slot `user_slot` (index of the user's record in the collection's user table)"""
    return int(user_slot) % world


def users_of_rank(n_users, rank, world):
    """user slots a rank loads under user sharding"""
    return list(range(rank, n_users, world))


def route_by_user(user_slots, world):
    """positions of a batch's (user, query) pairs per owning rank: [world] lists of batch positions, in batch order"""
    out = [[] for _ in range(world)]
    for i, u in enumerate(user_slots):
        out[user_owner(u, world)].append(i)
    return out


def route_by_batch(b, world):
    """contiguous slices of a batch of b queries, as position lists (split_batch)"""
    return [list(range(*split_batch(b, r, world))) for r in range(world)]


def rows_block_bytes(b, k):
    """one rank's block of finished result rows: doc ids [b][k] u128 | scores [b][k] f32 | counts [b] u32 | found [b] u8 | pad 16"""
    return (b * k * 20 + b * 5 + 15) // 16 * 16


class RowsExchange:
    """Exchange of FINISHED result rows of a query partitioning: one all-gather of fixed-size blocks per batch, then a gather by a
    precomputed permutation into batch order — no merge.  `routes` = positions per rank (route_by_user / route_by_batch); the
    search of this rank's `len(routes[rank])` queries writes straight into the views `ids`, `scores`, `counts`, `found`.

        ex = RowsExchange(b, k, routes, rank, "cuda")
        mdb_*_search(..., n_local, ..., ex.ids.data_ptr(), ex.scores.data_ptr(), ex.counts.data_ptr(), ex.found.data_ptr())
        docs, scores, counts, found = ex.gather()          # [b,k,2] int64 (lo, hi), [b,k], [b], [b] in batch order on every rank
    """

    def __init__(self, b, k, routes, rank, device, group=None):
        self.b, self.k, self.group, self.rank = b, k, group, rank
        self.world = len(routes)
        seen = sorted(i for r in routes for i in r)
        if seen != list(range(b)):
            raise ValueError("RowsExchange: the routes must cover every query of the batch exactly once")
        self.bmax = max(1, max(len(r) for r in routes))
        self.n_local = len(routes[rank])
        self.blk = rows_block_bytes(self.bmax, k)
        self.send = torch.zeros(self.blk, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * self.blk, dtype=torch.uint8, device=device)
        self.ids, self.scores, self.counts, self.found = self._views(self.send)
        perm = [0] * b
        for r, pos in enumerate(routes):
            for j, i in enumerate(pos):
                perm[i] = r * self.bmax + j
        self.perm = torch.tensor(perm, dtype=torch.int64, device=device)
        self.local = torch.tensor(routes[rank], dtype=torch.int64, device=device)
        self.lo = routes[rank][0] if routes[rank] else 0
        self.contiguous = routes[rank] == list(range(self.lo, self.lo + self.n_local))

    def _views(self, block):
        b, k = self.bmax, self.k
        ids = block[:b * k * 16].view(torch.int64).view(b, k, 2)
        scores = block[b * k * 16:b * k * 20].view(torch.float32).view(b, k)
        counts = block[b * k * 20:b * k * 20 + b * 4].view(torch.int32)
        found = block[b * k * 20 + b * 4:b * k * 20 + b * 5]
        return ids, scores, counts, found

    def local_queries(self, q):
        """this rank's rows of the batch's query matrix (a view when the route is a contiguous slice)"""
        if self.n_local == 0:
            return q[:0]
        if self.contiguous:
            return q[self.lo:self.lo + self.n_local]
        return q.index_select(0, self.local)

    def gather(self):
        with _span("rows_allgather"):
            if self.world > 1:
                dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            else:
                self.recv.copy_(self.send)
        return self.permute()

    def permute(self):
        """`recv` (the ranks' blocks, rank-major) -> rows in batch order.  Separate from gather() so that a single process can stand in
        for several ranks (tests: the blocks of simulated ranks copied into `recv`)."""
        with _span("rows_permute"):
            b, k, w = self.bmax, self.k, self.world
            blocks = self.recv.view(w, self.blk)
            ids = blocks[:, :b * k * 16].contiguous().view(torch.int64).view(w * b, k, 2).index_select(0, self.perm)
            scores = blocks[:, b * k * 16:b * k * 20].contiguous().view(torch.float32).view(w * b, k).index_select(0, self.perm)
            counts = blocks[:, b * k * 20:b * k * 20 + b * 4].contiguous().view(torch.int32).view(w * b).index_select(0, self.perm)
            found = blocks[:, b * k * 20 + b * 4:b * k * 20 + b * 5].contiguous().view(w * b).index_select(0, self.perm)
        return ids, scores, counts, found


def all_gather_topk(doc_ids, scores, counts, group=None):
    """Unpacked variant (three collectives; kept for callers that hold three separate tensors — the packed class above is
    the step's path): ([W,B,k,2], [W,B,k], [W,B]) on every rank."""
    world = dist.get_world_size(group)

    def gather(t):
        t = t.contiguous()
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)  # rank-major concatenation along dim 0
        return out.view((world,) + tuple(t.shape))

    return gather(doc_ids), gather(scores), gather(counts)


def merge_shards_device(ctx, gd, gs, gc, out_docs=None, out_scores=None, out_counts=None):
    """mdb_merge_shards on HBM-resident gathered blocks; returns (docs [B,k,2], scores [B,k], counts [B])."""
    world, b, k = gs.shape
    dev = gs.device
    out_docs = out_docs if out_docs is not None else torch.empty((b, k, 2), dtype=torch.int64, device=dev)
    out_scores = out_scores if out_scores is not None else torch.empty((b, k), dtype=torch.float32, device=dev)
    out_counts = out_counts if out_counts is not None else torch.empty(b, dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.mdb_merge_shards(ctx.h, C.c_void_p(gd.data_ptr()), C.c_void_p(gs.data_ptr()),
                                       C.c_void_p(gc.data_ptr()), C.c_size_t(world), C.c_size_t(b), C.c_size_t(k),
                                       C.c_void_p(out_docs.data_ptr()), C.c_void_p(out_scores.data_ptr()),
                                       C.c_void_p(out_counts.data_ptr())))
    return out_docs, out_scores, out_counts


def sharded_search(local_search, merge, group=None):
    """local_search() -> (docs, scores, counts) of this rank's shard; gathers and merges on every rank.
    `merge(gd, gs, gc)` is merge_shards_device bound to a context on GPUs."""
    docs, scores, counts = local_search()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return docs, scores, counts
    gd, gs, gc = all_gather_topk(docs, scores, counts, group)
    return merge(gd, gs, gc)


def coarse_range(num_clusters, rank, world):
    """Centroid range [first, first + count) of `rank` for the sharded coarse search: whole tiles of 64 centroids
    (`first` is always a multiple of 64; trailing ranks may get an empty range)."""
    per = ((num_clusters + world - 1) // world + 63) // 64 * 64
    first = rank * per
    if first >= num_clusters:
        return (num_clusters // 64) * 64, 0
    return first, min(per, num_clusters - first)


def gather_coarse_rows(keys, group=None):
    """This rank's coarse rows [b][P] (int64 bit patterns of the u64 keys) -> every rank's, laid out [b][world][P] (what
    mdb_ivf_merge_coarse_keys takes)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return keys.view(keys.shape[0], 1, keys.shape[1])
    world = dist.get_world_size(group)
    b, p = keys.shape
    allk = torch.empty((world * b, p), dtype=keys.dtype, device=keys.device)
    with _span("coarse_allgather"):
        dist.all_gather_into_tensor(allk, keys.contiguous(), group=group)  # rank-major
    return allk.view(world, b, p).permute(1, 0, 2).contiguous()


def sharded_probes(ctx, ivf, q_ptr, b, num_probes, device, group=None):
    """find_nearest_centroids with the coarse quantizer SHARDED over the ranks (every rank holds all centroids, but scans
    only its 1/world of them): local (distance, id) keys -> one all-gather of [b][num_probes] u64 per rank -> merge on every
    rank -> probe ids [b][num_probes] int32 on `device`, identical on all ranks and to the unsharded search."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = coarse_range(ivf.num_clusters(), rank, world)
    keys = torch.empty((b, num_probes), dtype=torch.int64, device=device)
    ctx.check(ctx.lib.mdb_ivf_coarse_keys(ivf.h, C.c_void_p(q_ptr), C.c_size_t(b), C.c_size_t(num_probes), C.c_size_t(first),
                                          C.c_size_t(count), C.c_int(1), C.c_void_p(keys.data_ptr())))
    keys = gather_coarse_rows(keys, group)  # [b][world][P]
    probes = torch.empty((b, num_probes), dtype=torch.int32, device=device)
    with _span("merge_coarse"):
        ctx.check(ctx.lib.mdb_ivf_merge_coarse_keys(ivf.h, C.c_void_p(keys.data_ptr()), C.c_size_t(b), C.c_size_t(world), C.c_size_t(num_probes),
                                                    C.c_int(1), C.c_void_p(probes.data_ptr())))
    return probes
