"""Multi-GPU search: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm).

Partitioning
  * IVF / SPANN / multi-user SPANN: every rank loads the same files with (shard_rank, shard_world);
    posting list l of every user of a multi-user collection is owned by rank l % world, the lists of a single IVF
    index (C5) are dealt size-balanced (`balanced_owners`: longest first to the least loaded rank), centroids / graphs / doc-id tables are
    replicated, so probe selection is identical on all ranks and the union of the per-rank top-k
    equals the single-GPU result exactly — with ONE caveat at exact score ties on the k-th boundary: a rank
    (like the unsharded path) selects its top-k by (distance, POINT id) and only then re-ranks by
    (score, DOC id), while the cross-rank merge sees doc ids only.  When several candidates tie exactly at
    rank k and doc ids are not monotone in point ids (a reindexed segment), the merged row may keep a
    different one of the tied documents than the unsharded search does (same scores, same count).  The
    reference's own cross-segment merge (Snapshot::search_for_user, collection/snapshot.rs:69-110) has the
    same property: it too merges per-segment rows by (score, doc id).
  * HNSW: the traversal does not partition (replicas only): ranks split the batch, no collective.
  * flat: row-range shards, same gather + merge.

Collective: ONE all-gather per batch of one packed, preallocated block per rank — doc ids [B][k] (2 x u64),
scores [B][k] f32, counts [B] (PackedTopkGather; (20*k + 4) bytes per query per rank): latency-bound, far below a
single xGMI link's bandwidth, so a direct all-gather (not a ring pipeline) is the right shape.  A host without
torch issues the same exchange through the C ABI: mdb_allgather_merge(ctx, ncclComm_t, ...).
Merge: per query, the `world` sorted rows are merged by IdWithScore order (score, doc id) and
truncated to k — Snapshot::search_for_users' rule (rs/index/src/collection/snapshot.rs:60-63), not
the aggregator's descending sort (rs/aggregator/src/aggregator.rs:135).
"""
import ctypes as C

import torch
import torch.distributed as dist


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
    """The sharded step's exchange: ONE all-gather per batch of one preallocated packed block per rank,
    then the device merge.  The search writes its outputs straight into this rank's send block (`ids`, `scores`,
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
    ctx.check(ctx.lib.mdb_ivf_merge_coarse_keys(ivf.h, C.c_void_p(keys.data_ptr()), C.c_size_t(b), C.c_size_t(world), C.c_size_t(num_probes),
                                                C.c_int(1), C.c_void_p(probes.data_ptr())))
    return probes
