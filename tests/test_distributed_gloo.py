"""world_size-2 `gloo` tests (CPU) of the multi-GPU host path: list sharding,
the per-batch all-gather of fixed-size blocks, the EXACT merge of points blocks ((distance, point id), then remap) and the
(score, doc id) merge of rows from different indexes.  The per-shard
results come from the CPU oracle here (the GPU kernels are covered by the -m gpu suite, including
`test_merge_shards_device` and the shard-union tests); what is under test is the collective plumbing
and that the union of per-rank top-k equals the unsharded answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from muopdb_amd import distributed as D
from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def merge_numpy(gd, gs, gc):
    """Host restatement of mdb_merge_shards (IdWithScore order, truncate to k)."""
    gd, gs, gc = gd.numpy().view(np.uint64), gs.numpy(), gc.numpy()
    world, b, k = gs.shape
    od = np.full((b, k, 2), np.iinfo(np.uint64).max, np.uint64)
    osc = np.full((b, k), np.inf, np.float32)
    ocn = np.zeros(b, np.int32)
    for qi in range(b):
        rows = []
        for w in range(world):
            for j in range(int(gc[w, qi])):
                rows.append((float(gs[w, qi, j]), int(gd[w, qi, j, 1]), int(gd[w, qi, j, 0])))
        rows.sort()
        rows = rows[:k]
        ocn[qi] = len(rows)
        for j, (s, hi, lo) in enumerate(rows):
            od[qi, j] = (lo, hi)
            osc[qi, j] = s
    return torch.from_numpy(od.view(np.int64)), torch.from_numpy(osc), torch.from_numpy(ocn)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    rng = np.random.default_rng(3)
    v = H.sift_like(1500, 16, n_clusters=12, seed=4)
    c = H.kmeans(v, 10, iters=3, seed=1)
    doc_ids = [5 * i + 1 + ((i % 2) << 80) for i in range(1500)]
    q = (v[rng.integers(0, 1500, 12)] + rng.normal(0, 1, (12, 16))).astype(np.float32)
    k, P = 7, 4
    full_index, full_vec, pls = H.build_ivf_files(v, doc_ids, c)
    full = oracle.BlockBasedIvf(full_index, full_vec)
    probes = full.find_nearest_centroids(q, P)  # replicated centroids => identical probes on every rank
    # this rank's shard: only the posting lists it owns (single index: the library's size-balanced map)
    from muopdb_amd import formats as F
    owner = D.balanced_owners([len(pl) for pl in pls], world)
    mine = [pl if owner[l] == rank else np.zeros(0, np.uint64) for l, pl in enumerate(pls)]
    shard = oracle.BlockBasedIvf(F.write_ivf_index(c, doc_ids, mine), full_vec)
    r = shard.search(q, k, probes=probes)

    def local():
        docs = np.stack([r.lo, r.hi], -1).astype(np.uint64).view(np.int64)
        return torch.from_numpy(docs), torch.from_numpy(r.scores.copy()), torch.from_numpy(r.counts.astype(np.int32))

    docs, scores, counts = D.sharded_search(local, merge_numpy)
    ref = full.search(q, k, probes=probes)
    got = docs.numpy().view(np.uint64)
    ok = True
    for qi in range(len(q)):
        n = int(ref.counts[qi])
        ok &= int(counts[qi]) == n
        ok &= [(int(got[qi, j, 1]) << 64) | int(got[qi, j, 0]) for j in range(n)] == ref.doc_ids(qi)
        ok &= np.array_equal(scores[qi, :n].numpy(), ref.scores[qi, :n])
    # the step's real exchange: results written INTO this rank's packed block, ONE all-gather, merge of the received blocks
    pg = D.PackedTopkGather(None, len(q), k, "cpu")
    d0, s0, c0 = local()
    pg.ids.copy_(d0); pg.scores.copy_(s0); pg.counts.copy_(c0)
    pg.gather()
    ok &= pg.recv.numel() == world * D.block_bytes(len(q), k) and D.block_bytes(len(q), k) % 16 == 0
    views = pg.recv_views()
    ok &= bool(torch.equal(views[rank][0], d0) and torch.equal(views[rank][1], s0) and torch.equal(views[rank][2], c0))
    pd, ps, pc = merge_numpy(torch.stack([v_[0] for v_ in views]), torch.stack([v_[1] for v_ in views]), torch.stack([v_[2] for v_ in views]))
    ok &= bool(torch.equal(pd, docs) and torch.equal(ps, scores) and torch.equal(pc, counts))
    # sharded coarse search: every rank ranks its centroid range, rows are gathered [b][world][P] and merged by key
    cent = rng.standard_normal((200, 16)).astype(np.float32) * 8
    cent[150] = cent[3]                                      # a tie across two ranks' ranges: the lower index wins
    qs = (cent[rng.integers(0, 200, 9)] + rng.normal(0, 1, (9, 16))).astype(np.float32)
    qs[0] = cent[3]
    Pc = 6

    def key_rows(first, count):
        rows = np.full((len(qs), Pc), -1, np.int64)          # all ones == UINT64_MAX padding
        for i, qq in enumerate(qs):
            ks = sorted(((int(np.float32(oracle.l2(qq, cent[c])).view(np.uint32)) | 0x80000000) << 32) | c for c in range(first, first + count))
            for j, kk in enumerate(ks[:Pc]):
                rows[i, j] = np.uint64(kk).astype(np.int64)
        return rows

    first, count = D.coarse_range(200, rank, world)
    gathered = D.gather_coarse_rows(torch.from_numpy(key_rows(first, count))).numpy().view(np.uint64)   # [b][world][P]
    ok &= gathered.shape == (len(qs), world, Pc)
    merged = np.sort(gathered.reshape(len(qs), -1), axis=1)[:, :Pc]
    want = key_rows(0, 200).view(np.uint64)
    ok &= np.array_equal(merged, want)
    ok &= int(want[0, 0] & np.uint64(0xFFFFFFFF)) == 3 and int(want[0, 1] & np.uint64(0xFFFFFFFF)) == 150
    lo, hi = D.split_batch(10, rank, world)
    ok &= (lo, hi) == (rank * 5, rank * 5 + 5)
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ivf_gather_merge_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1.0


def merge_points_numpy(views, doc_table, k):
    """Host restatement of mdb_ivf_merge_shards: the k smallest of the union by (distance, point id), THEN doc ids and the
    IdWithScore order (search_with_centroids :250-286 across ranks, then search_with_centroids_and_remap :298-332 once)."""
    b = views[0][1].shape[0]
    out = []
    for qi in range(b):
        rows = []
        for w, (pids, scores, counts, _found) in enumerate(views):
            for j in range(int(counts[qi])):
                rows.append((float(scores[qi, j]), int(pids[qi, j]) & 0xFFFFFFFF, w))
        rows.sort()
        rows = rows[:k]
        out.append(sorted((s, doc_table[p]) for s, p, _ in rows))
    return out


def _exact_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from muopdb_amd import formats as F
    rng = np.random.default_rng(11)
    n, d, Lc, P = 1200, 16, 10, 5
    v = rng.integers(0, 3, (n, d)).astype(np.float32)                  # low entropy + 2-bit PQ: a handful of distinct distances
    c = H.kmeans(v + rng.normal(0, 0.01, v.shape).astype(np.float32), Lc, iters=3, seed=2)
    cb = H.train_pq_codebook(v[:600], 4, 2, iters=3)
    opq = oracle.ProductQuantizer(d, 4, 2, cb)
    oq = oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 4, 2, cb)
    doc_ids = [int(x) for x in rng.permutation(n) + 7000]              # NOT monotone in point ids (a reindexed segment)
    q = (v[rng.integers(0, n, 16)] + rng.normal(0, 0.3, (16, d))).astype(np.float32)
    index, vec, pls = H.build_ivf_files(v, doc_ids, c, quantize=opq.quantize)
    full = oracle.BlockBasedIvf(index, vec, oq)
    probes = full.find_nearest_centroids(q, P)
    owner = D.balanced_owners([len(pl) for pl in pls], world)
    mine = [pl if owner[l] == rank else np.zeros(0, np.uint64) for l, pl in enumerate(pls)]
    # this rank's search_with_centroids rows: the oracle over its lists with doc id == point id, so (score, "doc") IS (distance, point id)
    shard = oracle.BlockBasedIvf(F.write_ivf_index(c, list(range(n)), mine, quantized_dimension=4), vec, oq)
    ok, stale = True, 0
    for k in (1, 6):
        r = shard.search(q, k, probes=probes)
        g = D.PointsGather(None, len(q), k, "cpu")
        pids, scores, counts, found = D.points_block_views(g.send, len(q), k)
        pids.copy_(torch.from_numpy(r.lo[:, :k].astype(np.uint32).view(np.int32)))
        scores.copy_(torch.from_numpy(r.scores[:, :k].copy()))
        counts.copy_(torch.from_numpy(r.counts.astype(np.int32)))
        found.fill_(1)
        g.gather()                                                     # ONE all-gather of the points blocks
        ok &= g.recv.numel() == world * D.points_block_bytes(len(q), k) and D.points_block_bytes(len(q), k) % 16 == 0
        views = [tuple(t.numpy() for t in vw) for vw in g.recv_views()]
        got = merge_points_numpy(views, doc_ids, k)
        ref = full.search(q, k, probes=probes)
        for qi in range(len(q)):
            nq = int(ref.counts[qi])
            ok &= [dd for _, dd in got[qi]] == ref.doc_ids(qi)
            ok &= [np.float32(s_) for s_, _ in got[qi]] == [np.float32(x) for x in ref.scores[qi, :nq]]
            # the old rule — per-rank remap, then (score, doc id) re-selection — picks other documents at rank-k ties
            rows = []
            for pv, sv, cv, _ in views:
                rows += [(float(sv[qi, j]), doc_ids[int(pv[qi, j])]) for j in range(int(cv[qi]))]
            rows.sort()
            stale += [dd for _, dd in rows[:k]] != ref.doc_ids(qi)
    ok &= stale > 0
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_exact_sharded_merge_points_blocks_world2():
    """permuted doc ids + duplicate PQ codes over two gloo ranks: all-gather of POINTS blocks + merge by (distance, point id),
    then remap == the unsharded oracle row for row (and the old (score, doc id) merge is shown to differ on this case)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1.0


def test_shard_assignment_covers_every_list_once():
    for world in (1, 2, 3, 8):
        for L in (1, 40, 64, 65, 300, 4096, 65536, 65600):   # coarse ranges: whole tiles, contiguous, covering
            rs = [D.coarse_range(L, r, world) for r in range(world)]
            assert sum(c for _, c in rs) == L and all(f % 64 == 0 and f <= L and c <= L - f for f, c in rs)
            pos = 0
            for f, c in rs:
                if c:
                    assert f == pos
                    pos += c
        sizes = [((7 * l) % 23) * 10 + (l % 3) for l in range(100)]          # skewed list lengths
        bal = D.balanced_owners(sizes, world)
        loads = [sum(sz for sz, o in zip(sizes, bal) if o == r) for r in range(world)]
        assert set(bal) == set(range(world)) and max(loads) - min(loads) <= max(sizes)   # greedy longest-first bound
        assert D.balanced_owners(sizes, world) == bal                                     # deterministic: every rank derives the same map
        owners = [D.shard_of_list(l, world) for l in range(100)]
        assert set(owners) == set(range(min(world, 100)))
        assert all(0 <= o < world for o in owners)
        spans = [D.split_batch(64, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _partition_worker(rank, world, port, out):
    """users and batch partitionings (RowsExchange): this rank answers ITS queries whole with the oracle (a subset of the user table /
    a replica of the index), one all-gather of finished rows, permutation into batch order == the unsharded rows"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from muopdb_amd import formats as F
    rng = np.random.default_rng(21)
    d, k = 16, 4
    ok = True
    # ---- by user: 5 users (odd: the ranks get 3 and 2), a batch with repeated users and uneven routing (6 pairs / 5 pairs)
    files = {}
    for u in range(5):
        v = H.sift_like(300 + 40 * u, d, n_clusters=6, seed=30 + u)
        files[u + 1], _, _ = H.build_spann_files(oracle, v, list(range(10_000 * u, 10_000 * u + len(v))), 6, max_neighbors=6, max_layers=2,
                                                 ef_construction=30)
    cat = F.concat_multi_spann(files)
    a = (d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    table = np.frombuffer(bytes(cat["user_table"]), np.uint8).reshape(5, -1)
    full = oracle.MultiSpannIndex(bytes(cat["user_table"]), *a)
    mine = oracle.MultiSpannIndex(table[D.users_of_rank(5, rank, world)].tobytes(), *a)
    slots = [0, 3, 1, 1, 4, 2, 0, 4, 4, 3, 2]                       # the batch's pairs by user slot
    b = len(slots)
    q = H.sift_like(b, d, n_clusters=6, seed=77).astype(np.float32)
    op = oracle.SearchParams(k, 30, num_explored_centroids=3)
    ref = full.search_for_user([s_ + 1 for s_ in slots], q, op)
    routes = D.route_by_user(slots, world)
    ok &= sorted(i for r in routes for i in r) == list(range(b)) and all(D.user_owner(slots[i], world) == r for r, pos in enumerate(routes) for i in pos)
    ex = D.RowsExchange(b, k, routes, rank, "cpu")
    ql = ex.local_queries(torch.from_numpy(q)).numpy()
    loc = mine.search_for_user([slots[i] + 1 for i in routes[rank]], ql, op)
    n_loc = len(routes[rank])

    def fill(ex_, res, n_):
        ex_.send.zero_()
        ids = np.zeros((ex_.bmax, k, 2), np.uint64)
        ids[:n_, :, 0], ids[:n_, :, 1] = res.lo[:, :k], res.hi[:, :k]
        ex_.ids.copy_(torch.from_numpy(ids.view(np.int64)))
        sc = np.zeros((ex_.bmax, k), np.float32); sc[:n_] = res.scores[:, :k]
        ex_.scores.copy_(torch.from_numpy(sc))
        cn = np.zeros(ex_.bmax, np.int32); cn[:n_] = res.counts
        ex_.counts.copy_(torch.from_numpy(cn))
        fo = np.zeros(ex_.bmax, np.uint8); fo[:n_] = res.found if hasattr(res, "found") else 1
        ex_.found.copy_(torch.from_numpy(fo))

    def same(got, want):
        gi, gs, gc, gf = (t.numpy() for t in got)
        good = True
        for i in range(len(gc)):
            c_ = int(want.counts[i])
            good &= int(gc[i]) == c_ and int(gf[i]) == (int(want.found[i]) if hasattr(want, "found") else 1)
            good &= [int(x) for x in gi[i, :c_, 0].view(np.uint64)] == [int(x) for x in want.lo[i, :c_]]
            good &= gs[i, :c_].tobytes() == np.asarray(want.scores[i, :c_], np.float32).tobytes()
        return good
    fill(ex, loc, n_loc)
    ok &= same(ex.gather(), ref)
    # ---- by batch: replicas of one IVF index, 11 queries over 2 ranks (5 + 6)
    v = H.sift_like(1500, d, n_clusters=12, seed=4)
    c = H.kmeans(v, 10, iters=3, seed=1)
    index, vec, _ = H.build_ivf_files(v, [3 * i + 1 for i in range(1500)], c)
    ivf = oracle.BlockBasedIvf(index, vec)
    ref2 = ivf.search(q, k, num_probes=4)
    ex2 = D.RowsExchange(b, k, D.route_by_batch(b, world), rank, "cpu")
    lo, hi = D.split_batch(b, rank, world)
    ok &= ex2.contiguous and ex2.n_local == hi - lo
    ql2 = ex2.local_queries(torch.from_numpy(q)).numpy()
    ok &= ql2.tobytes() == q[lo:hi].tobytes()
    fill(ex2, ivf.search(ql2, k, num_probes=4), hi - lo)
    ok &= same(ex2.gather(), ref2)
    try:
        D.RowsExchange(b, k, [[0, 1], [1, 2]], rank, "cpu")          # a query served twice / not at all is refused
        ok = False
    except ValueError:
        pass
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_user_and_batch_partitionings_equal_unsharded_world2():
    """SURVEY 8e 'measure both': next to the list shards (tests above) the QUERY partitionings of bench.py --shard users / batch —
    user slot u on rank u % world with pairs routed to their owner, and replicas answering contiguous batch slices — return the
    unsharded rows after one all-gather of finished rows (uneven routes, a batch that does not divide by the world size)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1.0


def _probe_rows_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for b, words in ((7, 5), (1, 3), (8, 18), (3, 2)):
        # the table every rank must end up with: row i = what the centroid stage of pair i yields (here: a function of i)
        table = (np.arange(b * words, dtype=np.int64).reshape(b, words) * 2654435761 % (1 << 31)).astype(np.int32)
        sh = D.ProbeRowsShare(None, b, words, "cpu")
        lo, hi = sh.slice
        ok &= 0 <= lo <= hi <= b and hi - lo <= sh.per and sh.send.shape == (max(sh.per, 1), words)
        covered = torch.zeros(b, dtype=torch.int32)
        covered[lo:hi] = 1
        dist.all_reduce(covered)
        ok &= bool((covered == 1).all())                       # every pair's closure runs on exactly one rank
        sh.send[:hi - lo] = torch.from_numpy(table[lo:hi])
        rows = sh.gather()
        ok &= rows.shape == (world * max(sh.per, 1), words) and bool(torch.equal(rows[:b], torch.from_numpy(table)))
        ok &= bool((rows[b:] == 0).all())                       # padding rows of short slices: count 0, found 0
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_probe_rows_share_slices_cover_the_batch(world):
    """ProbeRowsShare (list-sharded multi-user SPANN, closure once per pair): the ranks' slices partition the batch, and ONE
    all-gather of the slices IS the batch's probe-row table in batch order on every rank, for even, ragged, short (b < world) batches."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_rows_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1.0
