"""world_size-2 `gloo` tests (CPU) of the multi-GPU host path: list sharding,
the per-batch all-gather of fixed-size top-k blocks and the (score, doc id) merge.  The per-shard
results come from the CPU oracle here (the GPU kernels are covered by the -m gpu suite, including
`test_merge_shards_device` and the shard-union tests); what is under test is the collective plumbing
and that the union of per-rank top-k equals the unsharded answer."""
import os
import socket

import numpy as np
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
