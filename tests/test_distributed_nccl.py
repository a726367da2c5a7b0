"""Multi-GPU path on the REAL backend: `torch.distributed` "nccl" = RCCL over xGMI, one process per GPU.
Self-skips when fewer than 2 GPUs are visible (every gpurun box has one): the driver's 8-GPU node runs it the moment
one exists.  The same plumbing runs on every round under world_size-2 gloo on CPU (tests/test_distributed_gloo.py) and
with simulated ranks on one GPU (test_gpu_parity / test_gpu_traversal / test_gpu_fullsize).

What each rank checks, on its own GPU, against the UNSHARDED index it also loads:
  * sharded coarse search (mdb_ivf_coarse_keys -> all-gather -> mdb_ivf_merge_coarse_keys) == find_nearest_centroids;
  * list-sharded IVF-PQ search (PERMUTED doc ids, coarse PQ: score ties at rank k) written straight into this rank's POINTS
    block -> ONE all-gather -> mdb_ivf_merge_shards (merge by (distance, point id), then remap) == the unsharded rows
    (ids and score bits);
  * the torch-free form of the same step: mdb_allgather_blocks(ctx, ncclComm_t, ...) with a communicator created through
    librccl's C API (what a Rust host would do) + mdb_ivf_merge_shards == the torch.distributed result;
    mdb_allgather_merge (the IdWithScore merge of remapped rows, flat shards / segments) on the same communicator;
  * multi-user SPANN sharded l % world the same way (mdb_multi_spann_search_shard / _merge_shards)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    ok, why = True, []

    def check(cond, msg):
        nonlocal ok
        if not cond:
            ok = False
            why.append(msg)

    try:
        import oracle
        from muopdb_amd import distributed as D
        from muopdb_amd import formats as F
        from muopdb_amd import lib as L
        from muopdb_amd.index import BlockBasedIvf, MultiSpannIndex, ProductQuantizer, SearchParams
        from tests import helpers as H
        ctx = L.Context(rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dev = torch.device("cuda", rank)
        rng = np.random.default_rng(5)                      # same data on every rank
        n, d, nl, k, P, b = 6000, 32, 200, 10, 12, 48
        v = H.sift_like(n, d, n_clusters=40, seed=3)
        cent = H.kmeans(v, nl, iters=3, seed=1)
        cb = H.train_pq_codebook(v[:2000], 8, 6, iters=3)
        opq = oracle.ProductQuantizer(d, 8, 6, cb)
        doc_ids = [int(x) for x in np.random.default_rng(9).permutation(n) * 3 + 1]     # not monotone in point ids
        index, vec, pls = H.build_ivf_files(v, doc_ids, cent, quantize=opq.quantize)
        pq = ProductQuantizer(d, 8, 6, cb)
        full = BlockBasedIvf(ctx, index, vec, pq)
        shard = BlockBasedIvf(ctx, index, vec, pq, shard_rank=rank, shard_world=world)
        owner = D.balanced_owners([len(p_) for p_ in pls], world)
        check(shard.num_resident_vectors() == sum(len(p_) for p_, o in zip(pls, owner) if o == rank), "balanced ownership")
        qh = (v[rng.integers(0, n, b)] + rng.normal(0, 2, (b, d))).astype(np.float32)
        q = torch.from_numpy(qh).to(dev)
        want = full.search(qh, k, P)
        # ---- sharded coarse search + packed all-gather + merge through torch.distributed (RCCL)
        probes = D.sharded_probes(ctx, shard, q.data_ptr(), b, P, dev)
        ctx.sync()
        check(np.array_equal(probes.cpu().numpy().astype(np.uint32), full.find_nearest_centroids(qh, P)), "sharded probes")
        g = D.PointsGather(ctx, b, k, dev)
        ctx.check(ctx.lib.mdb_ivf_search_shard(shard.h, C.c_void_p(q.data_ptr()), C.c_size_t(b), C.c_void_p(probes.data_ptr()), C.c_size_t(P),
                                               C.c_size_t(k), C.c_int(L.MEM_DEVICE), None, C.c_size_t(0), C.c_size_t(0),
                                               C.c_void_p(g.send.data_ptr())))
        docs, scores, counts = g.gather_merge_ivf(shard)
        ctx.sync()
        hd, hs, hc = docs.cpu().numpy().view(np.uint64), scores.cpu().numpy(), counts.cpu().numpy()
        for i in range(b):
            c = int(want.counts[i])
            check(int(hc[i]) == c, "count of row %d" % i)
            check([(int(hd[i, j, 1]) << 64) | int(hd[i, j, 0]) for j in range(c)] == want.doc_ids(i), "ids of row %d" % i)
            check(np.array_equal(hs[i, :c].view(np.uint32), np.asarray(want.scores[i, :c], np.float32).view(np.uint32)), "scores of row %d" % i)
        # ---- the same exchange without torch: raw RCCL communicator + mdb_allgather_merge
        rccl = None
        for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
            try:
                rccl = C.CDLL(name)
                break
            except OSError:
                pass
        check(rccl is not None, "librccl not loadable")
        if rccl is not None:
            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]
            rccl.ncclGetLastError.restype = C.c_char_p
            rccl.ncclGetLastError.argtypes = [C.c_void_p]
            uid = UniqueId()
            if rank == 0:
                rc = rccl.ncclGetUniqueId(C.byref(uid))
                check(rc == 0, "ncclGetUniqueId rc=%d %r" % (rc, rccl.ncclGetLastError(None)))
            box = [C.string_at(C.byref(uid), 128)] if rank == 0 else [None]     # (bytes(uid.internal) would stop at the first NUL)
            dist.broadcast_object_list(box, src=0)
            C.memmove(C.byref(uid), box[0], 128)
            comm = C.c_void_p()
            rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            rccl.ncclCommInitRank.restype = C.c_int
            rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
            check(rc == 0, "ncclCommInitRank rc=%d %r" % (rc, rccl.ncclGetLastError(None)))
            recv = torch.zeros(world * D.points_block_bytes(b, k), dtype=torch.uint8, device=dev)
            o_docs, o_sc, o_cn = torch.zeros_like(docs), torch.zeros_like(scores), torch.zeros_like(counts)
            torch.cuda.synchronize()
            ctx.check(ctx.lib.mdb_allgather_blocks(ctx.h, comm, C.c_void_p(g.send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                                   C.c_size_t(D.points_block_bytes(b, k))))
            ctx.check(ctx.lib.mdb_ivf_merge_shards(shard.h, C.c_void_p(recv.data_ptr()), C.c_size_t(world), C.c_size_t(b), C.c_size_t(k),
                                                   C.c_void_p(o_docs.data_ptr()), C.c_void_p(o_sc.data_ptr()), C.c_void_p(o_cn.data_ptr())))
            ctx.sync()
            check(bool(torch.equal(o_docs, docs) and torch.equal(o_sc, scores) and torch.equal(o_cn, counts)), "mdb_allgather_blocks + merge != torch path")
            # the doc-id form (rows of different indexes): every rank contributes the SAME full rows, the merge keeps them
            pg = D.PackedTopkGather(ctx, b, k, dev)
            pg.ids.copy_(docs); pg.scores.copy_(scores); pg.counts.copy_(counts)
            recv2 = torch.zeros(world * D.block_bytes(b, k), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.check(ctx.lib.mdb_allgather_merge(ctx.h, comm, C.c_void_p(pg.send.data_ptr()), C.c_void_p(recv2.data_ptr()), C.c_size_t(world),
                                                  C.c_size_t(b), C.c_size_t(k), C.c_void_p(o_docs.data_ptr()), C.c_void_p(o_sc.data_ptr()),
                                                  C.c_void_p(o_cn.data_ptr())))
            ctx.sync()
            if world == 1:
                check(bool(torch.equal(o_docs, docs) and torch.equal(o_sc, scores)), "mdb_allgather_merge of one rank")
            else:   # duplicates of every row: the first k of the doubled, sorted rows
                check(bool(torch.equal(o_sc[:, 0], scores[:, 0]) and torch.equal(o_docs[:, 0], docs[:, 0])), "mdb_allgather_merge head row")
            rccl.ncclCommDestroy(comm)
        # ---- multi-user SPANN, posting lists l % world
        users = {}
        for u in range(6):
            vu = H.sift_like(800, 16, n_clusters=8, seed=30 + u)
            users[u + 1], _, _ = H.build_spann_files(oracle, vu, list(range(1000 * u, 1000 * u + 800)), 12, max_neighbors=8, max_layers=2,
                                                     ef_construction=40)
        cat = F.concat_multi_spann(users)
        margs = (cat["user_table"], 16, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
        mfull = MultiSpannIndex(ctx, *margs)
        mshard = MultiSpannIndex(ctx, *margs, None, rank, world)
        uq = [1 + (i % 6) for i in range(24)]
        mq = np.stack([H.sift_like(24, 16, n_clusters=8, seed=30 + (i % 6))[i] for i in range(24)]).astype(np.float32)
        sp = SearchParams(5, 40).with_num_explored_centroids(6).with_centroid_distance_ratio(0.5)
        mwant = mfull.search_for_user(uq, mq, sp)
        g2 = D.PointsGather(ctx, 24, 5, dev)
        mqd = torch.from_numpy(mq).to(dev)
        pc = sp.to_c()
        uarr = L.u128_array(uq)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.mdb_multi_spann_search_shard(mshard.h, uarr, C.c_void_p(mqd.data_ptr()), C.c_size_t(24), C.byref(pc),
                                                       C.c_int(L.MEM_DEVICE), None, C.c_size_t(0), C.c_size_t(0),
                                                       C.c_void_p(g2.send.data_ptr())))
        d2, s2, c2 = g2.gather_merge_multi(mshard, uarr)
        ctx.sync()
        check(bool(g2.out_found.cpu().numpy().tolist() == mwant.found.tolist()), "multi-user found flags")
        h2 = d2.cpu().numpy().view(np.uint64)
        for i in range(24):
            c = int(mwant.counts[i])
            check(int(c2[i]) == c and [(int(h2[i, j, 1]) << 64) | int(h2[i, j, 0]) for j in range(c)] == mwant.doc_ids(i), "multi-user row %d" % i)
        # ---- the same batch with the centroid-graph closure run ONCE per pair: this rank's slice -> probe rows -> one more all-gather
        sh = D.ProbeRowsShare(ctx, 24, int(ctx.lib.mdb_spann_probe_row_words(C.byref(pc))), dev)
        lo, hi = sh.slice
        if hi > lo:
            ctx.check(ctx.lib.mdb_multi_spann_probes(mshard.h, L.u128_array(uq[lo:hi]), C.c_void_p(mqd[lo:hi].data_ptr()), C.c_size_t(hi - lo),
                                                     C.byref(pc), C.c_int(L.MEM_DEVICE), C.c_void_p(sh.send.data_ptr())))
        rows = sh.gather()
        ctx.sync()
        check(bool(np.array_equal(rows[:24].cpu().numpy().view(np.uint32), mfull.probes(uq, mq, sp))), "gathered probe rows == one rank's rows of the whole batch")
        blk_ref = g2.send.clone()
        g2.send.zero_()
        ctx.check(ctx.lib.mdb_multi_spann_search_shard_probes(mshard.h, uarr, C.c_void_p(mqd.data_ptr()), C.c_size_t(24), C.byref(pc),
                                                              C.c_int(L.MEM_DEVICE), C.c_void_p(rows.data_ptr()), None, C.c_size_t(0), C.c_size_t(0),
                                                              C.c_void_p(g2.send.data_ptr())))
        ctx.sync()
        check(bool(torch.equal(g2.send, blk_ref)), "points block from gathered probe rows == search_shard's")
        d3, s3, c3 = g2.gather_merge_multi(mshard, uarr)
        ctx.sync()
        h3 = d3.cpu().numpy().view(np.uint64)
        for i in range(24):
            c = int(mwant.counts[i])
            check(int(c3[i]) == c and [(int(h3[i, j, 1]) << 64) | int(h3[i, j, 0]) for j in range(c)] == mwant.doc_ids(i), "shared-closure row %d" % i)
    except Exception as e:  # noqa: BLE001
        ok = False
        why.append(repr(e))
    t = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    out.put((rank, ok, why[:6]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_over_rccl_world2():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (one process per GPU over RCCL); single-GPU boxes run the gloo / simulated-rank tests")
    import torch.multiprocessing as mp
    # 2 ranks, and — on the driver's 8-GPU node — every visible device (the coarse ranges and the size-balanced owners of 3..8 ranks)
    for world in sorted({2, min(8, torch.cuda.device_count())}):
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        results = [out.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert all(ok for _, ok, _ in results), (world, results)


def test_rccl_path_single_rank_dry_run():
    """The same worker with world_size 1 on ONE GPU: the nccl process group, a real ncclComm_t from librccl's C API and
    mdb_allgather_blocks' / mdb_allgather_merge's ncclAllGather all run (a one-rank all-gather is a copy) — everything but the second GPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), out))
    p.start()
    rank, ok, why = out.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0 and ok, why
