"""BASELINE.json's FULL sizes on the GPU (C1 10k x 128 flat, C2 SIFT-1M HNSW ef=200, C3 SIFT-1M IVF-PQ), checked through
properties that do not depend on the size — sortedness, idempotence, batch-split invariance, self-retrieval,
tombstone monotonicity, recall against the exact scan — plus direct oracle parity on a handful of queries (the CPU
oracle needs ~1 ms per HNSW query and ~12 ms per 1M-row exact scan, so a few rows at full size are affordable).
The synthetic base is bench.py's (muopdb_amd.build.SiftLike — block low-rank + noise, clipped to [0, 218], rounded).
C4 runs at 1/8 of its users and at its FULL size (1024 users x 9766 x 768 = 30.7 GB resident); C5 runs as ONE GPU of the 8
sees it: rank 0's posting lists of the 100M-row index (12.5 M x 16-byte codes), the full 65 536-centroid coarse quantizer,
nprobe 64, batch 4096."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 128, 10


def rows_of(res, b):
    return [(res.doc_ids(i), np.asarray(res.scores[i, :int(res.counts[i])], np.float32).view(np.uint32).tolist()) for i in range(b)]


def assert_sorted(res, b):
    """IdWithScore order (rs/index/src/utils.rs:89-128): score ascending, then doc id."""
    for i in range(b):
        n = int(res.counts[i])
        sc = np.asarray(res.scores[i, :n], np.float32)
        ids = res.doc_ids(i)
        assert np.all(sc[:-1] <= sc[1:])
        for j in range(n - 1):
            if sc[j] == sc[j + 1]:
                assert ids[j] < ids[j + 1]


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def base(ctx):
    """(device rows, host rows, 96 host queries): the C2/C3 base of bench.py"""
    from muopdb_amd import build as B, synth as S
    gen = S.SiftLike(D, seed=1)
    x = gen.draw(N, seed=11)
    q = gen.draw(96, seed=4242)
    return x, x.cpu().numpy(), q.cpu().numpy().astype(np.float32)


@pytest.fixture(scope="module")
def flat_1m(ctx, base):
    from muopdb_amd.index import FlatIndex
    return FlatIndex(ctx, base[1])


def _threads(oracle):
    """the oracle's OpenMP threads for the full-size legs (one query per thread, the rows do not depend on it —
    64 queries cost what 4 cost on one thread)"""
    return max(1, min(64, oracle.num_threads()))


def test_c1_flat_10k_full_parity(ctx, oracle):
    """C1 is small enough for the oracle in full: 10 clusters x 1000 x 128 (py/create_test_hdf5.py semantics), 200 queries."""
    from muopdb_amd.index import FlatIndex
    x = H.test_hdf5_like()
    rng = np.random.default_rng(43)
    q = (x[rng.integers(0, len(x), 200)] + rng.normal(0, 5, (200, D))).astype(np.float32)
    g = FlatIndex(ctx, x)
    ids, dist, cnt = g.search(q, K)
    oids, odist = oracle.flat_topk(oracle.METRIC_L2, x, q, K)
    assert np.array_equal(ids, oids) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32)) and np.all(cnt == K)
    one = [g.search(q[i:i + 1], K) for i in range(0, 200, 37)]  # batch 1 (the configuration's batch) == rows of the batch
    for j, i in enumerate(range(0, 200, 37)):
        assert np.array_equal(one[j][0][0], ids[i]) and np.array_equal(one[j][1][0].view(np.uint32), dist[i].view(np.uint32))


def test_flat_1m_properties_and_oracle_rows(ctx, oracle, base, flat_1m):
    _, xh, q = base
    ids, dist, cnt = flat_1m.search(q[:64], K)             # batched path (sample bound + MFMA filter + exact refine)
    assert np.all(cnt == K)
    assert np.all(dist[:, :-1] <= dist[:, 1:])              # sorted by distance ...
    tie = dist[:, :-1] == dist[:, 1:]
    assert np.all(ids[:, :-1][tie] < ids[:, 1:][tie])       # ... then by row id
    again = flat_1m.search(q[:64], K)                       # idempotent
    assert np.array_equal(again[0], ids) and np.array_equal(again[1].view(np.uint32), dist.view(np.uint32))
    for lo, hi in [(0, 1), (1, 4), (4, 32), (32, 64)]:      # batch-split invariance: exact kernels (B <= 4) and batched path agree
        part = flat_1m.search(q[lo:hi], K)
        assert np.array_equal(part[0], ids[lo:hi]) and np.array_equal(part[1].view(np.uint32), dist[lo:hi].view(np.uint32))
    oids, odist = oracle.flat_topk(oracle.METRIC_L2, xh, q[:32], K, threads=_threads(oracle))   # the oracle itself on the full base
    assert np.array_equal(ids[:32], oids) and np.array_equal(dist[:32].view(np.uint32), odist.view(np.uint32))
    rows = np.array([5, 77_777, 500_000, 999_999])          # self-retrieval: a stored row finds itself at distance 0
    sids, sdist, _ = flat_1m.search(xh[rows], K)
    assert np.all(sdist[:, 0] == 0.0)
    for r, row in enumerate(rows):
        zero = sids[r][sdist[r] == 0.0]
        assert row in zero and np.all(np.diff(zero) > 0)    # duplicates of the clipped / rounded data: ascending ids


@pytest.fixture(scope="module")
def hnsw_1m(ctx, base):
    from muopdb_amd import build as B, synth as S
    from muopdb_amd.index import BlockBasedHnsw
    idx, vec = S.hnsw_files(base[0], max_neighbors=32, max_layers=8, kcand=64, seed=1)
    return idx, vec, BlockBasedHnsw(ctx, idx, vec, D)


def test_c2_hnsw_1m_ef200(ctx, oracle, base, flat_1m, hnsw_1m):
    _, xh, q = base
    idx, vec, g = hnsw_1m
    res = g.ann_search(q[:64], K, 200)
    assert all(int(c) == K for c in res.counts[:64])
    assert_sorted(res, 64)
    assert rows_of(g.ann_search(q[:64], K, 200), 64) == rows_of(res, 64)                 # idempotent
    whole = rows_of(res, 64)
    for lo, hi in [(0, 1), (1, 9), (9, 64)]:                                             # one block per query: any split, same rows
        assert rows_of(g.ann_search(q[lo:hi], K, 200), hi - lo) == whole[lo:hi]
    o = oracle.BlockBasedHnsw(idx, vec, D)                                               # the oracle on the same 768 MB of files
    ores = o.ann_search(q[:64], K, 200, threads=_threads(oracle))                         # all 64 queries of the batch (one per thread)
    evals, expanded = o.stats()
    g.ann_search(q[:64], K, 200)
    st = ctx.stats()
    assert rows_of(ores, 64) == whole
    assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded)             # same traversal, step for step
    # rows of any length; the all-in-one kernel; the upper layers on sorted positions for no launch / the layer-1 launch too (31 k upper
    # points: 16 registers per bitmap, the block's 156 us sort — mdb_hnsw_rank.hip.h; the default is the top launch only)
    for variant, value in (("MDB_HNSW_NO_ROW64", 1), ("MDB_HNSW_NO_TABLE", 1), ("MDB_HNSW_RANK", 0), ("MDB_HNSW_RANK", 3)):
        with ctx.option(variant, value):
            pres = g.ann_search(q[:64], K, 200)
            assert rows_of(pres, 64) == whole, (variant, value)
            st = ctx.stats()
            assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded), (variant, value)
            if variant == "MDB_HNSW_RANK":   # ... and below the split path's batch size: ONE upper launch over every layer >= 1
                assert rows_of(g.ann_search(q[:9], K, 200), 9) == whole[:9], (variant, value)
    exact_ids, _, _ = flat_1m.search(q[:64], K)                                          # recall@10 against the exact scan
    hit = sum(len(set(res.doc_ids(i)) & set(int(v) for v in exact_ids[i])) for i in range(64))
    assert hit / (64 * K) >= 0.99
    # ef is monotone for the result quality: a larger ef never loses exact neighbours on this base
    wide = g.ann_search(q[:64], K, 400)   # > 256: the eight-register beam of the table path
    st_w = ctx.stats()
    hit_w = sum(len(set(wide.doc_ids(i)) & set(int(v) for v in exact_ids[i])) for i in range(64))
    hit_n = sum(len(set(res.doc_ids(i)) & set(int(v) for v in exact_ids[i])) for i in range(64))
    assert hit_w >= hit_n
    o.stats()
    assert rows_of(o.ann_search(q[:64], K, 400, threads=_threads(oracle)), 64) == rows_of(wide, 64)   # the oracle's rows on 64 queries at ef = 400 ...
    assert (st_w["distance_evals"], st_w["expanded_nodes"]) == tuple(o.stats())          # ... and its traversal, step for step


def test_c3_ivfpq_1m_nprobe16(ctx, oracle, base):
    import torch
    from muopdb_amd import build as B, synth as S
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    x, xh, q = base
    nlist, P = 4096, 16
    cent = B.kmeans(ctx, x, nlist, iters=4, seed=3, sample=300_000)
    assign = B.assign_nearest(ctx, x, cent)
    cb = B.train_pq_codebook(ctx, x, 8, 8, iters=4, seed=4, sample=100_000)
    pq = ProductQuantizer(D, 8, 8, cb)
    codes = pq.quantize(ctx, xh)
    pls = B.posting_lists_from_assignment(assign, nlist)
    index = F.write_ivf_index(cent.cpu().numpy(), np.arange(N, dtype=np.uint64), pls, quantized_dimension=D // 8)
    vec = F.write_vector_file(codes)
    del assign
    torch.cuda.empty_cache()
    g = BlockBasedIvf(ctx, index, vec, pq)
    assert g.num_vectors() == N and g.num_clusters() == nlist
    res = g.search(q[:96], K, P)
    assert_sorted(res, 96)
    whole = rows_of(res, 96)
    assert rows_of(g.search(q[:96], K, P), 96) == whole                                   # idempotent
    for lo, hi in [(0, 1), (1, 40), (40, 96)]:                                            # batch-split invariance
        assert rows_of(g.search(q[lo:hi], K, P), hi - lo) == whole[lo:hi]
    # the configuration's own batch: 256 queries in ONE call (its launch geometry: one block per CU), row for row the rows of
    # the smaller calls, and the oracle's on a sample of them
    q256 = np.concatenate([q, S.SiftLike(D, seed=1).draw(160, seed=4343).cpu().numpy().astype(np.float32)])
    big = g.search(q256, K, P)
    assert_sorted(big, 256)
    rows256 = rows_of(big, 256)
    assert rows256[:96] == whole
    for lo, hi in [(96, 160), (160, 256)]:
        assert rows_of(g.search(q256[lo:hi], K, P), hi - lo) == rows256[lo:hi]
    o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 8, cb))
    assert np.array_equal(g.find_nearest_centroids(q[:96], P), o.find_nearest_centroids(q[:96], P))
    assert rows_of(o.search(q[:96], K, num_probes=P, threads=_threads(oracle)), 96) == whole[:96]   # the oracle on the full index, all 96 rows
    pick = [100, 129, 200, 255]
    assert rows_of(o.search(q256[pick], K, num_probes=P), 4) == [rows256[i] for i in pick]
    # more probes scan a superset of lists: the k-th symmetric-PQ score can only improve
    more = g.search(q[:32], K, 2 * P)
    for i in range(32):
        assert float(more.scores[i, K - 1]) <= float(res.scores[i, K - 1])
    # tombstones are monotone: invalidating a query's best document removes it and never changes the rest's order
    victims = sorted({res.doc_ids(i)[0] for i in range(8)})
    for doc in victims:
        assert g.invalidate(doc) and o.invalidate(doc)
    after = g.search(q[:8], K, P)
    assert rows_of(o.search(q[:8], K, num_probes=P), 8) == rows_of(after, 8)
    for i in range(8):
        kept = [dd for dd in res.doc_ids(i) if dd not in victims]
        assert after.doc_ids(i)[:len(kept)] == kept


def test_c4_shape_multi_user_spann_eighth(ctx, oracle):
    """C4's shape at 1/8 of its users (the full 1024 x 9766 x 768 = 30.7 GB is bench.py's --users 1024): 128 users x 9766 x
    768 unit-norm rows, one (user, query) pair per user, ef=200 >= the ~150 centroids of a user (closure kernel),
    num_explored_centroids 16, ratio 0.1.  Properties + the oracle on 24 users + union of two list shards == unsharded."""
    from muopdb_amd import build as B, synth as S
    from muopdb_amd import formats as F
    from muopdb_amd.index import MultiSpannIndex, SearchParams
    U, per, d, P = 128, 9766, 768, 16
    users, base = {}, []
    gen = S.EmbedLike(d, seed=3)
    for u in range(U):
        x = gen.draw(gen.user(u), per, seed=3_000_000 + u)
        cent = B.kmeans(ctx, x, per // 64, iters=3, seed=u)
        pls = B.posting_lists_from_assignment(B.assign_nearest(ctx, x, cent), cent.shape[0])
        hi, hv = S.hnsw_files(cent, max_neighbors=16, max_layers=4, kcand=32, seed=u)
        docs = np.arange(u * per, (u + 1) * per, dtype=np.uint64)
        users[u + 1] = dict(hnsw_index=hi, hnsw_vectors=hv, ivf_index=F.write_ivf_index(cent.cpu().numpy(), docs, pls),
                            ivf_vectors=F.write_vector_file(x.cpu().numpy()))
        base.append(x[:4].cpu().numpy())
    cat = F.concat_multi_spann(users)
    del users
    args = (cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    g = MultiSpannIndex(ctx, *args)
    rng = np.random.default_rng(5)
    uids = [u + 1 for u in range(U)]
    q = np.stack([base[u][1] + rng.normal(0, 0.3 / d ** 0.5, d) for u in range(U)]).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = SearchParams(K, 200).with_num_explored_centroids(P).with_centroid_distance_ratio(0.1)
    res = g.search_for_user(uids, q, p)
    assert all(res.found[i] for i in range(U))
    assert_sorted(res, U)
    whole = rows_of(res, U)
    assert rows_of(g.search_for_user(uids, q, p), U) == whole                                     # idempotent
    for lo, hi in [(0, 1), (1, 50), (50, U)]:                                                     # any batch composition
        assert rows_of(g.search_for_user(uids[lo:hi], q[lo:hi], p), hi - lo) == whole[lo:hi]
    for u in range(U):                                                                            # every hit belongs to the query's user
        assert all(u * per <= dd < (u + 1) * per for dd in res.doc_ids(u))
    o = oracle.MultiSpannIndex(*args)
    op = oracle.SearchParams(K, 200, num_explored_centroids=P, centroid_distance_ratio=0.1)
    assert rows_of(o.search_for_user(uids, q, op, threads=_threads(oracle)), U) == whole      # every pair of the batch
    shards = [MultiSpannIndex(ctx, *args, None, r, 2) for r in range(2)]                          # lists l % 2 == r
    merged = shards[0].merge_shards(uids, [sh.search_shard(uids, q, p) for sh in shards], U, K)   # the exact merge of the points blocks
    assert rows_of(merged, U) == whole


def test_c4_full_size_multi_user_spann(ctx, oracle):
    """BASELINE config C4 at its FULL size on one GPU: 1024 users x 9766 x 768 f32 = 10 M x 768 (30.7 GB resident), batch 1024
    = one (user, query) pair per user.  Size-independent properties, the oracle on 16 users of the full index, and the
    union of EIGHT posting-list shards (what the 8 GPUs hold) == the unsharded rows on 64 users."""
    import torch
    from muopdb_amd import build as B, synth as S
    from muopdb_amd import formats as F
    from muopdb_amd.index import MultiSpannIndex, SearchParams
    U, per, d, P = 1024, 9766, 768, 16
    gen = S.EmbedLike(d, seed=3)
    users, q = {}, []
    for u in range(U):
        uc = gen.user(u)
        x = gen.draw(uc, per, seed=3_000_000 + u)
        cent = B.kmeans(ctx, x, per // 64, iters=2, seed=u)
        pls = B.posting_lists_from_assignment(B.assign_nearest(ctx, x, cent), cent.shape[0])
        hi, hv = S.hnsw_files(cent, max_neighbors=16, max_layers=4, kcand=32, seed=u)
        docs = np.arange(u * per, (u + 1) * per, dtype=np.uint64)
        users[u + 1] = dict(hnsw_index=hi, hnsw_vectors=hv, ivf_index=F.write_ivf_index(cent.cpu().numpy(), docs, pls),
                            ivf_vectors=F.write_vector_file(x.cpu().numpy()))
        q.append(gen.draw(uc, 1, seed=7_000_000 + u).cpu().numpy()[0])
    del x
    torch.cuda.empty_cache()
    cat = F.concat_multi_spann(users)
    del users
    assert len(cat["ivf_vectors"]) > 30_000_000_000
    args = (cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    g = MultiSpannIndex(ctx, *args)
    assert g.num_users() == U
    uids = [u + 1 for u in range(U)]
    q = np.stack(q).astype(np.float32)
    p = SearchParams(K, 200).with_num_explored_centroids(P).with_centroid_distance_ratio(0.1)
    res = g.search_for_user(uids, q, p)                                                           # batch 1024
    assert all(res.found[i] for i in range(U))
    assert_sorted(res, U)
    whole = rows_of(res, U)
    assert rows_of(g.search_for_user(uids, q, p), U) == whole                                     # idempotent
    for lo, hi in [(0, 1), (1, 300), (300, U)]:                                                   # any batch composition
        assert rows_of(g.search_for_user(uids[lo:hi], q[lo:hi], p), hi - lo) == whole[lo:hi]
    for u in range(U):                                                                            # every hit belongs to the query's user
        assert all(u * per <= dd < (u + 1) * per for dd in res.doc_ids(u))
    rev = list(range(U - 1, -1, -1))                                                              # batch order is irrelevant
    rr = rows_of(g.search_for_user([uids[i] for i in rev], q[rev], p), U)
    assert [rr[U - 1 - i] for i in range(U)] == whole
    wide = g.search_for_user(uids[:64], q[:64], SearchParams(K, 200).with_num_explored_centroids(64).with_centroid_distance_ratio(0.3))
    for i in range(64):                                                                           # a superset of lists: k-th score can only improve
        assert float(wide.scores[i, K - 1]) <= float(res.scores[i, K - 1])
    sel = sorted(set([0, 1, 2, 3, 100, 101, 500, 511, 512, 777, 1000, 1020, 1021, 1022, 1023, 640] + list(range(5, 1024, 21))))   # 64 users
    o = oracle.MultiSpannIndex(*args)
    op = oracle.SearchParams(K, 200, num_explored_centroids=P, centroid_distance_ratio=0.1)
    assert rows_of(o.search_for_user([uids[i] for i in sel], q[sel], op, threads=_threads(oracle)), len(sel)) == [whole[i] for i in sel]
    del o
    g.close()
    sub = list(range(0, U, 16))                                                                   # 64 users through 8 list shards
    blocks, sh = [], None
    for r in range(8):
        if sh is not None:
            sh.close()
        sh = MultiSpannIndex(ctx, *args, None, r, 8)
        blocks.append(sh.search_shard([uids[i] for i in sub], q[sub], p))
    merged = sh.merge_shards([uids[i] for i in sub], blocks, len(sub), K)                         # any rank merges: doc-id tables are replicated
    sh.close()
    assert rows_of(merged, len(sub)) == [whole[i] for i in sub]


def test_c5_shard_ivfpq(ctx, oracle):
    """BASELINE config C5 as one GPU of the 8 runs it (ivf/block_based/index.rs:147-163, 250-286): rank 0's posting lists
    (l % 8 == 0, ~12.5 M x 16-byte PQ codes) of the 100M x 128 index, the FULL replicated coarse quantizer (65 536
    centroids), nprobe 64, batch 4096.  Properties (sorted, idempotent, batch-split invariant, tombstone monotone), the oracle's
    rows on 12 queries, and the coarse search sharded over a simulated world of 8 == the unsharded probe ids."""
    import torch
    from muopdb_amd import build as B, synth as S
    from muopdb_amd.distributed import coarse_range
    from muopdb_amd.index import BlockBasedIvf
    sh = S.c5_shard(ctx, total=100_000_000, world=8, rank=0, nlist=65536)
    torch.cuda.empty_cache()
    assert sh["nlist"] == 65536 and 11_000_000 < sh["n"] < 14_000_000 and sh["owned_lists"] > 8000
    g = BlockBasedIvf(ctx, sh["index"], sh["vectors"], sh["pq"])
    assert g.num_vectors() == sh["n"] and g.num_clusters() == 65536
    P, B_ = 64, 4096
    q = sh["gen"].draw(B_, seed=5000).cpu().numpy()
    res = g.search(q, K, P)                                                                # the configuration's batch
    assert_sorted(res, B_)
    whole = rows_of(res, B_)
    assert sum(len(r[0]) == K for r in whole) > B_ * 0.9                                   # ~8 owned probes x ~1500 codes per query
    assert rows_of(g.search(q, K, P), B_) == whole                                         # idempotent
    for lo, hi in [(0, 1), (1, 6), (6, 500), (500, 1525)]:                                 # batch-split invariance (exact / batched coarse paths)
        assert rows_of(g.search(q[lo:hi], K, P), hi - lo) == whole[lo:hi]
    probes = g.find_nearest_centroids(q[:256], P)
    assert np.array_equal(probes[:4], g.find_nearest_centroids(q[:4], P))                  # batched (MFMA-filtered) == exact coarse kernels
    rows = []
    for r in range(8):                                                                     # sharded coarse search, simulated world of 8
        first, count = coarse_range(65536, r, 8)
        rows.append(g.coarse_keys(q[:256], P, first, count))
    assert np.array_equal(g.merge_coarse_keys(np.stack(rows, axis=1), P), probes)
    assert rows_of(g.search_with_centroids_and_remap(q[:256], probes, K), 256) == whole[:256]
    o = oracle.BlockBasedIvf(sh["index"], sh["vectors"], oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 8, sh["codebook"]))
    assert np.array_equal(o.find_nearest_centroids(q[:12], P), probes[:12])
    assert rows_of(o.search(q[:64], K, num_probes=P, threads=_threads(oracle)), 64) == whole[:64]    # the oracle on the full shard
    more = g.search(q[:64], K, 2 * P)                                                      # more probes: the k-th score can only improve
    for i in range(64):
        if len(whole[i][0]) == K:
            assert float(more.scores[i, K - 1]) <= float(res.scores[i, K - 1])
    victims = sorted({res.doc_ids(i)[0] for i in range(12) if res.doc_ids(i)})             # tombstones are monotone
    for doc in victims:
        assert g.invalidate(doc) and o.invalidate(doc)
    after = g.search(q[:12], K, P)
    assert rows_of(o.search(q[:12], K, num_probes=P), 12) == rows_of(after, 12)
    for i in range(12):
        kept = [dd for dd in res.doc_ids(i) if dd not in victims]
        assert after.doc_ids(i)[:len(kept)] == kept


def test_c5_full_ivfpq_one_gpu_equals_merge_of_8_shards(ctx, oracle):
    """BASELINE config C5 WHOLE on one GPU: all 100 M x 16-byte PQ codes / 65 536 posting lists resident (1.6 GB
    of codes: 180 x under one MI355X's HBM), nprobe 64, batch 4096 — the N = 1 anchor of C5's strong scaling.  The unsharded rows
    must equal (a) the exact merge of the points blocks of 8 simulated list shards (l % 8 == r, loaded from the SAME files:
    ivf/block_based/index.rs:250-332 per shard, then the (distance, point id) merge + remap) on 256 queries, with the probes found
    by the coarse search sharded over the same world of 8; (b) the oracle's rows on 8 queries."""
    import torch
    from muopdb_amd import synth as S
    from muopdb_amd.distributed import coarse_range
    from muopdb_amd.index import BlockBasedIvf
    full = S.c5_index(ctx, total=100_000_000, world=1, rank=0, nlist=65536)
    torch.cuda.empty_cache()
    assert full["n"] == 100_000_000 and full["nlist"] == 65536 and full["owned_lists"] > 60000
    g = BlockBasedIvf(ctx, full["index"], full["vectors"], full["pq"])
    assert g.num_vectors() == 100_000_000 and g.num_clusters() == 65536
    P, B_ = 64, 4096
    q = full["gen"].draw(B_, seed=5000).cpu().numpy()
    res = g.search(q, K, P)                                                                # the configuration's batch
    assert_sorted(res, B_)
    whole = rows_of(res, B_)
    assert all(len(r[0]) == K for r in whole)                                              # 64 probes x ~1500 codes per query
    assert rows_of(g.search(q, K, P), B_) == whole                                         # idempotent
    for lo, hi in [(0, 1), (1, 6), (500, 1525)]:                                           # batch-split invariance
        assert rows_of(g.search(q[lo:hi], K, P), hi - lo) == whole[lo:hi]
    nq = 256
    probes = g.find_nearest_centroids(q[:nq], P)
    rows = []
    for r in range(8):                                                                     # the coarse search, sharded over 8
        first, count = coarse_range(65536, r, 8)
        rows.append(g.coarse_keys(q[:nq], P, first, count))
    assert np.array_equal(g.merge_coarse_keys(np.stack(rows, axis=1), P), probes)
    blocks, sh = [], None
    for r in range(8):                                                                     # 8 list shards of the same files
        if sh is not None:
            sh.close()
        sh = BlockBasedIvf(ctx, full["index"], full["vectors"], full["pq"], shard_rank=r, shard_world=8)
        blocks.append(sh.search_shard(q[:nq], K, probes=probes))
    merged = sh.merge_shards(blocks, nq, K)                                                # any rank merges: doc-id tables are replicated
    sh.close()
    assert rows_of(merged, nq) == whole[:nq]
    o = oracle.BlockBasedIvf(full["index"], full["vectors"], oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 8, full["codebook"]))
    assert np.array_equal(o.find_nearest_centroids(q[:8], P), probes[:8])
    assert rows_of(o.search(q[:64], K, num_probes=P, threads=_threads(oracle)), 64) == whole[:64]    # the oracle on the full index
    g.close()
