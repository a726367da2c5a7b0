"""GPU tests (-m gpu) of the build side: native Lloyd k-means (mdb_kmeans_fit) against the CPU
restatement of KMeansBuilder::run_lloyd (rs/utils/src/kmeans_builder/kmeans_builder.rs:163-360) — BIT parity from the same
initial points, incl. the size penalty, the lane-conforming distance variants and the empty-cluster repair — the
reference's own k-means tests through the GPU, PQ codebook training (quality parity: the reference trains with a third-party
crate) and IvfBuilder::build_centroids' list splitting."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


def _same(got, want):
    gc, ga, ge, gi = got
    wc, wa, we, wi = want
    assert gi == wi, "iterations %d != %d" % (gi, wi)
    assert np.array_equal(np.asarray(ga, np.int64), np.asarray(wa, np.int64))
    assert np.array_equal(np.asarray(gc, np.float32).view(np.uint32), np.asarray(wc, np.float32).view(np.uint32))
    assert np.float32(ge).view(np.uint32) == np.float32(we).view(np.uint32)


@pytest.mark.parametrize("n,d,k,tol,iters,seed", [
    (3000, 16, 20, 0.0, 10, 1),      # LaneConforming<16>
    (3000, 128, 37, 1e-4, 6, 2),     # C3 coarse shape, size penalty
    (2500, 24, 16, 1e-3, 8, 3),      # LaneConforming<8>
    (2000, 12, 9, 0.0, 8, 4),        # LaneConforming<4>
    (1500, 10, 7, 1e-4, 8, 5),       # full cascade (d % 4 != 0)
    (1200, 3, 5, 0.0, 20, 6),
    (4000, 8, 256, 0.0, 5, 7),       # PQ subvector shape
    (700, 768, 11, 0.0, 3, 8),       # C4 dimension
    (65, 5, 64, 0.0, 4, 9),          # nearly one point per cluster
    (300, 32, 1, 0.0, 5, 10),        # a single cluster: the labels repeat at once
])
def test_kmeans_bit_parity(ctx, oracle, n, d, k, tol, iters, seed):
    from muopdb_amd import build as B
    rng = np.random.default_rng(seed)
    x = H.sift_like(n, d, n_clusters=max(2, k // 2), seed=seed) if d >= 8 else rng.standard_normal((n, d)).astype(np.float32) * 10
    init = rng.choice(n, size=min(k, n), replace=False)
    _same(B.kmeans_fit(ctx, x, k, max_iter=iters, tolerance=tol, init_ids=init), oracle.kmeans_fit(x, k, iters, tol, init))


def test_kmeans_empty_cluster_repair_and_device_buffers(ctx, oracle):
    """duplicated rows and repeated init points force empty clusters (kmeans_builder.rs:268-314): same moves as the oracle;
    the same run with device-resident data (torch tensors) returns the same bits"""
    import torch
    from muopdb_amd import build as B
    rng = np.random.default_rng(12)
    base = rng.standard_normal((40, 16)).astype(np.float32) * 5
    x = np.concatenate([base, base, base[:10], rng.standard_normal((300, 16)).astype(np.float32)])
    init = np.array([0, 40, 80, 1, 41, 3, 3, 3, 100, 200, 201, 5], np.uint64)   # identical points -> empty clusters
    want = oracle.kmeans_fit(x, 12, 10, 0.0, init)
    _same(B.kmeans_fit(ctx, x, 12, max_iter=10, tolerance=0.0, init_ids=init), want)
    assert len(set(np.asarray(want[1]).tolist())) == 12
    xt = torch.from_numpy(x).cuda()
    c, a, e, it = B.kmeans_fit(ctx, xt, 12, max_iter=10, tolerance=0.0, init_ids=init)
    _same((c.cpu().numpy(), a.cpu().numpy(), e, it), want)
    want = oracle.kmeans_fit(x, 12, 10, 1e-3, init)                            # the penalty changes the assignment
    _same(B.kmeans_fit(ctx, x, 12, max_iter=10, tolerance=1e-3, init_ids=init), want)


def test_k14_reference_kmeans_tests_on_the_gpu(ctx):
    # rs/utils/src/kmeans_builder/kmeans_builder.rs:373-486 (see tests/test_oracle_kat.py::test_k14_*)
    from muopdb_amd import build as B
    from muopdb_amd import lib as L
    km = np.array([[0, 0], [40, 40], [90, 90], [1, 1], [41, 41], [91, 91], [2, 2], [42, 42], [92, 92]], np.float32)
    cent, a, err, it = B.kmeans_fit(ctx, km, 3, max_iter=100, tolerance=1e-4, init_ids=[0, 1, 2])
    assert a[0] == a[3] == a[6] and a[1] == a[4] == a[7] and a[2] == a[5] == a[8]
    assert cent.tolist() == [[1.0, 1.0], [41.0, 41.0], [91.0, 91.0]]
    km[7] = [5, 5]
    cent, a, err, it = B.kmeans_fit(ctx, km, 3, max_iter=100, tolerance=0.0, init_ids=[0, 1, 2])
    assert a[0] == a[3] == a[6] == a[7] and a[1] == a[4] and a[2] == a[5] == a[8]
    cent, a, err, it = B.kmeans_fit(ctx, km, 10, max_iter=100, tolerance=0.0, seed=3)     # 10 clusters of 9 points -> 9
    assert cent.shape == (9, 2) and set(a.tolist()) == set(range(9))
    with pytest.raises(L.MuopdbError):
        B.kmeans_fit(ctx, km, 3, init_ids=[0, 1])        # cluster_init_values of the wrong length
    with pytest.raises(L.MuopdbError):
        B.kmeans_fit(ctx, km, 3, init_ids=[0, 1, 99])    # not a point


def test_pq_training_quality_and_ivf_list_splitting(ctx, oracle):
    from muopdb_amd import build as B
    from muopdb_amd.index import ProductQuantizer
    x = H.sift_like(20000, 32, n_clusters=40, seed=21)
    cb = B.train_pq_codebook(ctx, x, 8, 6, iters=8, seed=1, sample=8000)
    ref = H.train_pq_codebook(x[:8000], 8, 6, iters=8)               # plain numpy Lloyd from another init
    pq, rq = ProductQuantizer(32, 8, 6, cb), ProductQuantizer(32, 8, 6, ref)

    def distortion(q):
        rec = q.original_vector(ctx, q.quantize(ctx, x))
        return float(((rec - x) ** 2).sum(1).mean())

    assert distortion(pq) <= 1.05 * distortion(rq)
    # IvfBuilder::build_centroids: no list longer than max_posting_list_size, every point in exactly one list
    cent, pls = B.ivf_build_centroids(ctx, x, num_clusters=16, max_posting_list_size=900, num_data_points_for_clustering=4000,
                                      max_iteration=6, tolerance=1e-5, seed=2)
    assert len(pls) == cent.shape[0] >= 23 and max(len(p) for p in pls) <= 900 and min(len(p) for p in pls) > 0
    allp = np.concatenate(pls)
    assert allp.size == len(x) and np.array_equal(np.sort(allp), np.arange(len(x), dtype=np.uint64))
    lab = B.assign_nearest(ctx, x, cent)
    sizes = np.bincount(lab, minlength=cent.shape[0])
    assert sizes.max() <= 4 * 900                                    # a final re-assignment stays balanced
    ids, cnt = B.assign_nearest(ctx, x[:500], cent, max_clusters_per_vector=2, distance_threshold=0.1)
    oids, ocnt = oracle.ivf_assign(cent, x[:500], 2, 0.1)
    assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt)


@pytest.mark.parametrize("pq", [False, True])
def test_open_segment_directory(ctx, oracle, tmp_path, pq):
    """a segment written as the reference lays it out on disk (odht `user_index_info`,
    centroids/..., ivf/..., quantizer YAML + codebook) and opened through MultiSpannReader::read's mirror
    (MultiSpannIndex.open_segment -> mdb_odht_user_table + mdb_multi_spann_load) gives the rows of the K9 data set."""
    from muopdb_amd import formats as F
    from muopdb_amd.index import MultiSpannIndex, ProductQuantizer, SearchParams
    v = np.concatenate([np.repeat(np.arange(1000, dtype=np.float32)[:, None], 4, 1), np.array([[1.2, 2.2, 3.2, 4.2]], np.float32)])
    quant = oquant = quantize = None
    pqcfg = None
    if pq:
        cb = H.train_pq_codebook(v, 2, 4)
        opq = oracle.ProductQuantizer(4, 2, 4, cb)
        quant, oquant, quantize, pqcfg = ProductQuantizer(4, 2, 4, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 2, 4, cb), opq.quantize, (4, 2, 4)
    f0, _, _ = H.build_spann_files(oracle, v, list(range(1001)), 10, quantize=quantize)
    f1, _, _ = H.build_spann_files(oracle, v[:50] + 0.5, list(range(5000, 5050)), 3, quantize=quantize)
    big = (1 << 70) + 1
    if pq:
        f0["codebook"] = f1["codebook"] = np.asarray(cb, np.float32).tobytes()
    cat = F.concat_multi_spann({0: f0, big: f1})
    seg = str(tmp_path / "segment")
    F.write_segment(seg, cat, 4, pq=pqcfg)
    g = MultiSpannIndex.open_segment(ctx, seg)
    assert g.num_users() == 2
    direct = MultiSpannIndex(ctx, cat["user_table"], 4, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], quant)
    o = oracle.MultiSpannIndex(cat["user_table"], 4, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    users = [0, big, 12345, 0]
    q = np.array([[1.4, 2.4, 3.4, 4.4], [3.1, 3.1, 3.1, 3.1], [0, 0, 0, 0], [500.2, 500.2, 500.2, 500.2]], np.float32)
    p, op = SearchParams(3, 2), oracle.SearchParams(3, 2)
    r, rd, ro = g.search_for_user(users, q, p), direct.search_for_user(users, q, p), o.search_for_user(users, q, op)
    assert r.found.tolist() == rd.found.tolist() == ro.found.tolist() == [1, 1, 0, 1]
    for i in range(4):
        assert r.doc_ids(i) == rd.doc_ids(i) == ro.doc_ids(i)
        assert np.array_equal(r.scores[i, :int(r.counts[i])].view(np.uint32), ro.scores[i, :int(ro.counts[i])].view(np.uint32))
    if not pq:
        assert r.doc_ids(0) == [1000, 3, 2]        # K9 (multi_spann/index.rs:358-412) through the on-disk tree


def test_k16_open_segment_replays_the_tombstone_log(ctx, oracle, tmp_path):
    """multi_spann/index.rs:525-599 (test_multi_spann_create_with_invalidation) on the GPU path: the K9 collection opened over
    a log holding (user 0, doc 1000) -> is_invalidated, num_entries 1, rows 3, 2, 4; then :602-668 (test_multi_spann_invalidate)
    and :671-760 (.._invalidate_batch): only EFFECTIVE invalidations reach the log, and a reopen sees them."""
    import os
    from muopdb_amd import formats as F
    from muopdb_amd.index import MultiSpannIndex, SearchParams
    v = np.concatenate([np.repeat(np.arange(1000, dtype=np.float32)[:, None], 4, 1), np.array([[1.2, 2.2, 3.2, 4.2]], np.float32)])
    f0, _, _ = H.build_spann_files(oracle, v, list(range(1001)), 10)
    cat = F.concat_multi_spann({0: f0})
    seg = str(tmp_path / "seg")
    os.makedirs(os.path.join(seg, "invalidated_ids_storage"))
    F.InvalidatedIdsStorage(os.path.join(seg, "invalidated_ids_storage"), 1024).invalidate(0, 1000)
    F.write_segment(seg, cat, 4)
    g = MultiSpannIndex.open_segment(ctx, seg)
    assert g.replayed_invalidations == 1 and g.is_invalidated(0, 1000) and not g.is_invalidated(0, 3)
    assert g.invalidated_ids_storage.num_entries() == 1
    r = g.search_for_user([0], [[1.4, 2.4, 3.4, 4.4]], SearchParams(3, 2))
    assert r.doc_ids(0) == [3, 2, 4]
    with pytest.raises(Exception, match="User not found"):
        g.is_invalidated(77, 1)
    # test_multi_spann_invalidate: effective -> logged, ineffective (absent doc / dead doc) -> not
    assert g.invalidate(0, 0) is True and g.invalidated_ids_storage.num_entries() == 2
    assert g.invalidate(0, 5000) is False and g.invalidate(0, 1000) is False and g.invalidated_ids_storage.num_entries() == 2
    # test_multi_spann_invalidate_batch: {0: [valid 1, valid 2, invalid]} -> 2; then [dead 2, valid 3, invalid] -> 1
    assert g.invalidate_batch({0: [1, 2, 5000]}) == 2 and g.invalidated_ids_storage.num_entries() == 4
    assert g.invalidate_batch({0: [2, 3, 5001]}) == 1
    assert sorted(g.invalidated_ids_storage) == [(0, 0), (0, 1), (0, 2), (0, 3), (0, 1000)]
    g.close()
    g2 = MultiSpannIndex.open_segment(ctx, seg)
    assert g2.replayed_invalidations == 5
    o = oracle.MultiSpannIndex(cat["user_table"], 4, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    o.apply_pending_invalidations(os.path.join(seg, "invalidated_ids_storage"))
    q = np.array([[1.4, 2.4, 3.4, 4.4], [0, 0, 0, 0]], np.float32)
    r, ro = g2.search_for_user([0, 0], q, SearchParams(3, 2)), o.search_for_user([0, 0], q, oracle.SearchParams(3, 2))
    for i in range(2):
        assert r.doc_ids(i) == ro.doc_ids(i)
    assert r.doc_ids(0) == [4, 5, 6]                                  # 0..3 and 1000 are dead


@pytest.mark.parametrize("pq", [False, True])
def test_reopened_segment_with_a_two_file_log_equals_oracle(ctx, oracle, tmp_path, pq):
    """A segment that had deletes: its log spans several files (64-byte files = 2 records), names a user the table does not
    hold, a document its user does not hold, and one pair twice.  Reopened (MultiSpannIndex::new multi_spann/index.rs:51-77 +
    get_or_create_index :121-124 -> mdb_multi_spann_replay_invalidations) it gives the rows of the oracle that applied the same
    log — and differs from the segment without the log; by-user shards skip the other users' records; the same through
    list shards of 2."""
    import os
    from muopdb_amd import formats as F
    from muopdb_amd.index import MultiSpannIndex, ProductQuantizer, SearchParams
    rng = np.random.default_rng(21)
    users, files, vecs = [5, (1 << 90) + 3, 77], {}, {}
    quant = oquant = quantize = pqcfg = None
    allv = H.sift_like(2400, 16, n_clusters=12, seed=9)
    if pq:
        cb = H.train_pq_codebook(allv[:1200], 4, 5, iters=3)
        opq = oracle.ProductQuantizer(16, 4, 5, cb)
        quant, oquant, quantize, pqcfg = ProductQuantizer(16, 4, 5, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 4, 5, cb), opq.quantize, (16, 4, 5)
    for ui, u in enumerate(users):
        vecs[u] = allv[ui * 800:(ui + 1) * 800]
        files[u], _, _ = H.build_spann_files(oracle, vecs[u], [1000 * ui + 3 * j for j in range(800)], 12, quantize=quantize, seed=ui,
                                             max_neighbors=8, max_layers=3, ef_construction=40)
        if pq:
            files[u]["codebook"] = np.asarray(cb, np.float32).tobytes()
    cat = F.concat_multi_spann(files)
    q = np.stack([vecs[u][j] + rng.normal(0, 2, 16) for u in users for j in (0, 100, 799)]).astype(np.float32)
    qu = [u for u in users for _ in range(3)]
    # the deletes: the nearest documents of every query's user (so rows MUST change), + noise records
    p, op = SearchParams(6, 40).with_num_explored_centroids(4), oracle.SearchParams(6, 40, num_explored_centroids=4)
    clean = oracle.MultiSpannIndex(cat["user_table"], 16, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    base_rows = clean.search_for_user(qu, q, op)
    dead = []
    for i, u in enumerate(qu):
        dead += [(u, d) for d in base_rows.doc_ids(i)[:2]]
    dead.insert(3, (424242, 1))                                       # a user absent from the table
    dead.insert(5, (5, 999999))                                       # a document its user does not hold
    dead.append(dead[0])                                              # one pair twice
    seg = str(tmp_path / "segment")
    F.write_segment(seg, cat, 16, pq=pqcfg, invalidated=dead, backing_file_size=64)
    names = sorted(os.listdir(os.path.join(seg, "invalidated_ids_storage")))
    assert len(names) == (len(dead) + 1) // 2 >= 10 and "invalidated_ids.bin.10" in names   # numeric suffix order matters
    o = oracle.MultiSpannIndex(cat["user_table"], 16, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    pending = o.apply_pending_invalidations(os.path.join(seg, "invalidated_ids_storage"))
    assert 424242 in pending
    want = o.search_for_user(qu, q, op)
    g = MultiSpannIndex.open_segment(ctx, seg)
    n_real = len({(u, d) for u, d in dead if u in users and d != 999999})
    assert g.replayed_invalidations == n_real
    got = g.search_for_user(qu, q, p)
    changed = 0
    for i in range(len(qu)):
        assert got.doc_ids(i) == want.doc_ids(i), i
        assert np.array_equal(got.scores[i, :int(got.counts[i])].view(np.uint32), want.scores[i, :int(want.counts[i])].view(np.uint32))
        changed += got.doc_ids(i) != base_rows.doc_ids(i)
        assert not set(got.doc_ids(i)) & {d for u, d in dead if u == qu[i]}
    assert changed == len(qu)
    # by-user shard: only user slot 1 (ids sorted: 5, 77, 2^90+3) resident -> the log's other records are skipped
    one = MultiSpannIndex.open_segment(ctx, seg, user_slots=[1])
    assert one.num_users() == 1 and one.replayed_invalidations == len({d for u, d in dead if u == 77})
    r1 = one.search_for_user(qu, q, p)
    for i in range(len(qu)):
        assert bool(r1.found[i]) == (qu[i] == 77)
        if qu[i] == 77:
            assert r1.doc_ids(i) == want.doc_ids(i)
    # list shards of 2: every rank replays the log over its replicated doc-id tables; merged rows == unsharded rows
    shards = [MultiSpannIndex.open_segment(ctx, seg, shard_rank=r, shard_world=2) for r in range(2)]
    blocks = [s_.search_shard(qu, q, p) for s_ in shards]
    merged = shards[0].merge_shards(qu, blocks, len(qu), 6)
    for i in range(len(qu)):
        assert merged.doc_ids(i) == want.doc_ids(i), i


def test_select_neighbors_heuristic(ctx, oracle):
    """mdb_hnsw_select_neighbors == select_neighbors_heuristic (hnsw/builder.rs:339-375) restated with the oracle's exact
    distance: pop order (distance asc, larger id first), keep e unless a kept x has distance(x, e) < distance(e, q)."""
    from muopdb_amd import build as B
    rng = np.random.default_rng(8)
    n, d, M, W = 600, 24, 8, 40
    x = H.sift_like(n, d, n_clusters=6, seed=5)
    x[100] = x[7]; x[200] = x[7]                                      # duplicates: exact ties
    cand = np.full((50, W), 0xFFFFFFFF, np.uint32)
    dist = np.full((50, W), np.inf, np.float32)
    for r in range(50):
        c = rng.choice(n, size=int(rng.integers(1, W + 1)), replace=False)
        if r % 5 == 0:
            c = np.unique(np.concatenate([c[:W - 3], [7, 100, 200]]))[:W]
        dd = np.array([oracle.l2(x[r], x[j]) for j in c], np.float32)
        order = np.lexsort((-c.astype(np.int64), dd))
        cand[r, :len(c)], dist[r, :len(c)] = c[order], dd[order]
    ids, ds, cnt = B.select_neighbors(ctx, x, cand, dist, M)
    for r in range(50):
        kept = []
        for j in range(W):
            e = int(cand[r, j])
            if e == 0xFFFFFFFF or len(kept) == M:
                break
            if all(not (np.float32(oracle.l2(x[e], x[k_])) < dist[r, j]) for k_, _ in kept):
                kept.append((e, dist[r, j]))
        assert int(cnt[r]) == len(kept) and ids[r, :len(kept)].tolist() == [e for e, _ in kept]
        assert np.array_equal(ds[r, :len(kept)].view(np.uint32), np.array([v for _, v in kept], np.float32).view(np.uint32))


def test_insert_hnsw_quality_parity_with_the_reference_algorithm(ctx, oracle):
    """the batched GPU construction (muopdb_amd.build.insert_hnsw: HnswBuilder::insert's steps over the
    library's traversal + selection kernels) against the oracle's sequential restatement of HnswBuilder::insert on the same
    points and parameters: recall@10 of the two graphs under the same search, degree bound, every point reachable."""
    from muopdb_amd import build as B
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedHnsw, FlatIndex
    n, d, M, efc = 9000, 32, 16, 64
    x = H.sift_like(n, d, n_clusters=60, seed=33)
    rng = np.random.default_rng(2)
    q = (x[rng.integers(0, n, 200)] + rng.normal(0, 4, (200, d))).astype(np.float32)
    layers, levels = B.insert_hnsw(ctx, x, max_neighbors=M, max_layers=4, ef_construction=efc, seed=3)
    _, indptr, edges = layers[0]
    deg = np.diff(indptr.astype(np.int64))
    assert deg.max() <= M and deg.min() >= 1 and len(layers) == int(levels.max()) + 1
    assert edges.max() < n and len(indptr) == n + 1
    idx = F.write_hnsw_index(layers, np.arange(n, dtype=np.uint64), d)
    vec = F.write_vector_file(x)
    g = BlockBasedHnsw(ctx, idx, vec, d)
    o = oracle.BlockBasedHnsw(idx, vec, d)                                # the oracle's traversal reads the GPU-built file too
    res = g.ann_search(q, 10, 50)
    ores = o.ann_search(q[:40], 10, 50)
    assert [res.doc_ids(i) for i in range(40)] == [ores.doc_ids(i) for i in range(40)]
    exact, _, _ = FlatIndex(ctx, x).search(q, 10)
    rec_gpu = np.mean([len(set(res.doc_ids(i)) & set(int(v) for v in exact[i])) / 10 for i in range(200)])
    ridx, rvec = H.build_hnsw_files(oracle, x, list(range(n)), max_neighbors=M, max_layers=4, ef_construction=efc, seed=3)
    rres = BlockBasedHnsw(ctx, ridx, rvec, d).ann_search(q, 10, 50)
    rec_ref = np.mean([len(set(rres.doc_ids(i)) & set(int(v) for v in exact[i])) / 10 for i in range(200)])
    assert rec_gpu >= 0.9 and rec_gpu >= rec_ref - 0.03, (rec_gpu, rec_ref)
    # nearly every point is reachable on layer 0 from the entry point (a build that dropped reverse edges would strand many)
    seen = np.zeros(n, bool)
    ep = int(layers[-1][0][0]) if len(layers) > 1 else int(np.argmax(deg > 0))
    stack, seen[ep] = [ep], True
    while stack:
        p_ = stack.pop()
        for e in edges[int(indptr[p_]):int(indptr[p_ + 1])].tolist():
            if not seen[e]:
                seen[e] = True
                stack.append(e)
    assert seen.mean() > 0.99      # (trimming can strand a few points in the sequential reference build as well)
