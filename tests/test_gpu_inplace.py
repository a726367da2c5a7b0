"""GPU tests (-m gpu) of two step-level paths around the scan kernels: device-resident query rows read IN PLACE (no staging
launch: stage_queries, mdb_flat.hip) and the merge of many sorted partial lists by bound + rank (merge_lists_fast).  Both must
return exactly the rows of the paths they replace (MDB_NO_INPLACE / MDB_FLAT_MERGE_OLD) and of the oracle."""
import ctypes as C

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


def _flat_device(ctx, g, qd, b, k):
    import torch
    dev = qd.device
    ids = torch.full((b, k), -1, dtype=torch.int32, device=dev)
    dist = torch.zeros((b, k), dtype=torch.float32, device=dev)
    cnt = torch.zeros(b, dtype=torch.int32, device=dev)
    g.search_device(qd.data_ptr(), b, k, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
    ctx.sync()
    return ids.cpu().numpy().view(np.uint32), dist.cpu().numpy(), cnt.cpu().numpy().view(np.uint32)


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("k", [1, 10, 64, 65])
def test_flat_one_query_in_place_rows_and_list_merge(ctx, oracle, k):
    """140 k x 128: 547 scan blocks -> 547 sorted partial lists per query (the bound + rank merge serves >= 512 lists of k <= 64)."""
    import torch
    from muopdb_amd.index import FlatIndex
    n, d = 140_000, 128
    x = H.sift_like(n, d, n_clusters=64, seed=3)
    rng = np.random.default_rng(4)
    q = (x[rng.integers(0, n, 8)] + rng.normal(0, 3, (8, d))).astype(np.float32)
    g = FlatIndex(ctx, x)
    dev = torch.device("cuda", torch.cuda.current_device())
    qd = torch.from_numpy(q).to(dev)
    oids, odist = oracle.flat_topk(oracle.METRIC_L2, x, q[:2], k)
    for b in (1, 4, 8, 3):   # 3: a batch with a padding row is staged
        with ctx.option("MDB_FLAT_NO_MFMA", 1):
            got = _flat_device(ctx, g, qd, b, k)
            with ctx.option("MDB_NO_INPLACE", 1):
                staged = _flat_device(ctx, g, qd, b, k)
            with ctx.option("MDB_FLAT_MERGE_OLD", 1):
                old = _flat_device(ctx, g, qd, b, k)
            host = g.search(q[:b], k)
        assert _same(got, staged) and _same(got, old) and _same(got, host)
        nb = min(b, 2)
        assert np.array_equal(got[0][:nb], oids[:nb]) and np.array_equal(got[1][:nb].view(np.uint32), odist[:nb].view(np.uint32))
        assert np.all(got[2] == k)
    # rows that do not start on a 16-byte boundary are staged as before
    pad = torch.zeros(8 * d + 1, dtype=torch.float32, device=dev)
    pad[1:] = qd.reshape(-1)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        odd = _flat_device(ctx, g, pad[1:], 4, k)
        ref = _flat_device(ctx, g, qd, 4, k)
    assert _same(odd, ref)
    # the batched path (sample bound + matrix-core filter + refine) with rows in place
    if k <= 64:
        a = _flat_device(ctx, g, qd, 8, k)
        with ctx.option("MDB_NO_INPLACE", 1):
            s = _flat_device(ctx, g, qd, 8, k)
        with ctx.option("MDB_FLAT_NO_MFMA", 1):
            e = _flat_device(ctx, g, qd, 8, k)
        assert _same(a, s) and _same(a, e)


@pytest.mark.parametrize("dups", [0, 300, 3000])
def test_flat_list_merge_under_ties(ctx, oracle, dups):
    """Exact ties at the k-th distance: `dups` copies of the query's nearest row spread over the base (300: they pass the bound and
    are ranked by id; 3000: more than the candidate capacity -> the streaming merge), and an all-equal base (every key ties)."""
    import torch
    from muopdb_amd.index import FlatIndex
    n, d, k = 140_000, 128, 10
    rng = np.random.default_rng(7)
    if dups:
        x = H.sift_like(n, d, n_clusters=64, seed=5)
        q = (x[:1] + 0.25).astype(np.float32)
        where = np.sort(rng.choice(n, dups, replace=False))
        x[where] = x[0]
    else:
        x = np.ones((n, d), np.float32)
        q = np.zeros((1, d), np.float32)
    g = FlatIndex(ctx, x)
    dev = torch.device("cuda", torch.cuda.current_device())
    qd = torch.from_numpy(q).to(dev)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        got = _flat_device(ctx, g, qd, 1, k)
        with ctx.option("MDB_FLAT_MERGE_OLD", 1):
            old = _flat_device(ctx, g, qd, 1, k)
    oids, odist = oracle.flat_topk(oracle.METRIC_L2, x, q, k)
    assert _same(got, old)
    assert np.array_equal(got[0], oids) and np.array_equal(got[1].view(np.uint32), odist.view(np.uint32))
    if dups:
        want = np.sort(np.unique(np.concatenate([where, [0]])))[:k]
        assert np.array_equal(got[0][0], want.astype(np.uint32))   # the smallest ids among the copies
    else:
        assert np.array_equal(got[0][0], np.arange(k, dtype=np.uint32))


def _rows128(ids_t, cnt_t, b):
    hi = ids_t.cpu().numpy().view(np.uint64)
    return [[(int(hi[i, j, 1]) << 64) | int(hi[i, j, 0]) for j in range(int(cnt_t[i]))] for i in range(b)]


def test_ivf_and_spann_device_rows_in_place(ctx, oracle):
    """f32 posting lists (NoQuantizer) and SPANN with device-resident rows: in place == staged == the oracle's rows."""
    import torch
    from muopdb_amd import lib as L_
    from muopdb_amd.index import BlockBasedIvf, Spann, SearchParams
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(11)
    n, d, nl, P, k, b = 4000, 128, 40, 8, 10, 8
    v = H.sift_like(n, d, n_clusters=30, seed=12)
    cent = H.kmeans(v, nl, iters=3, seed=1)
    index, vec, _ = H.build_ivf_files(v, list(range(100, 100 + n)), cent)
    g = BlockBasedIvf(ctx, index, vec, None)
    o = oracle.BlockBasedIvf(index, vec, None)
    q = (v[rng.integers(0, n, b)] + rng.normal(0, 2, (b, d))).astype(np.float32)
    want = o.search(q, k, num_probes=P)
    qd = torch.from_numpy(q).to(dev)

    def ivf_dev():
        ids = torch.zeros((b, k, 2), dtype=torch.int64, device=dev)
        sc = torch.zeros((b, k), dtype=torch.float32, device=dev)
        cn = torch.zeros(b, dtype=torch.int32, device=dev)
        ctx.check(ctx.lib.mdb_ivf_search(g.h, C.c_void_p(qd.data_ptr()), C.c_size_t(b), None, C.c_size_t(P), C.c_size_t(k),
                                         C.c_int(L_.MEM_DEVICE), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(cn.data_ptr())))
        ctx.sync()
        return _rows128(ids, cn, b), sc.cpu().numpy()

    rows, sc = ivf_dev()
    with ctx.option("MDB_NO_INPLACE", 1):
        rows_s, sc_s = ivf_dev()
    assert rows == rows_s == [want.doc_ids(i) for i in range(b)]
    assert np.array_equal(sc.view(np.uint32), sc_s.view(np.uint32))
    for i in range(b):
        c = int(want.counts[i])
        assert np.array_equal(sc[i, :c].view(np.uint32), np.asarray(want.scores[i, :c], np.float32).view(np.uint32))

    # SPANN: centroid graph + ratio filter + posting lists, d = 128, batch 8
    files, _, _ = H.build_spann_files(oracle, v, list(range(10, 10 + n)), 40, max_neighbors=8, max_layers=3, ef_construction=50)
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    p = SearchParams(k, 100).with_num_explored_centroids(8).with_centroid_distance_ratio(0.3)
    owant = osp.search(q, oracle.SearchParams(k, 100, num_explored_centroids=8, centroid_distance_ratio=0.3))
    pc = p.to_c()

    def spann_dev():
        ids = torch.zeros((b, k, 2), dtype=torch.int64, device=dev)
        sc = torch.zeros((b, k), dtype=torch.float32, device=dev)
        cn = torch.zeros(b, dtype=torch.int32, device=dev)
        fo = torch.zeros(b, dtype=torch.uint8, device=dev)
        ctx.check(ctx.lib.mdb_spann_search(sp.h, C.c_void_p(qd.data_ptr()), C.c_size_t(b), C.byref(pc), C.c_int(L_.MEM_DEVICE),
                                           C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(cn.data_ptr()),
                                           C.c_void_p(fo.data_ptr())))
        ctx.sync()
        return _rows128(ids, cn, b), sc.cpu().numpy()

    rows, sc = spann_dev()
    with ctx.option("MDB_NO_INPLACE", 1):
        rows_s, sc_s = spann_dev()
    assert rows == rows_s == [owant.doc_ids(i) for i in range(b)]
    assert np.array_equal(sc.view(np.uint32), sc_s.view(np.uint32))


def test_flat_list_merge_beyond_one_list_per_thread(ctx, oracle):
    """300 k x 16: 1172 scan blocks for one query (one round each) -> more lists than the merge block has threads."""
    import torch
    from muopdb_amd.index import FlatIndex
    n, d, k = 300_000, 16, 10
    x = H.sift_like(n, d, n_clusters=40, seed=9)
    rng = np.random.default_rng(10)
    q = (x[rng.integers(0, n, 2)] + rng.normal(0, 2, (2, d))).astype(np.float32)
    g = FlatIndex(ctx, x)
    dev = torch.device("cuda", torch.cuda.current_device())
    oids, odist = oracle.flat_topk(oracle.METRIC_L2, x, q, k)
    for i in range(2):
        qd = torch.from_numpy(q[i:i + 1].copy()).to(dev)
        with ctx.option("MDB_FLAT_NO_MFMA", 1):
            got = _flat_device(ctx, g, qd, 1, k)
            with ctx.option("MDB_FLAT_MERGE_OLD", 1):
                old = _flat_device(ctx, g, qd, 1, k)
            with ctx.option("MDB_FLAT_BLOCKS", 1024):
                few = _flat_device(ctx, g, qd, 1, k)
        assert _same(got, old) and _same(got, few)
        assert np.array_equal(got[0][0], oids[i]) and np.array_equal(got[1][0].view(np.uint32), odist[i].view(np.uint32))


@pytest.mark.parametrize("nsplit,blk", [(1, 0), (2, 0), (3, 0), (8, 0), (16, 0), (5, 64), (4, 128)])
def test_f32_posting_scan_splits_and_block_sizes(ctx, oracle, nsplit, blk):
    """ivf_scan_f32_kernel under every blocks-per-query / threads-per-block choice: splits beyond a query's tiles return empty rows,
    the split index is rotated by the query index (XCD-aware) — the rows must not notice."""
    from muopdb_amd.index import BlockBasedIvf
    rng = np.random.default_rng(100 + nsplit)
    n, d, nl, P, k, b = 6000, 64, 50, 12, 10, 37
    v = H.sift_like(n, d, n_clusters=25, seed=21)
    cent = H.kmeans(v, nl, iters=3, seed=2)
    index, vec, _ = H.build_ivf_files(v, list(range(n)), cent)
    g = BlockBasedIvf(ctx, index, vec, None)
    o = oracle.BlockBasedIvf(index, vec, None)
    q = (v[rng.integers(0, n, b)] + rng.normal(0, 2, (b, d))).astype(np.float32)
    want = o.search(q, k, num_probes=P)
    with ctx.option("MDB_SCAN_F32_NSPLIT", nsplit), ctx.option("MDB_SCAN_F32_BLK", blk):
        got = g.search(q, k, P)
        one = g.search(q, k, 1)          # one probe: every split but the first is beyond the tiles
    for i in range(b):
        assert got.doc_ids(i) == want.doc_ids(i)
        c = int(want.counts[i])
        assert np.array_equal(np.asarray(got.scores[i, :c], np.float32).view(np.uint32), np.asarray(want.scores[i, :c], np.float32).view(np.uint32))
    want1 = o.search(q, k, num_probes=1)
    assert all(one.doc_ids(i) == want1.doc_ids(i) for i in range(b))
