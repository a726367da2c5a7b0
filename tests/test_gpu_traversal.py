"""GPU parity tests (-m gpu) for the traversal rows: HNSW (H1/H2), SPANN (S1), multi-user SPANN
(M1) and the shard merge (§8e) — through the C ABI, against the CPU oracle and the reference's
end-to-end known answers (K8, K9, K10, K13)."""
import ctypes as C

import numpy as np
import pytest

from muopdb_amd import formats as F
from tests import helpers as H
from tests.test_gpu_parity import assert_result_rows, assert_scores

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


def _line_vectors(n=1000):
    return np.repeat(np.arange(n, dtype=np.float32)[:, None], 4, 1)


# ----------------------------------------------------------------------------------- HNSW
def test_hnsw_k13_hand_graph(ctx, oracle):
    # the 3-layer graph of rs/index/src/hnsw/writer.rs:269-430 with distinguishable vectors
    from muopdb_amd.index import BlockBasedHnsw
    l2 = {1: []}
    l1 = {1: [4, 5], 4: [1, 5], 5: [1, 4]}
    l0 = {1: [4, 5], 4: [1, 5], 5: [1, 4], 2: [1, 3], 3: [2, 4], 0: [1, 2]}
    vec = np.arange(6 * 16, dtype=np.float32).reshape(6, 16) % 7
    index = F.write_hnsw_index([l0, l1, l2], [1, 2, 3, 4, 5, 6], 16)
    vf = F.write_vector_file(vec)
    g = BlockBasedHnsw(ctx, index, vf, 16)
    o = oracle.BlockBasedHnsw(index, vf, 16)
    q = np.array([vec[3] + 0.25, vec[0] - 0.5, np.zeros(16)], np.float32)
    for k, ef in [(3, 1), (6, 10), (2, 2), (10, 3)]:
        assert_result_rows(g.ann_search(q, k, ef), o.ann_search(q, k, ef), len(q))


@pytest.mark.parametrize("n,d,M,layers,efc,metric,seed", [
    (1500, 8, 12, 4, 60, 0, 1), (2000, 128, 16, 3, 80, 0, 2), (1000, 4, 10, 2, 100, 0, 3), (800, 30, 8, 5, 40, 1, 4),
    (3000, 16, 6, 6, 30, 0, 5), (500, 17, 24, 1, 50, 0, 6), (300, 768, 8, 2, 40, 0, 7),
    (900, 16, 40, 3, 60, 0, 8)])   # the last: layer-0 rows of 80 edges — two 64-edge chunks per row (not the ROW64 kernel)
def test_hnsw_ann_search(ctx, oracle, n, d, M, layers, efc, metric, seed):
    from muopdb_amd.index import BlockBasedHnsw, NoQuantizer
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, d)).astype(np.float32)
    doc = [7 * i + 3 + ((i % 3) << 90) for i in range(n)]
    hidx, hvec = H.build_hnsw_files(oracle, v, doc, max_neighbors=M, max_layers=layers, ef_construction=efc, seed=seed,
                                    metric=metric)
    g = BlockBasedHnsw(ctx, hidx, hvec, d, NoQuantizer(d, metric))
    o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_NONE, metric))
    q = rng.standard_normal((33, d)).astype(np.float32)
    for k, ef in [(10, 200), (1, 1), (5, 7), (50, 64), (10, 0)]:
        ores = o.ann_search(q, k, ef)
        assert_result_rows(g.ann_search(q, k, ef), ores, len(q))
    evals, expanded = o.stats()
    o.ann_search(q, 10, 200)
    evals, expanded = o.stats()
    g.ann_search(q, 10, 200)
    st = ctx.stats()
    assert st["distance_evals"] == evals and st["expanded_nodes"] == expanded  # same traversal, step for step


@pytest.mark.parametrize("n,d,sub,bits,metric,ef", [(2000, 64, 8, 6, 0, 100), (1500, 32, 16, 5, 0, 300), (1200, 24, 6, 4, 0, 64),
                                                   (1000, 21, 7, 3, 1, 50), (1500, 32, 2, 4, 0, 40)])
def test_hnsw_over_pq_codes(ctx, oracle, n, d, sub, bits, metric, ef):
    """BlockBasedHnsw<ProductQuantizer>: the graph file stores quantized_dimension = m, the vector file the
    u8 codes; the query is quantized and every distance is ProductQuantizer::distance (symmetric, lane
    accumulators across subvectors, no sqrt).  Heavy ties (few distinct codes) included."""
    from muopdb_amd.index import BlockBasedHnsw, ProductQuantizer
    rng = np.random.default_rng(n + d)
    v = H.sift_like(n, d, n_clusters=25, seed=d)
    cb = H.train_pq_codebook(v[:1000], sub, bits, iters=3)
    opq = oracle.ProductQuantizer(d, sub, bits, cb, metric)
    codes = opq.quantize(v)
    m = d // sub
    b = oracle.HnswBuilder(d, 10, 4, 60, metric, 3)
    b.insert(v)
    layers, eps = b.layers(), b.entry_points()
    if len(layers) > 1:
        top = layers[-1]
        layers[-1] = {eps[0]: top[eps[0]], **{p: e for p, e in top.items() if p != eps[0]}}
    doc = [11 * i + 5 for i in range(n)]
    hidx, hvec = F.write_hnsw_index(layers, doc, m), F.write_vector_file(codes)
    g = BlockBasedHnsw(ctx, hidx, hvec, d, ProductQuantizer(d, sub, bits, cb, metric))
    o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_PQ, metric, sub, bits, cb))
    q = (v[rng.integers(0, n, 20)] + rng.normal(0, 5, (20, d))).astype(np.float32)
    for k, e in [(10, ef), (3, 5)]:
        assert_result_rows(g.ann_search(q, k, e), o.ann_search(q, k, e), len(q))
    o.stats()  # reset the oracle's counters
    o.ann_search(q, 10, ef)
    evals, expanded = o.stats()
    g.ann_search(q, 10, ef)
    st = ctx.stats()
    assert st["distance_evals"] == evals and st["expanded_nodes"] == expanded


@pytest.mark.parametrize("n,d,M,layers,metric,seed", [
    (1, 8, 4, 1, 0, 1), (2, 16, 4, 2, 0, 2), (17, 5, 4, 3, 0, 3), (152, 768, 16, 3, 0, 4), (300, 128, 8, 4, 1, 5),
    (600, 30, 3, 5, 0, 6), (2000, 12, 6, 4, 0, 7)])
def test_hnsw_small_graph_closure_kernel(ctx, oracle, n, d, M, layers, metric, seed):
    """Graphs with no more points than ef (SPANN centroid graphs) run hnsw_closure_kernel (whole frontiers per
    round); rows AND the evaluation / expansion counters must equal the sequential oracle's, and the
    sequential kernels' (MDB_HNSW_NO_CLOSURE)."""
    import os
    from muopdb_amd.index import BlockBasedHnsw, NoQuantizer
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, d)).astype(np.float32)
    if n > 20:
        v[n // 2] = v[3]  # exact duplicates: distance ties broken by id
        v[n // 3] = v[3]
    doc = [11 * i + 5 + ((i % 2) << 77) for i in range(n)]
    hidx, hvec = H.build_hnsw_files(oracle, v, doc, max_neighbors=M, max_layers=layers, ef_construction=30, seed=seed,
                                    metric=metric)
    g = BlockBasedHnsw(ctx, hidx, hvec, d, NoQuantizer(d, metric))
    o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_NONE, metric))
    q = rng.standard_normal((40, d)).astype(np.float32)
    q[0] = v[min(3, n - 1)]
    for k, ef in [(10, n), (n + 5, n + 7), (1, 4096), (16, max(n, 200))]:
        o.ann_search(q[:1], 1, 1)
        o.stats()
        ores = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        ctx.stats()
        gres = g.ann_search(q, k, ef)
        st = ctx.stats()
        assert_result_rows(gres, ores, len(q))
        assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded)
        with ctx.option("MDB_HNSW_NO_CLOSURE", 1):
            sres = g.ann_search(q, k, ef)
        st2 = ctx.stats()
        assert_result_rows(sres, ores, len(q))
        assert (st2["distance_evals"], st2["expanded_nodes"]) == (evals, expanded)
    if n > 1:   # one below n: the sequential kernels again
        assert_result_rows(g.ann_search(q, 5, n - 1), o.ann_search(q, 5, n - 1), len(q))


def test_hnsw_general_kernel_equals_beam_kernel(ctx, oracle):
    """hnsw_search_kernel (sorted LDS sets; serves ef > 256) must give the oracle's rows at small ef too:
    MDB_HNSW_NO_BEAM routes ef <= 256 through it."""
    import os
    from muopdb_amd.index import BlockBasedHnsw
    rng = np.random.default_rng(17)
    v = H.sift_like(2500, 48, n_clusters=20, seed=4)
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(2500)), max_neighbors=12, max_layers=4, ef_construction=60)
    g = BlockBasedHnsw(ctx, hidx, hvec, 48)
    o = oracle.BlockBasedHnsw(hidx, hvec, 48)
    q = (v[rng.integers(0, 2500, 25)] + rng.normal(0, 4, (25, 48))).astype(np.float32)
    with ctx.option("MDB_HNSW_NO_BEAM", 1):
        for k, ef in [(10, 100), (5, 8), (20, 256), (10, 600)]:
            assert_result_rows(g.ann_search(q, k, ef), o.ann_search(q, k, ef), len(q))
    assert_result_rows(g.ann_search(q, 10, 600), o.ann_search(q, 10, 600), len(q))


@pytest.mark.parametrize("variant", ["MDB_HNSW_NO_ROW64", "MDB_HNSW_GENERIC_DIST", "MDB_HNSW_NO_TABLE", "MDB_HNSW_NO_SPLIT",
                                     "MDB_HNSW_NO_WIDE", "MDB_HNSW_TABLE_NO_LDS", "MDB_HNSW_RANK=0", "MDB_HNSW_RANK=1", "MDB_HNSW_RANK=3"])
@pytest.mark.parametrize("d,metric", [(128, 0), (768, 1), (128, 1)])
def test_hnsw_beam_kernel_variants_equal_oracle(ctx, oracle, d, metric, variant):
    """hnsw_beam_kernel's variants — rows of any length (NO_ROW64), the generic distance cascade and the all-in-one kernel (NO_TABLE: the default route takes the
    upper layers through the distance table + hnsw_upper_kernel) — must give the oracle's rows AND counters: no speculative
    touch may be counted.  MDB_HNSW_RANK: the upper layers on sorted positions (mdb_hnsw_rank.hip.h) for neither launch, the
    layer-1 / single launch, both (default: the top launch of the split path only)."""
    variant, _, value = variant.partition("=")
    value = int(value or 1)
    from muopdb_amd.index import BlockBasedHnsw, NoQuantizer
    rng = np.random.default_rng(31)
    n = 3000
    v = H.sift_like(n, d, n_clusters=24, seed=6)
    if metric == 1:
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(n)), max_neighbors=12, max_layers=4, ef_construction=60, metric=metric)
    g = BlockBasedHnsw(ctx, hidx, hvec, d, NoQuantizer(d, metric))
    o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_NONE, metric))
    q = (v[rng.integers(0, n, 40)] + rng.normal(0, 0.05 if metric == 1 else 4, (40, d))).astype(np.float32)
    with ctx.option(variant, value):
        for k, ef in [(10, 100), (5, 8), (20, 256), (10, 1), (10, 40), (10, 300), (20, 448)]:   # (the last two: the 8-register beam)
            want = o.ann_search(q, k, ef)
            evals, expanded = o.stats()
            got = g.ann_search(q, k, ef)
            st = ctx.stats()
            assert_result_rows(got, want, len(q))
            assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded), (k, ef)


def test_hnsw_beam_overflow_falls_back_to_general_kernel(ctx, oracle):
    """Hundreds of exact distance ties with `furthest` overflow the 320-slot register beam; those queries are
    re-run by the general kernel inside the same call — rows and counters still equal the oracle's."""
    from muopdb_amd.index import BlockBasedHnsw
    rng = np.random.default_rng(23)
    v = np.repeat(rng.integers(0, 3, (6, 8)).astype(np.float32), 150, axis=0)  # 6 distinct points x 150 copies
    v = v[rng.permutation(len(v))]
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(len(v))), max_neighbors=16, max_layers=3, ef_construction=60)
    g = BlockBasedHnsw(ctx, hidx, hvec, 8)
    o = oracle.BlockBasedHnsw(hidx, hvec, 8)
    q = (v[:12] + 0.25).astype(np.float32)
    for k, ef in [(10, 100), (50, 200), (5, 256)]:
        o.stats()
        ores = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        assert_result_rows(g.ann_search(q, k, ef), ores, len(q))
        st = ctx.stats()
        assert st["distance_evals"] == evals and st["expanded_nodes"] == expanded


@pytest.mark.parametrize("layers,nq", [(2, 9), (3, 40), (4, 9)])
def test_hnsw_layer0_overflow_behind_a_clean_upper_traversal_counts_once(ctx, oracle, layers, nq):
    """scripts/stress_parity.py case 3958 (round 4): 2500 copies of 64 distinct points, dot product, ef 256.  The upper layers hold
    fewer points than ef (traversed completely, no overflow), the layer-0 beam then overflows on the ties and the block re-runs the
    WHOLE query with the general traversal: the upper launch's evaluations must not have been counted already (they travel with the
    hand-over state now and are added by the layer-0 block only when the query ends inside the beam).  Split path (>= 3 layers,
    batch >= 32) and single upper launch alike."""
    from muopdb_amd.index import BlockBasedHnsw, NoQuantizer
    rng = np.random.default_rng(3958 + layers)
    base = rng.integers(0, 4, (312, 3)).astype(np.float32)
    v = base[rng.integers(0, 312, 2500)]
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(len(v))), max_neighbors=32, max_layers=layers, ef_construction=40, metric=1)
    g = BlockBasedHnsw(ctx, hidx, hvec, 3, NoQuantizer(3, 1))
    o = oracle.BlockBasedHnsw(hidx, hvec, 3, oracle.Quant(oracle.QUANT_NONE, 1))
    q = (v[rng.integers(0, len(v), nq)] + rng.normal(0, 1, (nq, 3))).astype(np.float32)
    for k, ef in [(1, 256), (10, 200), (5, 400)]:
        o.stats()
        ores = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        assert_result_rows(g.ann_search(q, k, ef), ores, len(q))
        st = ctx.stats()
        assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded), (k, ef)


def test_hnsw_rows_with_duplicate_edges(ctx, oracle):
    """An adjacency row that names a point twice (nothing in the file format forbids it): the reference skips the second copy
    as already visited, and so must the kernels' test-and-set (two lanes of one row hitting the same bit: exactly one is new) —
    rows AND counters equal the oracle's."""
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedHnsw
    rng = np.random.default_rng(41)
    n, d, M = 1500, 32, 10
    v = H.sift_like(n, d, n_clusters=12, seed=9)
    b = oracle.HnswBuilder(d, M, 3, 40, 0, 1)
    b.insert(v)
    layers, eps = b.layers(), b.entry_points()
    if len(layers) > 1:   # the reader's entry point: FIRST point of the top layer
        top = layers[-1]
        ordered = {eps[0]: top[eps[0]]}
        ordered.update({p_: e for p_, e in top.items() if p_ != eps[0]})
        layers[-1] = ordered
    dup = 0
    for p_ in list(layers[0].keys()):
        e = list(layers[0][p_])
        if len(e) >= 3 and rng.random() < 0.3:      # repeat one or two of its neighbours, anywhere in the row
            for _ in range(int(rng.integers(1, 3))):
                e.insert(int(rng.integers(0, len(e) + 1)), e[int(rng.integers(0, len(e)))])
            layers[0][p_] = e
            dup += 1
    assert dup > 100
    hidx, hvec = F.write_hnsw_index(layers, list(range(n)), d), F.write_vector_file(v)
    g, o = BlockBasedHnsw(ctx, hidx, hvec, d), oracle.BlockBasedHnsw(hidx, hvec, d)
    q = (v[rng.integers(0, n, 30)] + rng.normal(0, 4, (30, d))).astype(np.float32)
    for k, ef in [(10, 100), (5, 16), (20, 256)]:
        o.stats()
        want = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        assert_result_rows(g.ann_search(q, k, ef), want, len(q))
        st = ctx.stats()
        assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded), (k, ef)


def test_hnsw_ties_and_duplicates(ctx, oracle):
    # many exact distance ties: pop order (largest id first) and eviction order must match
    from muopdb_amd.index import BlockBasedHnsw
    base = np.repeat(np.random.default_rng(3).integers(0, 3, (40, 6)).astype(np.float32), 25, axis=0)
    hidx, hvec = H.build_hnsw_files(oracle, base, list(range(1000)), max_neighbors=10, max_layers=3, ef_construction=40)
    g = BlockBasedHnsw(ctx, hidx, hvec, 6)
    o = oracle.BlockBasedHnsw(hidx, hvec, 6)
    q = base[::97][:10] + 0.0
    for k, ef in [(20, 30), (5, 5), (100, 128)]:
        assert_result_rows(g.ann_search(q, k, ef), o.ann_search(q, k, ef), len(q))


def test_hnsw_large_graph_uses_hbm_visited(ctx, oracle):
    # > ~1.1M points: the visited bitmap no longer fits in LDS -> HBM bitmap path, same answers
    from muopdb_amd.index import BlockBasedHnsw
    n, d = 1_300_000, 4
    rng = np.random.default_rng(8)
    v = rng.random((n, d), dtype=np.float32)
    # a cheap valid graph: ring + random long links (no need for quality, only for parity)
    nb = np.stack([(np.arange(n) + 1) % n, (np.arange(n) - 1) % n, rng.integers(0, n, n), rng.integers(0, n, n)], 1)
    indptr = np.arange(n + 1, dtype=np.uint64) * 4
    index = F.write_hnsw_index([(None, indptr, nb.reshape(-1).astype(np.uint32))], np.arange(n, dtype=np.uint64), d)
    vf = F.write_vector_file(v)
    g = BlockBasedHnsw(ctx, index, vf, d)
    o = oracle.BlockBasedHnsw(index, vf, d)
    q = rng.random((4, d), dtype=np.float32)
    assert_result_rows(g.ann_search(q, 10, 64), o.ann_search(q, 10, 64), 4)


# ----------------------------------------------------------------------------------- SPANN (K8)
def test_k8_spann_search(ctx, oracle):
    # rs/index/src/spann/index.rs:293-445
    from muopdb_amd.index import Spann, SearchParams
    v = _line_vectors()
    files, _, _ = H.build_spann_files(oracle, v, list(range(1000)), 10)
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    q = [[2.4, 3.4, 4.4, 5.4]]
    res = sp.search(q, SearchParams(2, 2))
    assert res.doc_ids(0) == [4, 3] and res.found[0] == 1
    assert sp.invalidate(4) and sp.is_invalidated(4) and not sp.invalidate(4)
    assert sp.search(q, SearchParams(2, 2)).doc_ids(0) == [3, 5]
    osp.invalidate(4)
    rng = np.random.default_rng(1)
    qs = (rng.random((40, 4)) * 1000).astype(np.float32)
    for params, op in [(SearchParams(2, 2), oracle.SearchParams(2, 2)),
                       (SearchParams(10, 100), oracle.SearchParams(10, 100)),
                       (SearchParams(5, 50).with_num_explored_centroids(4).with_centroid_distance_ratio(0.5),
                        oracle.SearchParams(5, 50, num_explored_centroids=4, centroid_distance_ratio=0.5)),
                       (SearchParams(3, 10).with_num_explored_centroids(0), oracle.SearchParams(3, 10, num_explored_centroids=0))]:
        r, ro = sp.search(qs, params), osp.search(qs, op)
        assert r.found.tolist() == ro.found.tolist()
        assert_result_rows(r, ro, len(qs))


def test_k8_spann_search_pq(ctx, oracle):
    # spann/index.rs:448-527: subdim 2 / 2 bits => top-5 scores all 0.0
    from muopdb_amd.index import Spann, SearchParams, ProductQuantizer
    v = _line_vectors()
    cb = H.train_pq_codebook(v, 2, 2)
    opq = oracle.ProductQuantizer(4, 2, 2, cb)
    files, _, _ = H.build_spann_files(oracle, v, list(range(1000)), 10, quantize=opq.quantize)
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"],
               ProductQuantizer(4, 2, 2, cb))
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"],
                       oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 2, 2, cb))
    res = sp.search([[2.4, 3.4, 4.4, 5.4]], SearchParams(5, 2))
    assert res.counts[0] == 5 and res.scores[0].tolist() == [0.0] * 5
    assert_result_rows(res, osp.search([[2.4, 3.4, 4.4, 5.4]], oracle.SearchParams(5, 2)), 1)


def test_spann_random_noq_and_pq(ctx, oracle):
    from muopdb_amd.index import Spann, SearchParams, ProductQuantizer
    rng = np.random.default_rng(21)
    v = H.sift_like(4000, 32, n_clusters=30, seed=9)
    doc = list(range(10, 4010))
    files, _, _ = H.build_spann_files(oracle, v, doc, 40, max_neighbors=8, max_layers=3, ef_construction=50)
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    q = (v[rng.integers(0, 4000, 50)] + rng.normal(0, 3, (50, 32))).astype(np.float32)
    p, op = SearchParams(10, 100).with_num_explored_centroids(8).with_centroid_distance_ratio(0.3), \
        oracle.SearchParams(10, 100, num_explored_centroids=8, centroid_distance_ratio=0.3)
    assert_result_rows(sp.search(q, p), osp.search(q, op), len(q))
    cb = H.train_pq_codebook(v[:2000], 8, 6, iters=3)
    opq = oracle.ProductQuantizer(32, 8, 6, cb)
    files, _, _ = H.build_spann_files(oracle, v, doc, 40, quantize=opq.quantize, max_neighbors=8, max_layers=3,
                                      ef_construction=50)
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"],
               ProductQuantizer(32, 8, 6, cb))
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"],
                       oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 6, cb))
    assert_result_rows(sp.search(q, p), osp.search(q, op), len(q))


def test_cpp_host_mirror_matches_ctypes_binding(ctx, oracle, tmp_path):
    """include/muopdb_host.hpp (C++ mirror of the reference surface) through the same C ABI:
    its rows must equal the ctypes binding's rows (which are oracle-checked above)."""
    import os
    import struct
    import subprocess
    from muopdb_amd.index import Spann, SearchParams, BlockBasedHnsw, BlockBasedIvf
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "muopdb_amd", "host_mirror_demo")
    assert os.path.exists(exe), "host_mirror_demo not built (run __graft_entry__.build())"
    rng = np.random.default_rng(5)
    v = H.sift_like(3000, 24, n_clusters=20, seed=3)
    doc = list(range(100, 3100))
    files, _, _ = H.build_spann_files(oracle, v, doc, 30, max_neighbors=8, max_layers=3, ef_construction=50)
    q = (v[rng.integers(0, 3000, 9)] + rng.normal(0, 3, (9, 24))).astype(np.float32)
    for name in ("hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors"):
        (tmp_path / name).write_bytes(files[name])
    (tmp_path / "queries.f32").write_bytes(q.tobytes())
    out = subprocess.run([exe, str(tmp_path), "24", "7", "60", "5", "0.25"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = {}
    assert "pq_kat ok" in out.stdout  # the reference's PQ known-answer test through the C++ Quantizer seams
    assert "pq_kat attach_ok" in out.stdout  # an attached HNSW handle on a second context gives the same rows
    for line in out.stdout.splitlines():
        t = line.split()
        if t[0] == "pq_kat":
            continue
        if t[2] == "none":
            got[(t[0], int(t[1]))] = None
        else:
            got[(t[0], int(t[1]))] = [(int(x.split(":")[0]), int(x.split(":")[1], 16)) for x in t[3:]]
            assert len(got[(t[0], int(t[1]))]) == int(t[2])

    def rows(res):
        return [[(int(i), struct.unpack("<I", struct.pack("<f", float(s)))[0]) for i, s in res.id_with_scores(qi)]
                if res.found[qi] else None for qi in range(res.b)]
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"])
    p = SearchParams(7, 60).with_num_explored_centroids(5).with_centroid_distance_ratio(0.25)
    want = {"spann": rows(sp.search(q, p)),
            "hnsw": rows(BlockBasedHnsw(ctx, files["hnsw_index"], files["hnsw_vectors"], 24).ann_search(q, 7, 60)),
            "ivf": rows(BlockBasedIvf(ctx, files["ivf_index"], files["ivf_vectors"]).search(q, 7, 5))}
    for name, rr in want.items():
        for i, r in enumerate(rr):
            assert got[(name, i)] == r, (name, i)


# ----------------------------------------------------------------------------------- multi-user (K9, K10)
def _multi(ctx, oracle, users, quant=None, oquant=None, **kw):
    from muopdb_amd.index import MultiSpannIndex
    cat = F.concat_multi_spann(users)
    g = MultiSpannIndex(ctx, cat["user_table"], 4 if "nf" not in kw else kw["nf"], cat["hnsw_index"],
                        cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], quant,
                        kw.get("shard_rank", 0), kw.get("shard_world", 1))
    o = oracle.MultiSpannIndex(cat["user_table"], 4 if "nf" not in kw else kw["nf"], cat["hnsw_index"],
                               cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    return g, o


def test_k9_multi_user(ctx, oracle):
    # rs/index/src/multi_spann/index.rs:358-412
    from muopdb_amd.index import SearchParams
    v = np.concatenate([_line_vectors(), np.array([[1.2, 2.2, 3.2, 4.2]], np.float32)])
    f0, _, _ = H.build_spann_files(oracle, v, list(range(1001)), 10)
    f1, _, _ = H.build_spann_files(oracle, _line_vectors(50) + 0.5, list(range(5000, 5050)), 3)
    big = (1 << 70) + 1
    g, o = _multi(ctx, oracle, {0: f0, big: f1})
    assert g.num_users() == 2
    res = g.search_for_user([0], [[1.4, 2.4, 3.4, 4.4]], SearchParams(3, 100))
    assert res.found[0] == 1 and res.doc_ids(0) == [1000, 3, 2]
    users = [big, 12345, 0, big]
    qs = np.array([[1.4, 2.4, 3.4, 4.4]] * 4, np.float32)
    r, ro = g.search_for_user(users, qs, SearchParams(2, 100)), o.search_for_user(users, qs, oracle.SearchParams(2, 100))
    assert r.found.tolist() == [1, 0, 1, 1] == ro.found.tolist()
    assert_result_rows(r, ro, 4)
    assert g.invalidate(0, 1000) and not g.invalidate(0, 1000) and not g.invalidate(777, 3)
    assert g.search_for_user([0], [[1.4, 2.4, 3.4, 4.4]], SearchParams(3, 100)).doc_ids(0) == [3, 2, 4]


def test_k10_ratio_filter(ctx, oracle):
    # rs/index/src/multi_spann/reader.rs:80-263
    from muopdb_amd.index import SearchParams, ProductQuantizer
    u0 = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)
    u1 = np.array([[9, 10, 11, 12]], np.float32)
    f0, _, _ = H.build_spann_files(oracle, u0, [1, 2], 2, centroids=u0.copy())
    f1, _, _ = H.build_spann_files(oracle, u1, [3], 1, centroids=u1.copy())
    g, o = _multi(ctx, oracle, {0: f0, 1: f1})
    p = SearchParams(3, 100)
    res = g.search_for_user([0, 1], [[1, 2, 3, 4]] * 2, p)
    assert res.doc_ids(0) == [1] and res.doc_ids(1) == [3]
    assert [d for d, _ in g.search_for_users([0, 1], [1, 2, 3, 4], p)] == [1, 3]  # snapshot.rs:39-66
    cb = H.train_pq_codebook(np.concatenate([u0, u1]), 2, 1)
    opq = oracle.ProductQuantizer(4, 2, 1, cb)
    g0, _, _ = H.build_spann_files(oracle, u0, [1, 2], 2, centroids=u0.copy(), quantize=opq.quantize)
    g1, _, _ = H.build_spann_files(oracle, u1, [3], 1, centroids=u1.copy(), quantize=opq.quantize)
    g, o = _multi(ctx, oracle, {0: g0, 1: g1}, ProductQuantizer(4, 2, 1, cb),
                  oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 2, 1, cb))
    res = g.search_for_user([0, 1], [[1, 2, 3, 4]] * 2, p)
    assert res.doc_ids(0) == [1] and res.doc_ids(1) == [3]


def test_multi_user_many_users_and_shards(ctx, oracle):
    from muopdb_amd.index import SearchParams
    rng = np.random.default_rng(31)
    users, allq, allu = {}, [], []
    for ui in range(12):
        n = int(rng.integers(150, 400))
        v = (rng.standard_normal((n, 24)) + ui).astype(np.float32)
        f, _, _ = H.build_spann_files(oracle, v, [1000 * ui + i for i in range(n)], int(rng.integers(3, 9)), seed=ui,
                                      max_neighbors=6, max_layers=2, ef_construction=30)
        users[(ui << 66) | ui] = f
        for _ in range(4):
            allq.append(v[rng.integers(0, n)] + rng.normal(0, 0.1, 24))
            allu.append((ui << 66) | ui)
    allq = np.asarray(allq, np.float32)
    g, o = _multi(ctx, oracle, users, nf=24)
    p, op = SearchParams(10, 50).with_num_explored_centroids(5).with_centroid_distance_ratio(1.0), \
        oracle.SearchParams(10, 50, num_explored_centroids=5, centroid_distance_ratio=1.0)
    full, ofull = g.search_for_user(allu, allq, p), o.search_for_user(allu, allq, op)
    assert_result_rows(full, ofull, len(allq))
    # posting-list sharding: the exact merge of the per-shard points blocks == unsharded answer
    shards = [_multi(ctx, oracle, users, nf=24, shard_rank=r, shard_world=3)[0] for r in range(3)]
    merged = shards[1].merge_shards(allu, [s.search_shard(allu, allq, p) for s in shards], len(allq), 10)
    assert_result_rows(merged, ofull, len(allq))


def _old_style_rows(parts, qi, k):
    """the inexact merge this suite used to accept: already remapped per-shard rows re-selected by (score, doc id)"""
    rows = []
    for pr in parts:
        rows += pr.id_with_scores(qi)
    rows.sort(key=lambda r: (r[1], r[0]))
    return [r[0] for r in rows[:k]]


@pytest.mark.parametrize("cpv", [1, 2])
@pytest.mark.parametrize("world", [8, 3])
def test_sharded_merge_is_exact_under_ties_ivfpq(ctx, oracle, world, cpv):
    """The reference selects its top-k by (distance, POINT id) over all probed lists and only then remaps and sorts by
    (score, doc id) (ivf/block_based/index.rs:250-286, then :298-332).  With PQ codes exact score ties are the norm, and a
    reindexed segment's doc ids are not monotone in point ids: the sharded result must still be the unsharded one row for
    row.  Low-entropy vectors + a 2-bit codebook (a handful of distinct distances), PERMUTED doc ids, posting lists dealt
    over `world` simulated ranks; cpv = 2 also puts one point into lists of different ranks (the reference keeps both)."""
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    rng = np.random.default_rng(100 + world + cpv)
    n, d, L, P = 4000, 16, 24, 9
    v = rng.integers(0, 3, (n, d)).astype(np.float32)
    cent = H.kmeans(v + rng.normal(0, 0.01, v.shape).astype(np.float32), L, iters=3, seed=5)
    cb = H.train_pq_codebook(v[:1500], 4, 2, iters=3)
    doc_ids = [int(x) for x in rng.permutation(n) + 50_000]           # not monotone in point ids
    doc_ids[7] += 1 << 90                                              # and a 128-bit one
    opq = oracle.ProductQuantizer(d, 4, 2, cb)
    index, vec, _ = H.build_ivf_files(v, doc_ids, cent, quantize=opq.quantize, clusters_per_vector=cpv)
    gq = ProductQuantizer(d, 4, 2, cb)
    o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 4, 2, cb))
    g = BlockBasedIvf(ctx, index, vec, gq)
    shards = [BlockBasedIvf(ctx, index, vec, gq, shard_rank=r, shard_world=world) for r in range(world)]
    q = (v[rng.integers(0, n, 40)] + rng.normal(0, 0.3, (40, d))).astype(np.float32)
    probes = g.find_nearest_centroids(q, P)
    stale = 0
    for k in (1, 10, 17):
        want = o.search(q, k, probes=probes)
        full = g.search_with_centroids_and_remap(q, probes, k)
        assert_result_rows(full, want, len(q))
        blocks = [s.search_shard(q, k, probes=probes) for s in shards]
        merged = shards[world - 1].merge_shards(blocks, len(q), k)     # any rank: the doc-id table is replicated
        assert_result_rows(merged, want, len(q))
        assert H.result_rows(merged, len(q)) == H.result_rows(full, len(q))
        parts = [s.search_with_centroids_and_remap(q, probes, k) for s in shards]
        stale += sum(_old_style_rows(parts, qi, k) != full.doc_ids(qi) for qi in range(len(q)))
    assert stale > 0, "the case must contain rank-k ties that a (score, doc id) merge resolves differently"
    # a per-call planner filter is applied by every rank alike
    from muopdb_amd.index import allow_bitmap
    bm = allow_bitmap(rng.choice(n, n // 2, replace=False), n)
    fblocks = [s.search_shard(q, 10, probes=probes, planner=bm) for s in shards]
    assert H.result_rows(shards[0].merge_shards(fblocks, len(q), 10), len(q)) == \
        H.result_rows(g.search_with_centroids_and_remap(q, probes, 10, planner=bm), len(q))


def test_sharded_merge_is_exact_under_ties_multi_user_spann(ctx, oracle):
    """The same for MultiSpannIndex::search_for_user with PQ posting lists: lists l % 8 per user, permuted doc ids, mixed
    users (one unknown) in one batch — merged == unsharded == oracle, and `found` survives the merge."""
    from muopdb_amd.index import MultiSpannIndex, ProductQuantizer, SearchParams
    rng = np.random.default_rng(77)
    d, world = 16, 8
    cb = H.train_pq_codebook(rng.integers(0, 3, (1500, d)).astype(np.float32), 4, 2, iters=3)
    opq = oracle.ProductQuantizer(d, 4, 2, cb)
    users, allq, allu = {}, [], []
    for ui in range(3):
        n = 900 + 200 * ui
        v = rng.integers(0, 3, (n, d)).astype(np.float32)
        docs = [int(x) for x in rng.permutation(n) + 10_000 * (ui + 1)]
        f, _, _ = H.build_spann_files(oracle, v, docs, 10 + ui, quantize=opq.quantize, seed=ui, max_neighbors=6, max_layers=2,
                                      ef_construction=30)
        uid = (ui << 66) | (ui + 1)
        users[uid] = f
        for _ in range(10):
            allq.append(v[rng.integers(0, n)] + rng.normal(0, 0.3, d))
            allu.append(uid)
    allq.append(allq[0]); allu.append(424242)                          # an unknown user: None
    allq = np.asarray(allq, np.float32)
    cat = F.concat_multi_spann(users)
    a = (cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    gq, oq = ProductQuantizer(d, 4, 2, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 4, 2, cb)
    g, o = MultiSpannIndex(ctx, *a, gq), oracle.MultiSpannIndex(*a, oq)
    shards = [MultiSpannIndex(ctx, *a, gq, shard_rank=r, shard_world=world) for r in range(world)]
    stale = 0
    for k in (1, 10):
        p = SearchParams(k, 40).with_num_explored_centroids(6).with_centroid_distance_ratio(2.0)
        op = oracle.SearchParams(k, 40, num_explored_centroids=6, centroid_distance_ratio=2.0)
        want, full = o.search_for_user(allu, allq, op), g.search_for_user(allu, allq, p)
        assert_result_rows(full, want, len(allq))
        blocks = [s.search_shard(allu, allq, p) for s in shards]
        merged = shards[3].merge_shards(allu, blocks, len(allq), k)
        assert_result_rows(merged, want, len(allq))
        assert merged.found.tolist() == full.found.tolist() and merged.found[-1] == 0
        parts = [s.search_for_user(allu, allq, p) for s in shards]
        stale += sum(_old_style_rows(parts, qi, k) != full.doc_ids(qi) for qi in range(len(allq)))
    assert stale > 0


def test_merge_shards_device(ctx):
    torch = pytest.importorskip("torch")
    from muopdb_amd import lib as L
    world, b, k = 4, 9, 6
    rng = np.random.default_rng(2)
    docs = rng.integers(0, 50, (world, b, k, 2)).astype(np.uint64)
    docs[..., 1] = rng.integers(0, 2, (world, b, k))
    scores = np.sort(rng.integers(0, 6, (world, b, k)).astype(np.float32), axis=2)
    counts = rng.integers(0, k + 1, (world, b)).astype(np.uint32)
    dev = torch.device("cuda:0")
    t_docs = torch.from_numpy(docs.view(np.int64)).to(dev)
    t_sc = torch.from_numpy(scores).to(dev)
    t_cn = torch.from_numpy(counts.view(np.int32)).to(dev)
    o_docs = torch.zeros((b, k, 2), dtype=torch.int64, device=dev)
    o_sc = torch.zeros((b, k), dtype=torch.float32, device=dev)
    o_cn = torch.zeros(b, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.mdb_merge_shards(ctx.h, C.c_void_p(t_docs.data_ptr()), C.c_void_p(t_sc.data_ptr()),
                                       C.c_void_p(t_cn.data_ptr()), C.c_size_t(world), C.c_size_t(b), C.c_size_t(k),
                                       C.c_void_p(o_docs.data_ptr()), C.c_void_p(o_sc.data_ptr()),
                                       C.c_void_p(o_cn.data_ptr())))
    ctx.sync()
    od, osc, ocn = o_docs.cpu().numpy().view(np.uint64), o_sc.cpu().numpy(), o_cn.cpu().numpy()
    for qi in range(b):
        rows = []
        for w in range(world):
            for j in range(int(counts[w, qi])):
                rows.append((float(scores[w, qi, j]), (int(docs[w, qi, j, 1]) << 64) | int(docs[w, qi, j, 0])))
        rows.sort()
        rows = rows[:k]
        assert int(ocn[qi]) == len(rows)
        got = [(float(osc[qi, j]), (int(od[qi, j, 1]) << 64) | int(od[qi, j, 0])) for j in range(len(rows))]
        assert got == rows
    # the packed form: one block per rank as ONE all-gather delivers them (PackedTopkGather with no process group = one
    # rank; here the `world` blocks are laid out by hand) -> mdb_merge_shards_packed == mdb_merge_shards
    from muopdb_amd import distributed as D
    nb = D.block_bytes(b, k)
    assert nb == int(ctx.lib.mdb_shard_block_bytes(C.c_size_t(b), C.c_size_t(k))) and nb % 16 == 0
    recv = torch.zeros(world * nb, dtype=torch.uint8, device=dev)
    for w in range(world):
        vi, vs, vc = D.block_views(recv[w * nb:(w + 1) * nb], b, k)
        vi.copy_(t_docs[w]); vs.copy_(t_sc[w]); vc.copy_(t_cn[w])
    p_docs, p_sc, p_cn = torch.zeros_like(o_docs), torch.zeros_like(o_sc), torch.zeros_like(o_cn)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.mdb_merge_shards_packed(ctx.h, C.c_void_p(recv.data_ptr()), C.c_size_t(world), C.c_size_t(b), C.c_size_t(k),
                                              C.c_void_p(p_docs.data_ptr()), C.c_void_p(p_sc.data_ptr()), C.c_void_p(p_cn.data_ptr())))
    ctx.sync()
    assert torch.equal(p_docs, o_docs) and torch.equal(p_sc, o_sc) and torch.equal(p_cn, o_cn)
    one = D.PackedTopkGather(ctx, b, k, dev)          # world 1: gather is a copy, merge re-ranks by (score, doc id)
    one.ids.copy_(t_docs[0]); one.scores.copy_(t_sc[0]); one.counts.copy_(t_cn[0])
    torch.cuda.synchronize()
    gd, gs, gc = one.gather_merge()
    ctx.sync()
    assert torch.equal(gc, t_cn[0])
    # mdb_allgather_merge refuses a null communicator instead of touching RCCL
    assert ctx.lib.mdb_allgather_merge(ctx.h, None, C.c_void_p(recv.data_ptr()), C.c_void_p(recv.data_ptr()), C.c_size_t(1), C.c_size_t(b),
                                       C.c_size_t(k), C.c_void_p(p_docs.data_ptr()), C.c_void_p(p_sc.data_ptr()), None) == L.MDB_ERR_INVALID_ARG


# ----------------------------------------------------------------------------------- planner hook
@pytest.mark.parametrize("pq", [False, True])
def test_planner_filter_hook(ctx, oracle, pq):
    """scan_posting_list's planner consumer (ivf/block_based/index.rs:214-226) as per-query allow bitmaps:
    the reference's own test keeps the even doc ids (multi_spann/index.rs:850-880); here per-query random
    subsets too, through IVF, SPANN and multi-user SPANN, NoQ and PQ, against the oracle."""
    from muopdb_amd.index import BlockBasedIvf, MultiSpannIndex, ProductQuantizer, SearchParams, Spann, allow_bitmap
    rng = np.random.default_rng(31 + pq)
    n, d = 3000, 32
    v = H.sift_like(n, d, n_clusters=25, seed=6)
    doc = list(range(n))  # doc id == point id, so "even doc ids" == even point ids
    quant = oquant = quantize = None
    if pq:
        cb = H.train_pq_codebook(v[:1500], 8, 6, iters=3)
        opq = oracle.ProductQuantizer(d, 8, 6, cb)
        quant, oquant, quantize = ProductQuantizer(d, 8, 6, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 6, cb), opq.quantize
    files, cent, _ = H.build_spann_files(oracle, v, doc, 30, quantize=quantize, max_neighbors=8, max_layers=3, ef_construction=50)
    q = (v[rng.integers(0, n, 16)] + rng.normal(0, 3, (16, d))).astype(np.float32)
    even = allow_bitmap(np.arange(0, n, 2), n)
    per_query = np.stack([allow_bitmap(rng.choice(n, size=int(rng.integers(1, n)), replace=False), n) for _ in range(16)])
    # IVF
    g = BlockBasedIvf(ctx, files["ivf_index"], files["ivf_vectors"], quant)
    o = oracle.BlockBasedIvf(files["ivf_index"], files["ivf_vectors"], oquant)
    for bm in (even, per_query):
        with oracle.planner_filter(bm):
            want = o.search(q, 10, num_probes=12)
        got = g.search(q, 10, 12, planner=bm)
        assert_result_rows(got, want, len(q))
        if bm is even:
            assert all(x % 2 == 0 for i in range(len(q)) for x in got.doc_ids(i))
    assert_result_rows(g.search(q, 10, 12), o.search(q, 10, num_probes=12), len(q))   # a filter never outlives its call
    # SPANN
    sp = Spann(ctx, files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"], quant)
    osp = oracle.Spann(files["hnsw_index"], files["hnsw_vectors"], files["ivf_index"], files["ivf_vectors"], oquant)
    p, op = SearchParams(10, 50).with_num_explored_centroids(6), oracle.SearchParams(10, 50, num_explored_centroids=6)
    with oracle.planner_filter(per_query):
        want = osp.search(q, op)
    assert_result_rows(sp.search(q, p, planner=per_query), want, len(q))
    # multi-user (one user): bitmaps are over the user's local point ids
    cat = F.concat_multi_spann({5: files})
    ms = MultiSpannIndex(ctx, cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], quant)
    oms = oracle.MultiSpannIndex(cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    with oracle.planner_filter(even):
        want = oms.search_for_user([5] * len(q), q, op)
    got = ms.search_for_user([5] * len(q), q, p, planner=even)
    assert_result_rows(got, want, len(q))
    assert all(x % 2 == 0 for i in range(len(q)) for x in got.doc_ids(i))


def test_hnsw_edge_outside_vector_storage_is_rejected_at_load(ctx, oracle):
    # the reference would fail reading the vector of such a neighbour; here the graph file is refused
    from muopdb_amd.index import BlockBasedHnsw
    from muopdb_amd import lib as L
    v = np.random.default_rng(1).standard_normal((50, 8)).astype(np.float32)
    layers = [{i: [(i + 1) % 50, 77 if i == 3 else (i + 2) % 50] for i in range(50)}]
    hidx, hvec = F.write_hnsw_index(layers, list(range(50)), 8), F.write_vector_file(v)
    with pytest.raises(L.MuopdbError) as e:
        BlockBasedHnsw(ctx, hidx, hvec, 8)
    assert e.value.status == 2  # MDB_ERR_FORMAT


def test_concurrent_searches_from_host_threads(ctx, oracle):
    """Quantizer: Send + Sync / one query per tokio task in the reference: handles must serve
    concurrent callers.  Four host threads search one IVF and one HNSW handle at once (ctypes drops the GIL);
    every row must still equal the serial result."""
    import threading
    from muopdb_amd.index import BlockBasedHnsw, BlockBasedIvf
    rng = np.random.default_rng(41)
    v = H.sift_like(4000, 32, n_clusters=30, seed=8)
    files, _, _ = H.build_spann_files(oracle, v, list(range(4000)), 40, max_neighbors=8, max_layers=3, ef_construction=50)
    ivf = BlockBasedIvf(ctx, files["ivf_index"], files["ivf_vectors"])
    hidx, hvec = H.build_hnsw_files(oracle, v[:1500], list(range(1500)), max_neighbors=12, max_layers=3, ef_construction=60)
    hn = BlockBasedHnsw(ctx, hidx, hvec, 32)
    qs = [(v[rng.integers(0, 1500, 20)] + rng.normal(0, 2, (20, 32))).astype(np.float32) for _ in range(4)]
    want = [(ivf.search(q, 10, 8), hn.ann_search(q, 10, 100)) for q in qs]
    errors = []

    def worker(i):
        try:
            for _ in range(15):
                a, b = ivf.search(qs[i], 10, 8), hn.ann_search(qs[i], 10, 100)
                for j in range(20):
                    if a.doc_ids(j) != want[i][0].doc_ids(j) or b.doc_ids(j) != want[i][1].doc_ids(j):
                        errors.append((i, j))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_hnsw_attached_handles_share_one_resident_index(ctx, oracle):
    """mdb_hnsw_attach: handles on other contexts (own stream + scratch) over the SAME device arrays.  Rows equal the
    owner's and the oracle's from four host threads at once, and the memory outlives the owner handle when that is
    freed first."""
    import threading
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedHnsw
    rng = np.random.default_rng(43)
    v = H.sift_like(3000, 48, n_clusters=25, seed=9)
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(3000)), max_neighbors=12, max_layers=4, ef_construction=60)
    owner = BlockBasedHnsw(ctx, hidx, hvec, 48)
    o = oracle.BlockBasedHnsw(hidx, hvec, 48)
    ctxs = [L.Context(0) for _ in range(4)]
    views = [owner.attach(c) for c in ctxs]
    views.append(views[0].attach(ctxs[1]))  # attaching to an attached handle attaches to its owner
    qs = [(v[rng.integers(0, 3000, 16)] + rng.normal(0, 2, (16, 48))).astype(np.float32) for _ in range(5)]
    want = [o.ann_search(q, 10, 120) for q in qs]
    owner.close()  # the device arrays stay until the last view goes
    errors = []

    def worker(i):
        try:
            for _ in range(10):
                assert_result_rows(views[i].ann_search(qs[i], 10, 120), want[i], 16)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    for vw in views:
        vw.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("pq", [False, True])
def test_reindexed_segment_sharded_x8_equals_unsharded_equals_oracle(ctx, oracle, pq):
    """SURVEY §8f-2 / VERDICT r3 #8: an IVF whose vectors sit in up to two posting lists (max_clusters_per_vector = 2) is renumbered
    by IvfBuilder::reindex's algorithm (muopdb_amd.build.reindex, ivf/builder.rs:596-761: list-contiguous new point ids around the
    shared "stopping points") — doc ids are then NOT monotone in point ids and the same point is scored from two lists (no dedup,
    index.rs:250-286).  The segment searched as 8 list shards + the exact (distance, point id) merge == unsharded == the oracle,
    row for row; the mapping file round-trips through the segment tree."""
    import os
    import tempfile
    from muopdb_amd import build as B
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    rng = np.random.default_rng(77)
    n, d, nl, k, P = 4000, 24, 48, 10, 10
    v = H.sift_like(n, d, n_clusters=30, seed=12)
    cent = H.kmeans(v, nl, iters=3, seed=2)
    dist = ((v[:, None, :] - cent[None, :, :]) ** 2).sum(2)
    near = np.argsort(dist, axis=1, kind="stable")[:, :2]
    lists = [[] for _ in range(nl)]
    for pid in range(n):                                       # second list only for a third of the vectors
        lists[int(near[pid, 0])].append(pid)
        if pid % 3 == 0:
            lists[int(near[pid, 1])].append(pid)
    docs = np.asarray([int(x) for x in (rng.permutation(n).astype(np.int64) * 7 + 3)], dtype=object)
    new_lists, ndocs, nvec, mapping = B.reindex(lists, docs, v)
    assert [int(x) for x in mapping] == [x & 0xFFFFFFFF for x in oracle.reassigned_ids(lists, n)]
    assert all(np.all(np.diff(np.asarray(pl, np.int64)) > 0) for pl in new_lists if len(pl) > 1)   # Elias-Fano needs ascending lists
    assert any(int(ndocs[i]) > int(ndocs[i + 1]) for i in range(n - 1))                              # doc ids not monotone in point ids
    if pq:
        cb = H.train_pq_codebook(v[:1500], 8, 5, iters=3)
        opq = oracle.ProductQuantizer(d, 8, 5, cb)
        stored, quant, oquant, qd = opq.quantize(nvec), ProductQuantizer(d, 8, 5, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 5, cb), d // 8
    else:
        stored, quant, oquant, qd = nvec, None, oracle.Quant(oracle.QUANT_NONE, oracle.METRIC_L2), d
    index = F.write_ivf_index(cent, [int(x) for x in ndocs], new_lists, quantized_dimension=qd)
    vec = F.write_vector_file(stored)
    full = BlockBasedIvf(ctx, index, vec, quant)
    o = oracle.BlockBasedIvf(index, vec, oquant)
    q = (v[rng.integers(0, n, 32)] + rng.normal(0, 3, (32, d))).astype(np.float32)
    want = o.search(q, k, num_probes=P)
    got = full.search(q, k, P)
    assert_result_rows(got, want, len(q))
    probes = full.find_nearest_centroids(q, P)
    blocks, sh = [], None
    for r in range(8):
        if sh is not None:
            sh.close()
        sh = BlockBasedIvf(ctx, index, vec, quant, shard_rank=r, shard_world=8)
        blocks.append(sh.search_shard(q, k, probes=probes))
    merged = sh.merge_shards(blocks, len(q), k)
    sh.close()
    assert_result_rows(merged, want, len(q))
    with tempfile.TemporaryDirectory() as tmp:                  # the mapping travels with the segment (reassigned_mappings.<user_id>)
        hn = F.write_hnsw_index([{0: [1], 1: [0]}], [0, 1], d)
        cat = F.concat_multi_spann({9: dict(hnsw_index=hn, hnsw_vectors=F.write_vector_file(cent[:2]), ivf_index=index, ivf_vectors=vec)})
        F.write_segment(os.path.join(tmp, "seg"), cat, d, reassigned={9: mapping})
        back = F.read_segment(os.path.join(tmp, "seg"))
        assert np.array_equal(back["reassigned"][9], mapping) and back["ivf_index"] == cat["ivf_index"]
    full.close()


@pytest.mark.parametrize("d,metric,nq", [(128, 0, 40), (16, 1, 33), (48, 0, 7)])
def test_hnsw_wide_beam_and_split_path_equal_oracle(ctx, oracle, d, metric, nq):
    """The default routes of round 4 on a 5-layer graph: batches >= 32 take the split path (top layers + table pass in one launch, then
    layer 1, then layer 0), 256 < ef <= 448 the 8-register beam, ef = 449 the general kernel — rows AND traversal counters equal the
    oracle's everywhere, including duplicate vectors (exact ties) and ef below / at / above the beam limits."""
    from muopdb_amd.index import BlockBasedHnsw, NoQuantizer
    rng = np.random.default_rng(41 + d)
    n = 4000
    v = H.sift_like(n, d, n_clusters=16, seed=9)
    v[n // 2:n // 2 + 200] = v[7]                      # 200 exact duplicates: ties with furthest
    if metric == 1:
        v = (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(n)), max_neighbors=6, max_layers=6, ef_construction=40, metric=metric, seed=3)
    g = BlockBasedHnsw(ctx, hidx, hvec, d, NoQuantizer(d, metric))
    o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_NONE, metric))
    q = (v[rng.integers(0, n, nq)] + rng.normal(0, 0.05 if metric == 1 else 4, (nq, d))).astype(np.float32)
    q[0] = v[7]
    for k, ef in [(10, 200), (10, 256), (10, 257), (30, 400), (10, 448), (10, 449), (5, 64)]:
        o.stats()
        want = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        got = g.ann_search(q, k, ef)
        st = ctx.stats()
        assert_result_rows(got, want, len(q))
        assert (st["distance_evals"], st["expanded_nodes"]) == (evals, expanded), (k, ef)


@pytest.mark.parametrize("world", [2, 3])
def test_user_and_batch_partitionings_on_the_device(ctx, oracle, world):
    """bench.py --shard users / --shard batch (SURVEY 8e 'measure both'): simulated ranks on ONE GPU, the very calls the bench issues —
    a rank's MultiSpannIndex over ITS rows of the user table (slot u -> rank u % world), mdb_multi_spann_search on device-resident
    routed queries straight into the RowsExchange send block, the blocks side by side as the all-gather leaves them, permute() —
    must equal the unsharded index's rows and the oracle's; the same for replicas of one IVF-PQ index answering batch slices."""
    import ctypes as C
    import torch
    from muopdb_amd import distributed as D
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf, MultiSpannIndex, ProductQuantizer, SearchParams
    rng = np.random.default_rng(5 + world)
    d, k, U = 24, 6, 7
    users, slots_q = {}, []
    for ui in range(U):
        n = int(rng.integers(200, 500))
        v = (rng.standard_normal((n, d)) + ui).astype(np.float32)
        users[100 + ui], _, _ = H.build_spann_files(oracle, v, [1000 * ui + i for i in range(n)], int(rng.integers(4, 9)), seed=ui,
                                                    max_neighbors=6, max_layers=2, ef_construction=30)
        slots_q += [(ui, v[rng.integers(0, n)] + rng.normal(0, 0.1, d)) for _ in range(int(rng.integers(1, 5)))]
    order = rng.permutation(len(slots_q))
    slots = [slots_q[i][0] for i in order]
    q = np.asarray([slots_q[i][1] for i in order], np.float32)
    b = len(slots)
    cat = F.concat_multi_spann(users)
    a = (d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    table = np.frombuffer(bytes(cat["user_table"]), np.uint8).reshape(U, -1)
    full = MultiSpannIndex(ctx, cat["user_table"], *a)
    ofull = oracle.MultiSpannIndex(cat["user_table"], *a)
    p = SearchParams(k, 40).with_num_explored_centroids(4).with_centroid_distance_ratio(0.5)
    op = oracle.SearchParams(k, 40, num_explored_centroids=4, centroid_distance_ratio=0.5)
    uids = [100 + s_ for s_ in slots]
    want = full.search_for_user(uids, q, p)
    assert_result_rows(want, ofull.search_for_user(uids, q, op), b)
    routes = D.route_by_user(slots, world)
    qd = torch.from_numpy(q).cuda()
    pc = p.to_c()
    exs = []
    for r in range(world):
        ex = D.RowsExchange(b, k, routes, r, "cuda")
        ms = MultiSpannIndex(ctx, table[D.users_of_rank(U, r, world)].tobytes(), *a)
        assert ms.num_users() == len(D.users_of_rank(U, r, world))
        ql = ex.local_queries(qd).contiguous()
        if ex.n_local:
            ctx.check(ctx.lib.mdb_multi_spann_search(ms.h, L.u128_array([uids[i] for i in routes[r]]), C.c_void_p(ql.data_ptr()), C.c_size_t(ex.n_local),
                                                     C.byref(pc), C.c_int(L.MEM_DEVICE), C.c_void_p(ex.ids.data_ptr()), C.c_void_p(ex.scores.data_ptr()),
                                                     C.c_void_p(ex.counts.data_ptr()), C.c_void_p(ex.found.data_ptr())))
        ctx.sync()
        exs.append(ex)
        ms.close()

    def check(exs_, want_, with_found):
        recv = torch.cat([e.send for e in exs_])
        for e in exs_:                                    # every rank ends with the same rows
            e.recv.copy_(recv)
            ids, sc, cn, fo = (t.cpu().numpy() for t in e.permute())
            for i in range(b):
                c_ = int(want_.counts[i])
                assert int(cn[i]) == c_ and (not with_found or int(fo[i]) == int(want_.found[i]))
                got = [(int(lo_) | (int(hi_) << 64), np.float32(s_).tobytes()) for lo_, hi_, s_ in
                       zip(ids[i, :c_, 0].view(np.uint64), ids[i, :c_, 1].view(np.uint64), sc[i, :c_])]
                assert got == [(int(doc), np.float32(s_).tobytes()) for doc, s_ in want_.id_with_scores(i)], i
    check(exs, want, True)
    full.close()
    # ---- replicas + batch slices of one IVF-PQ index
    n = 6000
    v = H.sift_like(n, 32, n_clusters=20, seed=9)
    cent = H.kmeans(v, 24, iters=3, seed=2)
    cb = H.train_pq_codebook(v[:2000], 8, 4, iters=3)
    pq = ProductQuantizer(32, 8, 4, cb)
    index, vec, _ = H.build_ivf_files(v, [7 * i + 3 for i in range(n)], cent, quantize=oracle.ProductQuantizer(32, 8, 4, cb).quantize)
    ivf = BlockBasedIvf(ctx, index, vec, pq)
    q2 = (v[rng.integers(0, n, b)] + rng.normal(0, 4, (b, 32))).astype(np.float32)
    want2 = ivf.search(q2, k, 5)
    assert_result_rows(want2, oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 4, cb)).search(q2, k, num_probes=5), b)
    q2d = torch.from_numpy(q2).cuda()
    routes2 = D.route_by_batch(b, world)
    exs2 = []
    for r in range(world):
        ex = D.RowsExchange(b, k, routes2, r, "cuda")
        ql = ex.local_queries(q2d)
        assert ex.contiguous and ql.data_ptr() == q2d.data_ptr() + routes2[r][0] * 32 * 4     # a view: the rows are read in place
        ctx.check(ctx.lib.mdb_ivf_search(ivf.h, C.c_void_p(ql.data_ptr()), C.c_size_t(ex.n_local), None, C.c_size_t(5), C.c_size_t(k),
                                         C.c_int(L.MEM_DEVICE), C.c_void_p(ex.ids.data_ptr()), C.c_void_p(ex.scores.data_ptr()),
                                         C.c_void_p(ex.counts.data_ptr())))
        ctx.sync()
        exs2.append(ex)
    check(exs2, want2, False)
    ivf.close()


def test_multi_spann_probe_rows_shared_closure(ctx, oracle):
    """List shards with the centroid stage run ONCE per (user, query) pair (mdb_multi_spann_probes on a slice of the batch, the
    rows concatenated as an all-gather would, mdb_multi_spann_search_shard_probes on every shard): the probe rows are the same on
    every shard and for every slicing, each POINTS block is byte-identical to search_shard's, and the merge == unsharded == oracle
    (Spann::search, spann/index.rs:211-266).  A row with a count beyond the row is clamped; a foreign list id is skipped and
    reported, not read."""
    from muopdb_amd import lib as L_
    from muopdb_amd.index import MultiSpannIndex, SearchParams
    from muopdb_amd.lib import MuopdbError
    rng = np.random.default_rng(91)
    d, world = 16, 4
    users, allq, allu = {}, [], []
    for ui in range(4):
        n = 700 + 150 * ui
        v = rng.integers(0, 4, (n, d)).astype(np.float32)
        docs = [int(x) for x in rng.permutation(n) + 5_000 * (ui + 1)]
        f, _, _ = H.build_spann_files(oracle, v, docs, 9 + ui, seed=ui, max_neighbors=6, max_layers=2, ef_construction=30)
        uid = (ui << 70) | (ui + 3)
        users[uid] = f
        for _ in range(7):
            allq.append(v[rng.integers(0, n)] + rng.normal(0, 0.3, d))
            allu.append(uid)
    allq.insert(5, allq[0]); allu.insert(5, 999_999)                  # an unknown user inside the batch: None
    allq = np.asarray(allq, np.float32)
    b = len(allq)
    cat = F.concat_multi_spann(users)
    a = (cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    g, o = MultiSpannIndex(ctx, *a), oracle.MultiSpannIndex(*a)
    graphs = {u: oracle.BlockBasedHnsw(f["hnsw_index"], f["hnsw_vectors"], d) for u, f in users.items()}
    shards = [MultiSpannIndex(ctx, *a, None, r, world) for r in range(world)]
    for k, ne, ratio in ((1, 1, 0.1), (5, 6, 2.0), (10, None, 0.5), (4, 0, 0.1)):
        p = SearchParams(k, 40).with_centroid_distance_ratio(ratio)
        op = oracle.SearchParams(k, 40, centroid_distance_ratio=ratio)
        if ne is not None:
            p.with_num_explored_centroids(ne)
            op = oracle.SearchParams(k, 40, num_explored_centroids=ne, centroid_distance_ratio=ratio)
        want = o.search_for_user(allu, allq, op)
        rows = g.probes(allu, allq, p)
        words = rows.shape[1]
        assert words == max(k if ne is None else ne, 1) + 2
        assert rows[5, 0] == 0 and rows[5, 1] == 0 and (rows[:, 0] <= words - 2).all()
        nexp = k if ne is None else ne
        for i in range(b):                                              # the rows against the oracle: ann_search + the ratio filter
            if allu[i] not in graphs:
                continue
            near = graphs[allu[i]].ann_search(allq[i:i + 1], max(nexp, 1), 40)
            ids, sc = near.doc_ids(0)[:nexp], near.scores[0, :min(int(near.counts[0]), nexp)]
            kept = [c for c, s_ in zip(ids, sc) if np.float32(s_ - sc.min()) <= np.float32(sc.min() * np.float32(ratio))] if ids else []
            assert rows[i, 1] == (1 if ids else 0) and rows[i, 0] == len(kept) and rows[i, 2:2 + len(kept)].tolist() == kept, (k, ne, i)
        for r in range(world):                                          # every shard holds every centroid graph: the same rows
            assert np.array_equal(shards[r].probes(allu, allq, p), rows)
        per = (b + world - 1) // world                                  # the slices ProbeRowsShare hands the ranks
        parts = [shards[r].probes(allu[r * per:(r + 1) * per], allq[r * per:(r + 1) * per], p) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), rows)
        blocks = [s.search_shard_probes(allu, allq, p, rows) for s in shards]
        for s, blk in zip(shards, blocks):
            assert np.array_equal(blk, s.search_shard(allu, allq, p))
        merged = shards[2].merge_shards(allu, blocks, b, k)
        assert_result_rows(merged, want, b)
        assert merged.found.tolist() == g.search_for_user(allu, allq, p).found.tolist() and merged.found[5] == 0
    # empty batch, missing buffers
    p0 = SearchParams(5, 40).with_num_explored_centroids(6)
    assert g.probes([], np.zeros((0, d), np.float32), p0).shape == (0, 8)
    pc0 = p0.to_c()
    one = L_.u128_array([allu[0]])
    assert ctx.lib.mdb_multi_spann_probes(g.h, one, L_.ptr(allq[:1], C.c_float), C.c_size_t(1), C.byref(pc0), C.c_int(L_.MEM_HOST), None) == 1
    assert ctx.lib.mdb_multi_spann_search_shard_probes(g.h, one, L_.ptr(allq[:1], C.c_float), C.c_size_t(1), C.byref(pc0), C.c_int(L_.MEM_HOST), None,
                                                       None, C.c_size_t(0), C.c_size_t(0), None) == 1
    # hostile rows: a count beyond the row reads no further than the row; a list id the user does not have is an error, not a read
    p = SearchParams(5, 40).with_num_explored_centroids(6).with_centroid_distance_ratio(2.0)
    rows = g.probes(allu, allq, p)
    full = rows.copy()
    full[:, 0] = np.where(full[:, 1] != 0, 1000, 0)
    exact = np.array([r_[0] == 6 for r_ in rows])
    blk_big, blk_ref = shards[0].search_shard_probes(allu, allq, p, full), shards[0].search_shard(allu, allq, p)
    if exact.all():
        assert np.array_equal(blk_big, blk_ref)
    bad = rows.copy()
    bad[0, 2] = 1 << 30
    with pytest.raises(MuopdbError):
        shards[0].search_shard_probes(allu, allq, p, bad)
    assert np.array_equal(shards[0].search_shard_probes(allu, allq, p, rows), blk_ref)   # the handle is fine afterwards
