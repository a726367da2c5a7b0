"""GPU parity tests (-m gpu): every call goes through the C ABI (libmuopdb_hip.so) and is
compared with the CPU oracle on the same seeded inputs.  Bar: neighbour ids bit-exact; f32
scores bit-exact too (the kernels keep the reference's lane association), asserted at the
north-star tolerance 1e-4 relative AND, separately, exactly.
"""
import numpy as np
import pytest

from muopdb_amd import formats as F
from tests import helpers as H

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # BASELINE.json north_star: "distances within 1e-4 relative for f32"


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


def assert_scores(gpu, cpu):
    gpu, cpu = np.asarray(gpu, np.float32), np.asarray(cpu, np.float32)
    fin = np.isfinite(cpu)
    assert np.array_equal(np.isfinite(gpu), fin)
    assert np.allclose(gpu[fin], cpu[fin], rtol=RTOL, atol=0)
    assert np.array_equal(gpu[fin].view(np.uint32), cpu[fin].view(np.uint32)), "scores are not bit-identical"


def assert_result_rows(res, ores, b):
    for qi in range(b):
        assert int(res.counts[qi]) == int(ores.counts[qi])
        assert res.doc_ids(qi) == ores.doc_ids(qi), "query %d" % qi
        n = int(res.counts[qi])
        assert_scores(res.scores[qi, :n], ores.scores[qi, :n])


# ----------------------------------------------------------------------------------- D1/D2 seams
@pytest.mark.parametrize("d", [1, 3, 4, 5, 8, 9, 12, 15, 16, 17, 24, 30, 31, 32, 33, 100, 128, 768, 1000])
def test_pair_distances(ctx, oracle, d):
    rng = np.random.default_rng(d)
    a = (rng.standard_normal((64, d)) * 10).astype(np.float32)
    b = (rng.standard_normal((64, d)) * 10).astype(np.float32)
    l2 = ctx.l2_distance(a, b)
    l2sq = ctx.l2_distance(a, b, squared=True)
    dot = ctx.dot_distance(a, b)
    assert_scores(l2, [oracle.l2(x, y) for x, y in zip(a, b)])
    assert_scores(l2sq, [oracle.l2_squared(x, y) for x, y in zip(a, b)])
    assert_scores(dot, [oracle.dot(x, y) for x, y in zip(a, b)])


def test_sqrt_is_correctly_rounded(ctx):
    rng = np.random.default_rng(0)
    x = np.abs(rng.standard_normal(4096).astype(np.float32)) * np.float32(10) ** rng.integers(-10, 10, 4096).astype(np.float32)
    a = np.sqrt(x.astype(np.float32)).reshape(-1, 1)
    got = ctx.l2_distance(x.reshape(-1, 1), np.zeros((4096, 1), np.float32), squared=False)
    want = np.sqrt((x * x).astype(np.float32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert a is not None


# ----------------------------------------------------------------------------------- Q3 seams
@pytest.mark.parametrize("d,sub,bits", [(128, 8, 8), (128, 4, 4), (256, 16, 8), (128, 32, 4), (64, 8, 1), (10, 2, 1),
                                        (3, 1, 1), (30, 6, 3), (48, 12, 5), (40, 20, 2), (21, 7, 2)])
def test_pq_quantize_and_distance(ctx, oracle, d, sub, bits):
    from muopdb_amd.index import ProductQuantizer
    from muopdb_amd import lib as L
    rng = np.random.default_rng(d * 31 + sub)
    m, K = d // sub, 1 << bits
    cb = rng.random(m * K * sub, dtype=np.float32)
    for metric in (L.METRIC_L2, L.METRIC_DOT):
        pq = ProductQuantizer(d, sub, bits, cb, metric)
        opq = oracle.ProductQuantizer(d, sub, bits, cb, metric)
        v = rng.random((200, d), dtype=np.float32)
        codes = pq.quantize(ctx, v)
        assert np.array_equal(codes, opq.quantize(v))
        rec = pq.original_vector(ctx, codes[:50])   # pq/mod.rs:184-200
        assert np.array_equal(rec.view(np.uint32), np.stack([opq.original_vector(c) for c in codes[:50]]).astype(np.float32).view(np.uint32))
        a = rng.integers(0, K, (300, m)).astype(np.uint8)
        b = rng.integers(0, K, (300, m)).astype(np.uint8)
        for impl, oimpl in ((L.IMPL_STREAMING_SIMD, oracle.PQ_STREAMING), (L.IMPL_SIMD, oracle.PQ_SIMD),
                            (L.IMPL_SCALAR, oracle.PQ_SCALAR)):
            if metric == L.METRIC_DOT and impl == L.IMPL_SCALAR:
                continue  # the Scalar arm is L2-only in the reference (pq/mod.rs:221-230)
            assert_scores(pq.distance(ctx, a, b, impl), opq.distance(a, b, oimpl))


def test_k5_k6_pq_quantize_kat(ctx):
    # reference KATs straight through the GPU path (ivf/writer.rs:511-676, pq/mod.rs:321-371)
    from muopdb_amd.index import ProductQuantizer
    pq = ProductQuantizer(3, 1, 1, [1.5, 4.5, 2.3, 5.3, 3.1, 6.1])
    assert pq.quantize(ctx, [[1, 2, 3], [4, 5, 6]]).tolist() == [[0, 0, 0], [1, 1, 1]]
    cb = []
    for s in range(5):
        for i in range(2):
            cb += [2 * s + i, 2 * s + i]
    pq = ProductQuantizer(10, 2, 1, cb)
    assert pq.quantize(ctx, [[1, 1, 3, 3, 5, 5, 7, 7, 9, 9]]).tolist() == [[1, 1, 1, 1, 1]]
    assert pq.original_vector(ctx, [[1, 1, 1, 1, 1]]).tolist() == [[1, 1, 3, 3, 5, 5, 7, 7, 9, 9]]  # pq/mod.rs:355-360


# ----------------------------------------------------------------------------------- E1
@pytest.mark.parametrize("values,universe", [
    ([5, 8, 8, 15, 32], 36), ([0, 1, 2, 3, 4], 5), ([10], 20), ([1000, 2000, 3000, 4000, 5000], 6000),
    ([1, 5, 10, 15, 20, 25, 30], 100), (list(range(1, 201)), 500), ([42], 100)])
def test_ef_decode_kat(ctx, values, universe):
    assert ctx.ef_decode(F.ef_encode(values, universe)).tolist() == values


def test_ef_decode_random(ctx, oracle):
    rng = np.random.default_rng(4)
    for n, bits in [(1, 3), (63, 10), (64, 20), (65, 31), (1000, 12), (5000, 32), (20000, 24), (300, 40), (7, 63)]:
        v = np.sort(rng.integers(0, 1 << bits, n, dtype=np.uint64))
        blob = F.ef_encode(v)
        assert np.array_equal(ctx.ef_decode(blob), v)
        assert np.array_equal(oracle.ef_decode(blob), v)
    assert ctx.ef_decode(F.ef_encode([])).tolist() == []
    dense = np.arange(100000, dtype=np.uint64)  # L = 0, all gaps 0/1
    assert np.array_equal(ctx.ef_decode(F.ef_encode(dense)), dense)


# ----------------------------------------------------------------------------------- flat (C1)
@pytest.mark.parametrize("n,d,b,k,metric", [(1000, 128, 1, 10, 0), (1000, 128, 7, 10, 1), (10000, 128, 16, 10, 0),
                                            (777, 30, 5, 3, 0), (300, 17, 4, 300, 1), (64, 4, 2, 100, 0),
                                            (5000, 768, 3, 20, 0), (1, 8, 1, 1, 0), (4097, 16, 9, 1, 1)])
def test_flat_topk(ctx, oracle, n, d, b, k, metric):
    from muopdb_amd.index import FlatIndex
    rng = np.random.default_rng(n + d)
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((b, d)).astype(np.float32)
    idx = FlatIndex(ctx, base, metric)
    ids, dist, counts = idx.search(q, k)
    oids, odist = oracle.flat_topk(metric, base, q, k)
    kk = min(k, n)
    assert counts.tolist() == [kk] * b
    assert np.array_equal(ids[:, :kk], oids[:, :kk])
    assert_scores(dist[:, :kk], odist[:, :kk])
    assert np.all(ids[:, kk:] == 0xFFFFFFFF)


def test_flat_c1_dataset_and_ties(ctx, oracle):
    from muopdb_amd.index import FlatIndex
    base = H.test_hdf5_like()  # BASELINE config C1: 10k x 128 (py/create_test_hdf5.py semantics)
    q = H.test_hdf5_like(n_per=10, seed=43)[:32]
    idx = FlatIndex(ctx, base)
    ids, dist, _ = idx.search(q, 10)
    oids, odist = oracle.flat_topk(0, base, q, 10)
    assert np.array_equal(ids, oids)
    assert_scores(dist, odist)
    d64 = np.sqrt(((q[:, None, :].astype(np.float64) - base[None]) ** 2).sum(-1))
    assert np.array_equal(np.sort(ids, 1), np.sort(np.argsort(d64, 1)[:, :10], 1))  # recall@10 = 1.0 vs f64
    # exact ties: duplicated rows must come out ordered by row id
    dup = np.repeat(base[:50], 4, axis=0)
    idx2 = FlatIndex(ctx, dup)
    ids2, dist2, _ = idx2.search(base[:5], 8)
    oids2, odist2 = oracle.flat_topk(0, dup, base[:5], 8)
    assert np.array_equal(ids2, oids2)
    assert ids2[0, :4].tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("n,d,b,k,metric", [(10000, 128, 1, 10, 0), (10000, 128, 3, 64, 0), (65536, 16, 4, 10, 1), (63, 48, 2, 64, 0), (1, 128, 1, 5, 0),
                                            (4100, 96, 1, 1, 1), (129, 112, 4, 33, 0)])
def test_flat_small_base_kernel(ctx, oracle, n, d, b, k, metric):
    """flat_small_scan_kernel (bases of <= 1024 tiles, batches <= 4, d = 16 .. 128 in whole chunks: BASELINE config 1's shape) — one wave
    per tile, the wave's keys ordered by a shuffle network, the lists merged by bound + rank — gives the oracle's rows bit for bit,
    incl. duplicated rows (ties ordered by row id), k above the number of rows, a short last tile, and the general kernel's rows."""
    from muopdb_amd.index import FlatIndex
    rng = np.random.default_rng(n * 7 + d)
    base = np.rint(rng.standard_normal((n, d)) * 3).astype(np.float32)      # coarse values: plenty of exact distance ties
    if n > 200:
        base[100:140] = base[7]                                              # 41 copies of one row
    q = np.rint(rng.standard_normal((b, d)) * 3).astype(np.float32)
    if n > 200:
        q[0] = base[7]
    idx = FlatIndex(ctx, base, metric)
    ids, dist, counts = idx.search(q, k)
    oids, odist = oracle.flat_topk(metric, base, q, k)
    kk = min(k, n)
    assert counts.tolist() == [kk] * b
    assert np.array_equal(ids[:, :kk], oids[:, :kk])
    assert np.array_equal(dist[:, :kk].view(np.uint32), odist[:, :kk].view(np.uint32))
    assert np.all(ids[:, kk:] == 0xFFFFFFFF)
    for form in (1, 2, 4):   # the general kernel; unordered keys + group bound; ONE launch with block tickets (k <= 16, else two launches)
        with ctx.option("MDB_FLAT_NO_SMALL", form):
            ids2, dist2, counts2 = idx.search(q, k)
            if form == 4:
                ids3, _, _ = idx.search(q, k)                              # the tickets were re-armed
                assert np.array_equal(ids2, ids3)
        assert np.array_equal(ids, ids2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32)) and np.array_equal(counts, counts2), form


def test_flat_batched_filter_rows_with_infinite_components(ctx, oracle):
    """scripts/stress_parity.py case 3900 (round 4): 70 000 x 30 rows, one in fifty with an infinite component, batch 33, top-200.
    With one bf16 product per pair the sample's bounds of such rows are inf - inf = NaN made by the bound's own arithmetic, and the
    image of a NEGATIVE NaN sorted below every distance: the k-th smallest bound fell, true neighbours were filtered out.  NaN
    bounds count as +inf now (sample_bound_kernel); rows must equal the exact kernels' and the oracle's."""
    from muopdb_amd.index import FlatIndex
    rng = np.random.default_rng(3900)
    n, d, b, k = 70_000, 30, 33, 200
    base = (rng.standard_normal((n, d)) * 10).astype(np.float32)
    rows = rng.integers(0, n, n // 50)
    base[rows, rng.integers(0, d, len(rows))] = np.inf
    q = (base[rng.integers(0, n, b)] + rng.normal(0, 1, (b, d))).astype(np.float32)
    q = np.where(np.isfinite(q), q, np.float32(0))
    idx = FlatIndex(ctx, base, 0)
    ids, dist, counts = idx.search(q, k)
    for opt in ("MDB_FLAT_NO_MFMA", "MDB_BF_EXACT_SAMPLE"):
        with ctx.option(opt, 1):
            eids, edist, ecounts = idx.search(q, k)
        assert np.array_equal(ids, eids) and np.array_equal(counts, ecounts), opt
        assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32)), opt
    with ctx.option("MDB_BF_X1", 0):   # the three-product filter: its lo halves exist only in a store LOADED under the option
        idx3 = FlatIndex(ctx, base, 0)
        tids, tdist, _ = idx3.search(q, k)
        xids, xdist, _ = idx.search(q, k)   # a store without lo halves keeps the one-product form whatever the option says now
    assert np.array_equal(ids, tids) and np.array_equal(dist.view(np.uint32), tdist.view(np.uint32))
    assert np.array_equal(ids, xids) and np.array_equal(dist.view(np.uint32), xdist.view(np.uint32))
    oids, odist = oracle.flat_topk(0, base, q[:4], k)
    assert np.array_equal(ids[:4], oids)
    assert_scores(dist[:4], odist)


@pytest.mark.parametrize("n,d,b,k,metric", [(120_000, 128, 600, 10, 0), (90_000, 120, 1100, 20, 0), (70_000, 128, 513, 10, 1)])
def test_flat_large_batch_block_filter_equals_exact(ctx, oracle, n, d, b, k, metric):
    """Batches of >= 512 queries over d <= 128 take the block-shared bf16 x 1 filter (flat_bf16x1_block_kernel: a bound pass and a
    filter pass over the whole base, a tile's B fragments through LDS, the A fragments in registers, a batch that is no multiple of
    the query groups): rows must be bit-identical to the exact kernels', to the per-wave x 1 filter's, to the sample-bounded block
    filter's, to the x 3 filter's and to the oracle's."""
    from muopdb_amd.index import FlatIndex
    rng = np.random.default_rng(n + b)
    base = H.sift_like(n, d, n_clusters=60, seed=n)
    q = (base[rng.integers(0, n, b)] + rng.normal(0, 12, (b, d))).astype(np.float32)
    idx = FlatIndex(ctx, base, metric)
    with ctx.option("MDB_BF_X1", 2):   # (dot stores take the x 3 filter by default)
        ids, dist, counts = idx.search(q, k)
        with ctx.option("MDB_BF_BLOCK_MIN_B", 1 << 30):
            wids, wdist, wcounts = idx.search(q, k)
        with ctx.option("MDB_BF_NO_FULL_BOUND", 1):   # the block filter behind the 1/4 sample's bound instead of the whole-base bound pass
            sids, sdist, scounts = idx.search(q, k)
    with ctx.option("MDB_BF_X1", 0):   # (the lo halves are built at load: a store of its own for the three-product filter)
        idx3 = FlatIndex(ctx, base, metric)
        tids, tdist, tcounts = idx3.search(q, k)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        eids, edist, ecounts = idx.search(q, k)
    for a_ids, a_dist, a_counts in ((wids, wdist, wcounts), (sids, sdist, scounts), (tids, tdist, tcounts), (eids, edist, ecounts)):
        assert np.array_equal(ids, a_ids) and np.array_equal(counts, a_counts)
        assert np.array_equal(dist.view(np.uint32), a_dist.view(np.uint32))
    sel = np.r_[0:6, b - 2:b]
    oids, odist = oracle.flat_topk(metric, base, q[sel], k)
    assert np.array_equal(ids[sel], oids)
    assert_scores(dist[sel], odist)


@pytest.mark.parametrize("n,d,b,k,metric,kind", [
    (100_000, 128, 40, 10, 0, "sift"), (100_000, 128, 100, 10, 1, "gauss"), (70_000, 30, 17, 5, 0, "gauss"),
    (66_000, 768, 33, 10, 0, "unit"), (80_000, 4, 64, 3, 0, "ramp"), (70_000, 16, 9, 10, 0, "same"),
    (130_000, 64, 8, 40, 1, "sift")])
def test_flat_batched_mfma_filter_equals_exact(ctx, oracle, n, d, b, k, metric, kind):
    """Batched flat path (sample top-k -> MFMA filter -> exact refine, mdb_flat_mfma.hip): ids and
    scores must be bit-identical to the exact kernel's and to the oracle's, including on data that
    defeats the sample bound (a ramp sorted by distance, all-identical rows -> overflow rescan)."""
    import os
    from muopdb_amd.index import FlatIndex
    rng = np.random.default_rng(n + d + b)
    if kind == "sift":
        base = H.sift_like(n, d, n_clusters=50, seed=n)
        q = (base[rng.integers(0, n, b)] + rng.normal(0, 10, (b, d))).astype(np.float32)
    elif kind == "gauss":
        base = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((b, d)).astype(np.float32)
    elif kind == "unit":
        base = rng.standard_normal((n, d)).astype(np.float32)
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        q = base[rng.integers(0, n, b)] + rng.normal(0, 0.01, (b, d)).astype(np.float32)
    elif kind == "ramp":  # spann/index.rs test data: row i = [i,i,i,i]
        base = np.repeat(np.arange(n, dtype=np.float32)[:, None], d, 1)
        q = np.repeat(rng.uniform(0, n, (b, 1)).astype(np.float32), d, 1) + np.float32(0.4)
    else:
        base = np.ones((n, d), np.float32)
        q = rng.standard_normal((b, d)).astype(np.float32)
    q = q.astype(np.float32)
    idx = FlatIndex(ctx, base, metric)
    ids, dist, counts = idx.search(q, k)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        eids, edist, ecounts = idx.search(q, k)
    assert np.array_equal(ids, eids) and np.array_equal(counts, ecounts)
    assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    # the refine by slices + merge launch (the default below MDB_REFINE_GROUP_MIN_B until round 6; still what a store without a row-major
    # copy takes) and the wave-per-slice form of it
    for opts in ({"MDB_REFINE_NO_GROUPS": 1}, {"MDB_REFINE_NO_GROUPS": 1, "MDB_REFINE_WAVE_MIN_B": 8}):
        import contextlib
        with contextlib.ExitStack() as st:
            for name, val in opts.items():
                st.enter_context(ctx.option(name, val))
            sids, sdist, scounts = idx.search(q, k)
        assert np.array_equal(ids, sids) and np.array_equal(counts, scounts) and np.array_equal(dist.view(np.uint32), sdist.view(np.uint32)), opts
    oids, odist = oracle.flat_topk(metric, base, q[:8], k)
    assert np.array_equal(ids[:8], oids)
    assert_scores(dist[:8], odist)


def test_flat_c1_golden_fixture(ctx):
    # the committed C1 fixture (tests/golden/c1_flat.npz) through the GPU path, one query per call (batch 1)
    import os
    from muopdb_amd.index import FlatIndex
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_flat.npz"))
    idx = FlatIndex(ctx, H.test_hdf5_like())
    for i, q in enumerate(g["queries"]):
        ids, dist, _ = idx.search(q[None, :], 10)
        assert np.array_equal(ids[0], g["ids"][i]) and np.array_equal(dist[0].view(np.uint32), g["dist"][i].view(np.uint32))


def test_flat_batched_path_nan_and_inf(ctx):
    # the MFMA filter never evaluates the exact distance of most rows: a NaN row must still raise (the reference
    # panics), an infinite row must behave as in the exact kernels
    import os
    from muopdb_amd.index import FlatIndex
    from muopdb_amd import lib as L
    rng = np.random.default_rng(9)
    base = rng.standard_normal((70000, 16)).astype(np.float32)
    q = rng.standard_normal((12, 16)).astype(np.float32)
    binf = base.copy(); binf[1234, 3] = np.inf; binf[40000] = -np.inf
    idx = FlatIndex(ctx, binf)
    ids, dist, _ = idx.search(q, 10)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        eids, edist, _ = idx.search(q, 10)
    assert np.array_equal(ids, eids) and np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    qinf = q.copy(); qinf[3, 5] = np.inf   # an infinite query: every distance is inf, the top-k is the first k rows
    idx2 = FlatIndex(ctx, base)
    ids, dist, _ = idx2.search(qinf, 10)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        eids, edist, _ = idx2.search(qinf, 10)
    assert np.array_equal(ids, eids) and np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    assert ids[3].tolist() == list(range(10))
    bnan = base.copy(); bnan[65000, 7] = np.nan
    with pytest.raises(L.MuopdbError) as e:
        FlatIndex(ctx, bnan).search(q, 10)
    assert e.value.status == 5


def test_flat_nan_is_an_error(ctx):
    from muopdb_amd.index import FlatIndex
    from muopdb_amd import lib as L
    base = np.zeros((10, 4), np.float32)
    base[3, 1] = np.nan
    idx = FlatIndex(ctx, base)
    with pytest.raises(L.MuopdbError) as e:
        idx.search(np.zeros((1, 4), np.float32), 2)
    assert e.value.status == 5  # MDB_ERR_NAN: the reference panics in NotNan::new(..).unwrap()
    ids, _, _ = FlatIndex(ctx, np.ones((10, 4), np.float32)).search(np.zeros((1, 4), np.float32), 2)  # context still usable
    assert ids.tolist() == [[0, 1]]


@pytest.mark.parametrize("lanes,d,metric", [(4, 16, 0), (8, 128, 0), (16, 768, 0), (16, 128, 1), (4, 100, 1), (8, 24, 0)])
def test_lane_conforming_distance(ctx, oracle, lanes, d, metric):
    # D3: LaneConformingDistanceCalculator (k-means' distance): bit-identical to the oracle
    rng = np.random.default_rng(lanes * d)
    a = (rng.standard_normal((50, d)) * 10).astype(np.float32)
    b = (rng.standard_normal((50, d)) * 10).astype(np.float32)
    got = ctx.lane_conforming_distance(a, b, lanes, metric)
    want = np.array([oracle.lane_conforming(metric, lanes, a[i], b[i]) for i in range(50)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n,d,L,mc,thr", [(20000, 32, 64, 3, 0.1), (70000, 16, 10, 1, 0.1), (5000, 128, 300, 8, 0.5), (3000, 7, 5, 5, 0.0)])
def test_ivf_assign_matches_builder_rule(ctx, oracle, n, d, L, mc, thr):
    # IvfBuilder::build_posting_lists' assignment (squared L2, threshold rule)
    from muopdb_amd.index import ivf_assign, posting_lists_from_assignment
    from muopdb_amd import lib as Lb
    v = H.sift_like(n, d, n_clusters=max(L // 2, 2), seed=n)
    cent = H.kmeans(v[:4000], L, iters=3, seed=1)
    ids, cnt = ivf_assign(ctx, cent, v, mc, thr)
    oids, ocnt = oracle.ivf_assign(cent, v, mc, thr)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids)
    pls = posting_lists_from_assignment(ids, cnt, cent.shape[0])
    assert sum(len(p) for p in pls) == int(cnt.sum()) and all(np.all(np.diff(p.astype(np.int64)) > 0) for p in pls if len(p) > 1)
    with pytest.raises(Lb.MuopdbError):
        ivf_assign(ctx, cent, v[:10], cent.shape[0] + 1, thr)


# ----------------------------------------------------------------------------------- IVF (I1-I3)
def _ivf_case(oracle, ctx, n, d, L, seed, quant=None, cpv=1, doc_base=100):
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    rng = np.random.default_rng(seed)
    v = H.sift_like(n, d, n_clusters=max(L // 2, 1), seed=seed)
    c = H.kmeans(v, L, iters=4, seed=seed)
    doc_ids = [doc_base + 3 * i + ((i % 7) << 70) for i in range(n)]
    if quant:
        sub, bits = quant
        cb = H.train_pq_codebook(v[: min(n, 2000)], sub, bits, iters=3)
        opq = oracle.ProductQuantizer(d, sub, bits, cb)
        index, vec, pls = H.build_ivf_files(v, doc_ids, c, quantize=opq.quantize, clusters_per_vector=cpv)
        oq = oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, sub, bits, cb)
        gq = ProductQuantizer(d, sub, bits, cb)
    else:
        index, vec, pls = H.build_ivf_files(v, doc_ids, c, clusters_per_vector=cpv)
        oq, gq = None, None
    o = oracle.BlockBasedIvf(index, vec, oq)
    g = BlockBasedIvf(ctx, index, vec, gq)
    q = v[rng.integers(0, n, 24)] + rng.normal(0, 2, (24, d)).astype(np.float32)
    return o, g, q.astype(np.float32), v, doc_ids


@pytest.mark.parametrize("n,d,L,P,k", [(3000, 32, 20, 5, 10), (3000, 128, 40, 40, 10), (500, 4, 10, 1, 3),
                                       (2000, 100, 7, 3, 50), (4000, 768, 16, 4, 10)])
def test_ivf_noq(ctx, oracle, n, d, L, P, k):
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, n, d, L, seed=n + d)
    assert g.num_clusters() == L and g.num_vectors() == n
    assert np.array_equal(g.find_nearest_centroids(q, P), o.find_nearest_centroids(q, P))
    assert_result_rows(g.search(q, k, P), o.search(q, k, num_probes=P), len(q))
    probes = o.find_nearest_centroids(q, P)[:, ::-1].copy()  # explicit centroid ids, any order
    assert_result_rows(g.search_with_centroids_and_remap(q, probes, k), o.search(q, k, probes=probes), len(q))


@pytest.mark.parametrize("n,d,sub,bits,L,P,k", [(4000, 128, 8, 8, 32, 8, 10), (3000, 64, 4, 4, 10, 10, 25),
                                                (2000, 30, 6, 3, 8, 3, 10), (2000, 48, 16, 6, 8, 8, 5),
                                                (1500, 21, 7, 2, 6, 6, 7)])
def test_ivf_pq(ctx, oracle, n, d, sub, bits, L, P, k):
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, n, d, L, seed=n + d + sub, quant=(sub, bits))
    assert_result_rows(g.search(q, k, P), o.search(q, k, num_probes=P), len(q))


@pytest.mark.parametrize("n,d,sub,bits,L,P,k", [(3000, 256, 32, 5, 12, 6, 10),      # code words per vector: 2
                                                (3000, 256, 8, 4, 12, 12, 64),     # 8 code words, k = 64 (warm start limit)
                                                (6000, 32, 8, 6, 700, 650, 10),    # > 512 probes: two chunks of the tile map
                                                (3000, 64, 4, 8, 9, 9, 100)])      # k > 64: no warm start
def test_ivf_pq_fast_scan_shapes(ctx, oracle, n, d, sub, bits, L, P, k):
    """ivf_scan_pq2_kernel (compile-time subvector width, flattened tile sequence, 3-stage pipeline):
    shapes around its template / chunk boundaries, with tombstones, against the oracle."""
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, n, d, L, seed=n + d + sub + bits, quant=(sub, bits))
    assert_result_rows(g.search(q, k, P), o.search(q, k, num_probes=P), len(q))
    first = g.search(q, k, P)
    dead = sorted({first.doc_ids(i)[0] for i in range(len(q)) if first.counts[i]})[:8]
    for doc in dead:
        assert g.invalidate(doc) and o.invalidate(doc)
    assert_result_rows(g.search(q, k, P), o.search(q, k, num_probes=P), len(q))


@pytest.mark.parametrize("n,d,sub,L,P,k", [(5000, 128, 8, 48, 16, 10),      # C3's shape in small: m = 16 (4 code words)
                                           (3000, 64, 4, 20, 1, 1),        # one probe, k = 1, m = 16
                                           (4000, 32, 8, 70, 64, 64),      # probes and k at the step's limits (one wave), m = 4 (1 code word)
                                           (3000, 128, 16, 12, 5, 25),     # m = 8 (2 code words)
                                           (2500, 256, 32, 9, 9, 7),       # widest subvectors, m = 8
                                           (3000, 128, 4, 33, 7, 10)])     # m = 32 (8 code words)
def test_ivf_pq_fused_step(ctx, oracle, n, d, sub, L, P, k):
    """ivf_pq_fused_kernel: coarse search + query quantization + bound table + scan + exact evaluation of the candidates + remap in
    ONE launch for small batches of an 8-bit L2 PQ index.  Rows, score bits and the scored-vector counter equal the oracle's
    and the unfused step's (MDB_PQ_NO_FUSED), with the library's own coarse search and with caller-given probes, through
    host and device buffers, as doc rows / point rows / a points block, with tombstones and per-call filters, and with a
    candidate list of 8 slots (every block overflows into its second, exact pass)."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from muopdb_amd import lib as L_
    from muopdb_amd.index import allow_bitmap
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, n, d, L, seed=n + d + sub + P, quant=(sub, 8))
    b = len(q)
    want = o.search(q, k, num_probes=P)
    probes = o.find_nearest_centroids(q, P)
    for cap in (2048, 8):
        with ctx.option("MDB_PQF_CAP", cap):
            got = g.search(q, k, P)                                              # coarse search inside the kernel
            st = ctx.stats()
            assert_result_rows(got, want, b)
            assert_result_rows(g.search_with_centroids_and_remap(q, probes, k), want, b)   # caller's probes
    with ctx.option("MDB_PQF_QUANT_IN_PREP", 1):                                 # the queries' codes from the prep launch instead of the per-query kernel
        assert_result_rows(g.search(q, k, P), want, b)
        assert_result_rows(g.search_with_centroids_and_remap(q, probes, k), want, b)
    with ctx.option("MDB_PQ_NO_FUSED", 1):
        ref = g.search(q, k, P)
        st_ref = ctx.stats()
    assert H.result_rows(ref, b) == H.result_rows(got, b)
    assert st["scored_vectors"] == st_ref["scored_vectors"] > 0
    assert np.array_equal(g.find_nearest_centroids(q, P), probes)
    # point rows and the points block of the sharded path come from the same kernel
    pi, ps, pc = g.search_points(q, k, P)
    with ctx.option("MDB_PQ_NO_FUSED", 1):
        ri, rs_, rc = g.search_points(q, k, P)
        blk_ref = g.search_shard(q, k, P)
    assert np.array_equal(pc, rc) and all(np.array_equal(pi[i, :pc[i]], ri[i, :rc[i]]) and
                                          np.array_equal(ps[i, :pc[i]].view(np.uint32), rs_[i, :rc[i]].view(np.uint32)) for i in range(b))
    assert np.array_equal(g.search_shard(q, k, P), blk_ref)
    # device-resident queries are read IN PLACE (row stride d, no staging copy): same rows
    dev = torch.device("cuda", torch.cuda.current_device())
    qd = torch.from_numpy(q).to(dev)
    ids = torch.zeros((b, k, 2), dtype=torch.int64, device=dev)
    sc = torch.zeros((b, k), dtype=torch.float32, device=dev)
    cn = torch.zeros(b, dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.mdb_ivf_search(g.h, C.c_void_p(qd.data_ptr()), C.c_size_t(b), None, C.c_size_t(P), C.c_size_t(k), C.c_int(L_.MEM_DEVICE),
                                     C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(cn.data_ptr())))
    ctx.sync()
    hi = ids.cpu().numpy().view(np.uint64)
    for i in range(b):
        c = int(want.counts[i])
        assert int(cn[i]) == c and [(int(hi[i, j, 1]) << 64) | int(hi[i, j, 0]) for j in range(c)] == want.doc_ids(i)
    # tombstones + per-call filters (shared and per query)
    dead = sorted({want.doc_ids(i)[0] for i in range(b) if want.counts[i]})[:6]
    for doc in dead:
        assert g.invalidate(doc) and o.invalidate(doc)
    assert_result_rows(g.search(q, k, P), o.search(q, k, num_probes=P), b)
    rng = np.random.default_rng(5)
    shared = allow_bitmap(np.sort(rng.choice(n, n // 2, replace=False)), n)
    per_q = np.stack([allow_bitmap(np.sort(rng.choice(n, n // 3, replace=False)), n) for _ in range(b)])
    for bm in (shared, per_q):
        with oracle.planner_filter(bm):
            fw = o.search(q, k, num_probes=P)
        assert_result_rows(g.search(q, k, P, planner=bm), fw, b)
        with ctx.option("MDB_PQF_CAP", 8):
            assert_result_rows(g.search(q, k, P, planner=bm), fw, b)
    # batch 1 and an odd batch
    assert_result_rows(g.search(q[:1], k, P), o.search(q[:1], k, num_probes=P), 1)
    assert_result_rows(g.search(q[3:10], k, P), o.search(q[3:10], k, num_probes=P), 7)


@pytest.mark.parametrize("case", ["wide_range", "ties", "overflow_to_inf", "zeros"])
def test_ivf_pq_bound_filter_adversarial(ctx, oracle, case):
    """The L2 bound filter of ivf_scan_pq2_kernel (bf16 lower bounds in front of the exact row sums) must never
    drop a true neighbour: codebooks with 60 decades of dynamic range, few distinct rows (thousands of exact
    ties with the admission threshold), squares that overflow to +inf, all-zero distances — rows, scores and the
    scored-vector counter equal the oracle's and the unfiltered kernel's (MDB_PQ_NO_FILTER)."""
    import os
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    rng = np.random.default_rng({"wide_range": 1, "ties": 2, "overflow_to_inf": 3, "zeros": 4}[case])
    n, d, sub, bits, L, P, k = 6000, 64, 8, 8, 6, 6, 10
    m, K = d // sub, 1 << bits
    if case == "wide_range":
        cb = rng.standard_normal((m, K, sub)).astype(np.float32)
        cb *= (np.float32(10) ** rng.integers(-15, 15, (m, K, 1)).astype(np.float32))
        cb[:, ::17, :] = 0
        cb[1, 5, :] = np.float32(1e-42)  # denormal
    elif case == "ties":
        base = rng.integers(0, 3, (m, 4, sub)).astype(np.float32)
        cb = base[:, rng.integers(0, 4, K), :]  # 4 distinct rows per subspace
    elif case == "overflow_to_inf":
        cb = rng.standard_normal((m, K, sub)).astype(np.float32)
        cb[:, :8, :] = np.float32(3e38) * np.sign(cb[:, :8, :])  # (a - b)^2 -> +inf against ordinary rows
    else:
        cb = np.zeros((m, K, sub), np.float32)
    cb = np.ascontiguousarray(cb.reshape(-1))
    codes = rng.integers(0, K, (n, m)).astype(np.uint8)
    if case == "overflow_to_inf":
        codes[rng.random((n, m)) < 0.9] = 40  # most subvectors ordinary, some vectors fully finite
    # vectors = their own reconstruction, so quantize() gives the codes back (nearest row; ties -> lowest index)
    v = cb.reshape(m, K, sub)[np.arange(m)[None, :], codes].reshape(n, d)
    if case == "overflow_to_inf":
        v = np.where(np.abs(v) > 1e38, np.sign(v) * np.float32(3e38), v).astype(np.float32)
    cent = v[rng.choice(n, L, replace=False)].astype(np.float32)
    if case in ("overflow_to_inf", "wide_range"):
        cent = rng.standard_normal((L, d)).astype(np.float32)
    doc_ids = [5 + 2 * i for i in range(n)]
    opq = oracle.ProductQuantizer(d, sub, bits, cb)
    with np.errstate(all="ignore"):
        stored = opq.quantize(v.astype(np.float32))
    index, vec, pls = H.build_ivf_files(np.nan_to_num(v, posinf=3e38, neginf=-3e38).clip(-1e18, 1e18) if case != "zeros" else v,
                                        doc_ids, cent, quantize=lambda x: stored)
    o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, sub, bits, cb))
    g = BlockBasedIvf(ctx, index, vec, ProductQuantizer(d, sub, bits, cb))
    q = v[rng.integers(0, n, 16)].astype(np.float32)
    probes = np.tile(np.arange(L, dtype=np.uint32), (len(q), 1))
    for kk in (k, 1, 64, 200):
        ores = o.search(q, kk, probes=probes)
        gres = g.search_with_centroids_and_remap(q, probes, kk)
        st = ctx.stats()
        assert_result_rows(gres, ores, len(q))
        with ctx.option("MDB_PQ_NO_FUSED", 1):                                    # the unfused step: table kernel with / without its bound filter
            gres1 = g.search_with_centroids_and_remap(q, probes, kk)
            st1 = ctx.stats()
            with ctx.option("MDB_PQ_NO_FILTER", 1):
                gres2 = g.search_with_centroids_and_remap(q, probes, kk)
            st2 = ctx.stats()
        assert_result_rows(gres1, ores, len(q))
        assert_result_rows(gres2, ores, len(q))
        assert st["scored_vectors"] == st1["scored_vectors"] == st2["scored_vectors"] == n * len(q)
        with ctx.option("MDB_PQ_SDC_MAX_MB", 0):   # the blocks build their row-sum words themselves instead of copying rows of the table
            assert_result_rows(g.search_with_centroids_and_remap(q, probes, kk), ores, len(q))
            assert ctx.stats()["scored_vectors"] == n * len(q)
        if kk <= 64:
            # the two-phase scan (bf16 lower / upper bounds, then exact distances of the candidates: batches >= 512);
            # then with candidate lists of 8 slots: every list overflows and the gated one-phase launch behind redoes the batch
            # (its scored count must not be added a second time)
            qb = np.concatenate([q, v[rng.integers(0, n, 560 - len(q))].astype(np.float32)])     # 560 queries: the path's own batch range
            pb = np.tile(np.arange(L, dtype=np.uint32), (len(qb), 1))
            oresb = o.search(qb, kk, probes=pb)
            for cap in (2048, 8):
                with ctx.option("MDB_PQ3_CAP", cap):
                    gres3 = g.search_with_centroids_and_remap(qb, pb, kk)
                st3 = ctx.stats()
                assert_result_rows(gres3, oresb, len(qb))
                assert st3["scored_vectors"] == n * len(qb), (cap, st3["scored_vectors"])
            with ctx.option("MDB_PQ_NO_TWO_PHASE", 1):                                                 # and the one-phase kernel on the same batch
                assert_result_rows(g.search_with_centroids_and_remap(qb, pb, kk), oresb, len(qb))
            with ctx.option("MDB_PQ_SDC_MAX_MB", 0):                                                   # the two-phase scan without the row-sum table
                assert_result_rows(g.search_with_centroids_and_remap(qb, pb, kk), oresb, len(qb))


def test_ivf_large_coarse_quantizer_block_filter_and_slices(ctx, oracle):
    """65 536 centroids of d = 128 (C5's coarse quantizer) and a batch of 600: the coarse search runs the block-shared bf16 x 1 filter
    (bound pass + filter pass over the whole quantizer, products handed to the group refine).  Probe ids must equal the exact
    kernels', the sample-bounded and the three-product filters', the oracle's — and the merge of the eight ranks' slice searches
    (mdb_ivf_coarse_keys over views of the filter operands: 256 tiles each, the same kernels)."""
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedIvf
    rng = np.random.default_rng(128)
    L, d, P, b = 65_536, 128, 48, 600
    cent = H.sift_like(L, d, n_clusters=300, seed=5)
    cent[4242] = cent[17]                                  # duplicate centroids: ties broken by index
    v = cent[:2000] + rng.standard_normal((2000, d)).astype(np.float32)
    pls = [np.array([i], np.uint64) if i < 2000 else np.zeros(0, np.uint64) for i in range(L)]
    index, vec = F.write_ivf_index(cent, list(range(2000)), pls), F.write_vector_file(v.astype(np.float32))
    g, o = BlockBasedIvf(ctx, index, vec), oracle.BlockBasedIvf(index, vec)
    q = (cent[rng.integers(0, L, b)] + rng.normal(0, 15, (b, d))).astype(np.float32)
    q[3] = cent[17]
    got = g.find_nearest_centroids(q, P)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        assert np.array_equal(got, g.find_nearest_centroids(q, P))
    for opt, val in (("MDB_BF_NO_FULL_BOUND", 1), ("MDB_BF_BLOCK_MIN_B", 1 << 30), ("MDB_BF_X1", 0), ("MDB_REFINE_NO_SECOND_BOUND", 1)):
        with ctx.option(opt, val):
            assert np.array_equal(got, g.find_nearest_centroids(q, P)), opt
    with ctx.option("MDB_BF_X1", 0):   # the three-product filter reads lo halves that only a load under the option builds
        g3 = BlockBasedIvf(ctx, index, vec)
        assert np.array_equal(got, g3.find_nearest_centroids(q, P))
        g3.close()
    assert np.array_equal(got[:6], o.find_nearest_centroids(q[:6], P))
    parts = [g.coarse_keys(q, P, first, 8192) for first in range(0, L, 8192)]
    assert np.array_equal(g.merge_coarse_keys(np.stack(parts, 1), P), got)


def test_ivf_large_coarse_quantizer_batched_path(ctx, oracle):
    """>= 64K centroids (C5 has 65 536 lists): batches of >= 8 queries find their probes through the batched flat path
    (sample bound + matrix-core filter + exact refine) — probe ids and final rows must equal the oracle's, for a batch
    that is not a multiple of 64, and equal the exact kernels' (small batch, MDB_FLAT_NO_MFMA)."""
    import os
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedIvf
    rng = np.random.default_rng(77)
    L, d, extra, P = 65_600, 24, 3_000, 24
    cent = (rng.standard_normal((L, d)) * 30).astype(np.float32)
    cent[100] = cent[7]                                   # duplicate centroids: ties broken by index
    own = np.concatenate([np.arange(L), rng.integers(0, L, extra)])
    v = (cent[own] + rng.standard_normal((L + extra, d))).astype(np.float32)
    order = np.argsort(own, kind="stable")
    bounds = np.searchsorted(own[order], np.arange(L + 1))
    pls = [order[bounds[i]:bounds[i + 1]].astype(np.uint64) for i in range(L)]
    doc_ids = [3 * i + 1 for i in range(L + extra)]
    index, vec = F.write_ivf_index(cent, doc_ids, pls), F.write_vector_file(v)
    g, o = BlockBasedIvf(ctx, index, vec), oracle.BlockBasedIvf(index, vec)
    q = (cent[rng.integers(0, L, 70)] + rng.standard_normal((70, d)) * 2).astype(np.float32)
    q[3] = cent[7]
    want = o.find_nearest_centroids(q, P)
    assert np.array_equal(g.find_nearest_centroids(q, P), want)          # batched path (70 queries)
    assert np.array_equal(g.find_nearest_centroids(q[:5], P), want[:5])  # exact kernels (batch < 8)
    with ctx.option("MDB_FLAT_NO_MFMA", 1):
        assert np.array_equal(g.find_nearest_centroids(q, P), want)
    assert_result_rows(g.search(q, 10, P), o.search(q, 10, num_probes=P), len(q))
    assert_result_rows(g.search(q[:9], 3, 1), o.search(q[:9], 3, num_probes=1), 9)
    # the coarse search SHARDED over 8 ranks' centroid slices (muopdb_amd.distributed.sharded_probes): aligned slices of a large
    # coarse quantizer take the batched path over a view of the filter operands (mdb_ivf_coarse_keys), the ragged tail the exact
    # kernels; the merged rows are find_nearest_centroids' probes
    parts = [g.coarse_keys(q, P, first, 8192) for first in range(0, 65536, 8192)] + [g.coarse_keys(q, P, 65536, L - 65536)]
    assert np.array_equal(g.merge_coarse_keys(np.stack(parts, 1), P), want)
    parts5 = [g.coarse_keys(q[:5], P, first, 8192) for first in range(0, 65536, 8192)] + [g.coarse_keys(q[:5], P, 65536, L - 65536)]
    assert np.array_equal(g.merge_coarse_keys(np.stack(parts5, 1), P), want[:5])
    # a per-call planner filter WITH the library's own (batched, matrix-core) coarse search: the staged host bitmaps must
    # survive the coarse search's scratch use (they once shared a slot).  Expected rows: the oracle under the same filter,
    # and the unfiltered rows with the dropped points removed wherever k of them remain.
    from muopdb_amd.index import allow_bitmap
    nv = L + extra
    keep = np.sort(rng.choice(nv, nv // 2, replace=False))
    shared = allow_bitmap(keep, nv)
    per_q = np.stack([allow_bitmap(np.sort(rng.choice(nv, nv // 3, replace=False)), nv) for _ in range(len(q))])
    for bm in (shared, per_q):
        with oracle.planner_filter(bm):
            want_f = o.search(q, 10, num_probes=P)
        assert_result_rows(g.search(q, 10, P, planner=bm), want_f, len(q))               # host bitmaps, internal coarse search
        pend = g.search_submit(q, 10, P, planner=bm)                                      # and through submit / wait
        assert_result_rows(pend.wait(), want_f, len(q))
    kept = set(int(3 * i + 1) for i in keep)
    wide = g.search(q, 40, P)
    filt = g.search(q, 10, P, planner=shared)
    for qi in range(len(q)):
        expect = [dd for dd in wide.doc_ids(qi) if dd in kept][:10]
        if len(expect) == 10:
            assert filt.doc_ids(qi) == expect


@pytest.mark.parametrize("d", [24, 128])
def test_coarse_refine_by_groups_ties_chunks_overflow(ctx, oracle, d):
    """Large batches over a coarse quantizer are refined one block per query (flat_refine_group_kernel: a 16-lane group per
    candidate, radix select + rank counting, final rows without a merge).  On a small batch (MDB_REFINE_GROUP_MIN_B; the default since round 6)
    over centroids with blocks of 1 200 / 3 000 / 9 000 IDENTICAL rows: ties beyond the survivor buffer (the count over all
    keys), lists longer than one chunk, and a list that overflows its capacity (the block scans the whole base) — probes must
    equal the oracle's, for d = 24 (general cascade) and d = 128 (the unrolled 16-lane pass), for k below and above 64."""
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedIvf
    rng = np.random.default_rng(d)
    L = 65_600
    cent = (rng.standard_normal((L, d)) * 30).astype(np.float32)
    sets = {}
    at = 500
    for cnt in (1200, 3000, 9000):
        rows = rng.choice(np.arange(at, at + 3 * cnt), cnt, replace=False)
        cent[rows] = cent[rows[0]]
        sets[cnt] = rows
        at += 3 * cnt
    pls = [np.array([i], dtype=np.uint64) for i in range(L)]
    index, vec = F.write_ivf_index(cent, list(range(1, L + 1)), pls), F.write_vector_file(cent)
    g, o = BlockBasedIvf(ctx, index, vec), oracle.BlockBasedIvf(index, vec)
    q = (cent[rng.integers(0, L, 40)] + rng.standard_normal((40, d)) * 2).astype(np.float32)
    q[1] = cent[sets[1200][0]]
    q[2] = cent[sets[3000][0]]
    q[3] = cent[sets[9000][0]]
    q[4] = cent[sets[3000][0]] + 0.5
    q[5] = cent[sets[9000][0]] - 0.25
    for P in (24, 200):
        want = o.find_nearest_centroids(q, P)
        with ctx.option("MDB_REFINE_GROUP_MIN_B", 8), ctx.option("MDB_REFINE_WAVE_MIN_B", 8), ctx.option("MDB_MF_COOLDOWN", 0):
            assert np.array_equal(g.find_nearest_centroids(q, P), want), P
            assert np.array_equal(g.find_nearest_centroids(q[:9], P), want[:9]), P
            with ctx.option("MDB_REFINE_NO_GROUPS", 1):
                assert np.array_equal(g.find_nearest_centroids(q, P), want), P
            with ctx.option("MDB_REFINE_NO_SECOND_BOUND", 1):     # the filter's candidates refined as they are
                assert np.array_equal(g.find_nearest_centroids(q, P), want), P
        assert np.array_equal(g.find_nearest_centroids(q, P), want), P


def test_merge_coarse_keys_rows_sorted_unsorted_duplicates_padding(ctx):
    """mdb_ivf_merge_coarse_keys on hand-made rows: ascending rows (ranks by binary search), rows that are NOT ascending (the
    block falls back to rank counting), the same key in several rows (both copies kept, ordered by row), rows padded with
    UINT64_MAX, one part, more keys than fit LDS (the selector merge), fewer valid keys than num_probes."""
    from muopdb_amd import formats as F
    from muopdb_amd.index import BlockBasedIvf
    rng = np.random.default_rng(5)
    cent = rng.standard_normal((8, 4)).astype(np.float32)
    g = BlockBasedIvf(ctx, F.write_ivf_index(cent, [1, 2, 3, 4, 5, 6, 7, 8], [np.array([i], dtype=np.uint64) for i in range(8)]),
                      F.write_vector_file(cent))
    MAXK = np.uint64(0xFFFFFFFFFFFFFFFF)

    def expect(keys, P):
        b = keys.shape[0]
        out = np.empty((b, P), np.uint32)
        for q in range(b):
            flat = np.sort(keys[q].reshape(-1), kind="stable")[:P]
            out[q] = [0xFFFFFFFF if kk == MAXK else int(kk) & 0xFFFFFFFF for kk in flat]
        return out

    for b, parts, P, mode in [(5, 8, 64, "sorted"), (5, 8, 64, "unsorted"), (3, 1, 17, "sorted"), (4, 3, 10, "dups"), (4, 5, 33, "padded"),
                              (2, 8, 1024, "sorted"), (3, 4, 7, "few")]:
        dist = rng.integers(0, 1 << 20, (b, parts, P)).astype(np.uint64)
        ids = rng.permutation(1 << 22)[: b * parts * P].reshape(b, parts, P).astype(np.uint64)
        keys = (dist << np.uint64(32)) | ids
        if mode == "dups":
            keys[:, 1, :] = keys[:, 0, :]          # a whole row twice: both copies are kept
        if mode != "unsorted":
            keys = np.sort(keys, axis=2)
        if mode == "padded":
            keys[:, :, P // 2:] = MAXK
            keys[:, 0, 3:] = MAXK
        if mode == "few":
            keys[:, :, 1:] = MAXK                  # 4 valid keys for 7 probes
        got = g.merge_coarse_keys(keys, P)
        assert np.array_equal(got, expect(keys, P)), (b, parts, P, mode)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_ivf_sharded_coarse_search_equals_unsharded(ctx, oracle, world):
    """Multi-GPU IVF shards the coarse quantizer too (muopdb_amd.distributed.sharded_probes): every rank scans its
    centroid range (mdb_ivf_coarse_keys), the (distance, id) rows are all-gathered and merged
    (mdb_ivf_merge_coarse_keys).  Simulated on one GPU: the merged probe ids equal find_nearest_centroids'."""
    from muopdb_amd.distributed import coarse_range
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, 6000, 20, 300, seed=91)
    P = 12
    want = o.find_nearest_centroids(q, P)
    assert np.array_equal(g.find_nearest_centroids(q, P), want)
    rows = []
    for r in range(world):
        first, count = coarse_range(300, r, world)
        rows.append(g.coarse_keys(q, P, first, count))
        ids = (rows[-1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        valid = rows[-1] != np.uint64(0xFFFFFFFFFFFFFFFF)
        assert np.all((ids[valid] >= first) & (ids[valid] < first + count)) and np.all(valid.sum(1) == min(P, count))
        assert np.all(rows[-1][:, :-1] <= rows[-1][:, 1:])
    keys = np.stack(rows, axis=1)                    # [b][world][P], what the all-gather + permute produces
    assert np.array_equal(g.merge_coarse_keys(keys, P), want)
    probes = g.merge_coarse_keys(keys, P)
    assert_result_rows(g.search_with_centroids_and_remap(q, probes, 10), o.search(q, 10, num_probes=P), len(q))
    if world == 1:  # the torch-side helper itself (no process group: one rank), device buffers
        import torch
        from muopdb_amd import distributed as D
        qd = torch.from_numpy(q).cuda()
        torch.cuda.synchronize()
        pr = D.sharded_probes(ctx, g, qd.data_ptr(), len(q), P, qd.device)
        ctx.stats()  # synchronises the context's stream
        assert np.array_equal(pr.cpu().numpy().astype(np.uint32), want)


def test_ivf_duplicates_tombstones_and_errors(ctx, oracle):
    from muopdb_amd import lib as L
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, 1200, 16, 6, seed=5, cpv=2)
    # a point in two probed lists is returned twice (index.rs:250-286 has no dedup)
    r, ro = g.search(q, 10, 6), o.search(q, 10, num_probes=6)
    assert_result_rows(r, ro, len(q))
    assert any(len(set(r.doc_ids(i))) < len(r.doc_ids(i)) for i in range(len(q)))
    # invalidate (index.rs:421-470)
    victims = ro.doc_ids(0)[:3]
    for d_ in victims:
        assert g.invalidate(d_) == o.invalidate(d_)
    assert not g.invalidate(victims[0]) and g.is_invalidated(victims[0]) and not g.invalidate(12345678901)
    assert not g.is_invalidated(doc_ids[-1])
    assert_result_rows(g.search(q, 10, 6), o.search(q, 10, num_probes=6), len(q))
    assert not set(victims) & set(g.search(q, 10, 6).doc_ids(0))
    # num_probes out of range: the reference panics (select_nth_unstable_by(num_probes - 1))
    for bad in (0, 7):
        with pytest.raises(L.MuopdbError) as e:
            g.find_nearest_centroids(q, bad)
        assert e.value.status == 8
    # k larger than the candidates: short rows, padded
    r = g.search(q[:2], 2000, 1)
    ro = o.search(q[:2], 2000, num_probes=1)
    assert_result_rows(r, ro, 2)
    assert int(r.counts[0]) < 2000 and r.doc_lo[0, int(r.counts[0])] == 0xFFFFFFFFFFFFFFFF
    # k = 0
    assert g.search(q[:2], 0, 2).counts.tolist() == [0, 0]


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_ivf_f32_list_lengths_around_the_unit_boundaries(ctx, oracle, metric):
    """f32 posting lists are stored in 16-slot units — whole 64-wide tiles, then a 16 / 32 / 48-wide tail (mdb_ivf.hip,
    gather_f32_units_kernel): lists of every length around those boundaries (0, 1, 15..17, 31..33, 47..49, 63..65, ... 200), a
    point in several lists, d not a multiple of 4, low-entropy vectors (score ties), tombstones and per-call filters; every probe
    subset == the oracle's search_with_centroids_and_remap (ivf/block_based/index.rs:250-332)."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf, NoQuantizer, allow_bitmap
    rng = np.random.default_rng(2024)
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 111, 112, 113, 127, 128, 129, 143, 200, 0, 64]
    n, d = 1500, 22
    v = rng.integers(0, 3, (n, d)).astype(np.float32)
    doc_ids = [int(x) + ((i % 3) << 90) for i, x in enumerate(rng.permutation(n) + 1000)]
    pls = [np.sort(rng.choice(n, ln, replace=False)).astype(np.uint64) for ln in lens]        # lists overlap: duplicates across probes
    cent = rng.normal(0, 1, (len(lens), d)).astype(np.float32)
    index, vec = F.write_ivf_index(cent, doc_ids, pls), F.write_vector_file(v)
    m = oracle.METRIC_L2 if metric == "l2" else oracle.METRIC_DOT
    o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_NONE, m))
    gq = NoQuantizer(d, L.METRIC_L2 if metric == "l2" else L.METRIC_DOT)
    g = BlockBasedIvf(ctx, index, vec, gq)
    q = (v[rng.integers(0, n, 9)] + rng.integers(0, 2, (9, d))).astype(np.float32)
    allp = np.tile(np.arange(len(lens), dtype=np.uint32), (len(q), 1))
    cases = [allp] + [allp[:, j:j + 1].copy() for j in range(len(lens))] + [allp[:, ::-1][:, 3:17].copy(), allp[:, 10:14].copy()]
    for k in (1, 10, 64):
        for probes in cases if k == 10 else cases[:1] + cases[-2:]:
            assert_result_rows(g.search_with_centroids_and_remap(q, probes, k), o.search(q, k, probes=probes), len(q))
    dead = o.search(q, 5, probes=allp).doc_ids(0)[:3] + [doc_ids[int(pls[5][0])], doc_ids[int(pls[27][199])]]
    for doc in dead:
        assert g.invalidate(doc) == o.invalidate(doc)
    bm = allow_bitmap(np.sort(rng.choice(n, n // 2, replace=False)), n)
    for probes in (allp, cases[-2]):
        assert_result_rows(g.search_with_centroids_and_remap(q, probes, 10), o.search(q, 10, probes=probes), len(q))
        with oracle.planner_filter(bm):
            fw = o.search(q, 10, probes=probes)
        assert_result_rows(g.search_with_centroids_and_remap(q, probes, 10, planner=bm), fw, len(q))
    assert g.num_vectors() == n and g.num_clusters() == len(lens)
    # the granularity is a load-time option (MDB_IVF_LIST_PAD_UNITS: 1 = 16 slots, 2, 4 = whole 64-slot tiles): same rows, tombstones included
    for pad in (2, 4):
        with ctx.option("MDB_IVF_LIST_PAD_UNITS", pad):
            gp = BlockBasedIvf(ctx, index, vec, gq)
        for doc in dead:
            gp.invalidate(doc)
        for probes in (allp, cases[-2]):
            assert_result_rows(gp.search_with_centroids_and_remap(q, probes, 10), o.search(q, 10, probes=probes), len(q))
        gp.close()
    g.close()


def test_ivf_kat_k7_container(ctx, oracle):
    # hand-assembled container of combined_file.rs:172-300 searched through the GPU path
    from muopdb_amd.index import BlockBasedIvf
    centroids = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)
    vecs = np.array([[1, 2, 3, 4], [1, 2, 3, 5], [5, 6, 7, 8], [5, 6, 7, 9]], np.float32)
    index = F.write_ivf_index(centroids, [100, 200, 300, 400], [np.array([0, 1], np.uint64), np.array([2, 3], np.uint64)])
    g = BlockBasedIvf(ctx, index, F.write_vector_file(vecs))
    r = g.search([[5, 6, 7, 8.4]], 3, 1)
    assert r.doc_ids(0) == [300, 400] and r.counts[0] == 2
    assert g.find_nearest_centroids([[1, 2, 3, 4]], 2).tolist() == [[0, 1]]
    # empty posting list + offset load (multi-user style): blob placed at a 16-aligned offset
    index2 = F.write_ivf_index(centroids, [1, 2], [np.array([], np.uint64), np.array([0, 1], np.uint64)])
    pad_i, pad_v = b"\xAA" * 32, b"\xBB" * 24
    g2 = BlockBasedIvf(ctx, pad_i + index2, pad_v + F.write_vector_file(vecs[:2]), index_offset=32, vector_offset=24)
    assert g2.search([[1, 2, 3, 4]], 5, 2).doc_ids(0) == [1, 2]
    assert g2.search([[1, 2, 3, 4]], 5, 1).counts[0] == 0  # nearest list is the empty one


def test_ivf_sharded_union_equals_unsharded(ctx, oracle):
    # list sharding: the union of per-shard top-k == the single-GPU top-k
    from muopdb_amd.index import BlockBasedIvf
    o, g, q, v, doc_ids = _ivf_case(oracle, ctx, 3000, 32, 12, seed=77)
    index, vec, _ = H.build_ivf_files(v, doc_ids, H.kmeans(v, 12, iters=4, seed=77))
    full_ids, full_sc, full_cn = g.search_points(q, 10, 5)
    shards = [BlockBasedIvf(ctx, index, vec, shard_rank=r, shard_world=3) for r in range(3)]
    probes = g.find_nearest_centroids(q, 5)
    for qi in range(len(q)):
        rows = []
        for s in shards:
            ids, sc, cn = s.search_points(q[qi:qi + 1], 10, 5, probes=probes[qi:qi + 1])
            rows += list(zip(sc[0, :cn[0]].tolist(), ids[0, :cn[0]].tolist()))
        rows.sort()
        assert [r[1] for r in rows[:10]] == full_ids[qi, :full_cn[qi]].tolist()
    # ownership of a single index is size-balanced (longest list first to the least loaded rank): the shards hold what
    # muopdb_amd.distributed.balanced_owners says, every vector exactly once, loads within one list of each other
    from muopdb_amd.distributed import balanced_owners
    sizes = [len(pl) for pl in H.build_ivf_files(v, doc_ids, H.kmeans(v, 12, iters=4, seed=77))[2]]
    owner = balanced_owners(sizes, 3)
    loads = [s.num_resident_vectors() for s in shards]
    assert loads == [sum(sz for sz, o in zip(sizes, owner) if o == r) for r in range(3)]
    assert sum(loads) == g.num_resident_vectors() == len(v) and max(loads) - min(loads) <= max(sizes)


def test_handles_outlive_context_close(oracle):
    """mdb_device_close only drops the caller's reference: a handle freed (or even searched) after
    the context was closed must not crash (Python GC frees in arbitrary order)."""
    from muopdb_amd import lib
    from muopdb_amd.index import FlatIndex
    c = lib.Context(0)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((500, 16)).astype(np.float32)
    q = rng.standard_normal((3, 16)).astype(np.float32)
    f = FlatIndex(c, base)
    want = f.search(q, 5)[0]
    h, c.h = c.h, None
    c.lib.mdb_device_close(h)          # context "closed" while f is alive
    c.h = h                            # still usable through the handle's reference
    assert np.array_equal(f.search(q, 5)[0], want)
    c.h = None
    f.close()                          # last reference: destroys the context


def test_golden_index_fixtures_through_the_hip_path(ctx):
    """The committed fixtures (tests/golden/*.npz: index files + queries + expected doc ids / score bits / traversal
    counters, scripts/make_index_fixtures.py) through the C ABI — no oracle in the loop."""
    import os
    from muopdb_amd.index import BlockBasedHnsw, BlockBasedIvf, MultiSpannIndex, ProductQuantizer, SearchParams
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "hnsw_small.npz"))
    h = BlockBasedHnsw(ctx, g["index"].tobytes(), g["vectors"].tobytes(), int(g["dimension"]))
    for ef in (40, 600):   # 40: beam kernel; 600 >= 500 points: closure kernel
        r = h.ann_search(g["queries"], int(g["k"]), ef)
        st = ctx.stats()
        assert H.result_rows(r, len(g["queries"])) == H.golden_rows(g, "ef%d_" % ef)
        assert [st["distance_evals"], st["expanded_nodes"]] == [int(x) for x in g["ef%d_counters" % ef]]
    g = np.load(os.path.join(gold, "ivfpq_small.npz"))
    ivf = BlockBasedIvf(ctx, g["index"].tobytes(), g["vectors"].tobytes(), ProductQuantizer(32, 8, 5, g["codebook"]))
    q, k, P = g["queries"], int(g["k"]), int(g["nprobe"])
    assert np.array_equal(ivf.find_nearest_centroids(q, P), g["probes"])
    assert H.result_rows(ivf.search(q, k, P), len(q)) == H.golden_rows(g, "a_")
    for lo, hi in zip(g["dead_lo"], g["dead_hi"]):
        assert ivf.invalidate((int(hi) << 64) | int(lo))
    assert H.result_rows(ivf.search(q, k, P), len(q)) == H.golden_rows(g, "b_")
    g = np.load(os.path.join(gold, "mspann_small.npz"))
    ms = MultiSpannIndex(ctx, g["user_table"].tobytes(), 8, g["hnsw_index"].tobytes(), g["hnsw_vectors"].tobytes(),
                         g["ivf_index"].tobytes(), g["ivf_vectors"].tobytes())
    p = SearchParams(5, 50).with_num_explored_centroids(4).with_centroid_distance_ratio(0.3)
    r = ms.search_for_user([int(u) for u in g["user_ids"]], g["queries"], p)
    assert [bool(f) for f in r.found] == [bool(f) for f in g["found"]]
    docs, bits = H.golden_rows(g)
    for i, f in enumerate(g["found"]):
        if f:
            assert r.doc_ids(i) == docs[i]
            assert [int(x) for x in np.asarray(r.scores[i, :len(docs[i])], np.float32).view(np.uint32)] == bits[i]
