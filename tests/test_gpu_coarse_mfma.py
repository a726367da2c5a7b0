"""find_nearest_centroids (rs/index/src/ivf/block_based/index.rs:147-163) of a mid-sized coarse quantizer through the matrix-core filter
of mdb_ivf_coarse.hip.h: bf16 products as a NECESSARY test, exact reference-association distances for what passes.  Bar: the probe
ids equal the oracle's find_nearest_centroids row for row (nearest first, the index breaks ties), on the exact path
(MDB_IVF_COARSE_MFMA=0) and on the filtered one, alone and inside the fused IVF-PQ step, on random and adversarial centroid sets."""
import numpy as np
import pytest

from tests import helpers as H
from tests.test_gpu_parity import assert_result_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


def _index(oracle, ctx, cent, n_vec=3000, seed=0, pq=None):
    """an IVF index over the given centroids: a few thousand vectors drawn around them (most lists short or empty: the coarse search
    does not care), NoQuantizer or PQ (subdim, bits)"""
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    rng = np.random.default_rng(seed)
    L_, d = cent.shape
    fin = np.where(np.isfinite(cent).all(axis=1))[0]
    pick = rng.choice(fin, n_vec)
    with np.errstate(all="ignore"):
        v = (cent[pick] * (1.0 + rng.normal(0, 1e-3, (n_vec, d))) + rng.normal(0, 1.0, (n_vec, d))).astype(np.float32)
    doc_ids = list(range(10, 10 + n_vec))
    if pq:
        sub, bits = pq
        cb = H.train_pq_codebook(v[:1500], sub, bits, iters=2)
        opq = oracle.ProductQuantizer(d, sub, bits, cb)
        index, vec, _ = H.build_ivf_files(v, doc_ids, cent, quantize=opq.quantize)
        return (oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, sub, bits, cb)),
                BlockBasedIvf(ctx, index, vec, ProductQuantizer(d, sub, bits, cb)), v)
    index, vec, _ = H.build_ivf_files(v, doc_ids, cent)
    return oracle.BlockBasedIvf(index, vec, None), BlockBasedIvf(ctx, index, vec, None), v


def _check_probes(ctx, o, g, q, probes=(1, 8, 16, 17, 40, 64)):
    for P in probes:
        want = o.find_nearest_centroids(q, P)
        got = g.find_nearest_centroids(q, P)
        assert np.array_equal(got, want), "P=%d: first differing query %d" % (P, int(np.argmax((got != want).any(axis=1))))
        with ctx.option("MDB_IVF_COARSE_MFMA", 0):
            assert np.array_equal(g.find_nearest_centroids(q, P), want)


@pytest.mark.parametrize("L_,d,b", [(4096, 128, 256),    # C3's coarse quantizer and batch
                                    (1024, 64, 37),      # smallest quantizer served, a ragged batch (one full group + 5 queries)
                                    (1500, 128, 100),    # L not a multiple of 32: padded rows in the last tile
                                    (2048, 96, 64), (1100, 192, 33), (1024, 256, 32),
                                    (16384, 64, 48)])    # largest quantizer the fused step takes: 16 splits of 32 tiles
def test_coarse_mfma_probes_equal_oracle(ctx, oracle, L_, d, b):
    rng = np.random.default_rng(L_ + d + b)
    cent = H.sift_like(L_, d, n_clusters=max(8, L_ // 64), seed=L_ + d)
    o, g, v = _index(oracle, ctx, cent, seed=L_)
    q = (cent[rng.integers(0, L_, b)] + rng.normal(0, 6.0, (b, d))).astype(np.float32)
    _check_probes(ctx, o, g, q)
    # batches below the filter's range keep the exact kernels; same ids
    assert np.array_equal(g.find_nearest_centroids(q[:5], 16), o.find_nearest_centroids(q[:5], 16))
    # far queries (every centroid about equally far: a weak bound, many candidates)
    far = (rng.normal(0, 1.0, (b, d)) * 400.0 + 100.0).astype(np.float32)
    _check_probes(ctx, o, g, far, probes=(1, 16, 64))


@pytest.mark.parametrize("case", ["duplicates", "tight_far_from_origin", "zeros", "huge_norms", "one_infinite", "mixed_scales", "integer_grid"])
def test_coarse_mfma_adversarial_centroids(ctx, oracle, case):
    """sets on which the bf16 products decide nothing (or lie): the exact evaluation behind the filter must still return the oracle's
    ids — through overflowing candidate segments (the exact scan of all centroids inside the query's block) where it has to"""
    rng = np.random.default_rng(7)
    L_, d, b = 2048, 128, 64
    if case == "duplicates":           # every centroid eight times: ties decided by the index, hundreds of candidates at the bound
        base = H.sift_like(L_ // 8, d, n_clusters=16, seed=3)
        cent = np.repeat(base, 8, axis=0)[rng.permutation(L_)]
    elif case == "tight_far_from_origin":   # spread 1e-2 around 1e4: below bf16's resolution even after centring on the mean of two clusters
        cent = (np.where(rng.random((L_, 1)) < 0.5, 1e4, -1e4) + rng.normal(0, 1e-2, (L_, d))).astype(np.float32)
    elif case == "zeros":
        cent = np.zeros((L_, d), np.float32)
    elif case == "huge_norms":         # 1e18: squares overflow the guard of the bound (1e30), not f32
        cent = (rng.normal(0, 1.0, (L_, d)) * 1e18).astype(np.float32)
    elif case == "one_infinite":       # one centroid with an infinite coordinate: its distance is +inf (never NaN against finite queries)
        cent = H.sift_like(L_, d, n_clusters=32, seed=5)
        cent[77, 5] = np.inf
    elif case == "mixed_scales":       # a few centroids 1e6 away set XNMAX, the rest sit within 1 of each other
        cent = rng.normal(0, 1.0, (L_, d)).astype(np.float32)
        cent[::200] *= 1e6
    else:                              # small integers: many exactly equal distances
        cent = rng.integers(0, 3, (L_, d)).astype(np.float32)
    o, g, v = _index(oracle, ctx, cent, seed=11)
    q = (cent[rng.integers(0, L_, b)] + rng.normal(0, 0.5, (b, d)).astype(np.float32)).astype(np.float32)
    q = np.where(np.isfinite(q), q, 0.0).astype(np.float32)
    if case == "zeros":
        q[::2] = 0.0
    _check_probes(ctx, o, g, q, probes=(1, 16, 64))


def test_coarse_mfma_nan_is_reported(ctx, oracle):
    """NotNan::new(..).unwrap() panics in the reference: a NaN distance anywhere in the coarse search is MDB_ERR_NAN, with or without
    the filter (a NaN product admits its pair, so the exact evaluation meets it)"""
    from muopdb_amd.lib import MuopdbError
    rng = np.random.default_rng(2)
    cent = H.sift_like(1024, 64, n_clusters=16, seed=9)
    o, g, v = _index(oracle, ctx, cent, seed=4)
    q = (cent[:40] + 1.0).astype(np.float32)
    q[3, 7] = np.nan
    for flag in (1, 0):
        with ctx.option("MDB_IVF_COARSE_MFMA", flag):
            with pytest.raises(MuopdbError) as e:
                g.find_nearest_centroids(q, 8)
            assert e.value.status == 5
    bad = cent.copy()
    bad[500, 1] = np.nan
    o2, g2, _ = _index(oracle, ctx, bad, seed=4)
    for flag in (1, 0):
        with ctx.option("MDB_IVF_COARSE_MFMA", flag):
            with pytest.raises(MuopdbError) as e:
                g2.find_nearest_centroids(cent[:40] + 1.0, 8)
            assert e.value.status == 5


@pytest.mark.parametrize("L_,d,sub,b,P,k", [(4096, 128, 8, 256, 16, 10),   # C3: nlist 4096, m = 16, batch 256, nprobe 16
                                            (1024, 64, 4, 70, 64, 20), (1536, 128, 16, 32, 1, 5)])
def test_fused_step_with_coarse_mfma(ctx, oracle, L_, d, sub, b, P, k):
    """the fused IVF-PQ step with its coarse search on the matrix cores (ivf_coarse_mfma_kernel + ivf_pq_fused_kernel<.., 2>): rows and
    score bits equal the oracle's and the [B][L] path's, the scored-vector counter equals the exact path's"""
    rng = np.random.default_rng(L_ + b)
    cent = H.sift_like(L_, d, n_clusters=max(8, L_ // 64), seed=L_ + 1)
    o, g, v = _index(oracle, ctx, cent, n_vec=6000, seed=L_, pq=(sub, 8))
    q = (v[rng.integers(0, len(v), b)] + rng.normal(0, 2.0, (b, d))).astype(np.float32)
    want = o.search(q, k, num_probes=P)
    got = g.search(q, k, P)
    st = ctx.stats()
    assert_result_rows(got, want, b)
    with ctx.option("MDB_IVF_COARSE_MFMA", 0):
        ref = g.search(q, k, P)
        st_ref = ctx.stats()
    assert H.result_rows(ref, b) == H.result_rows(got, b)
    assert st["scored_vectors"] == st_ref["scored_vectors"] > 0
    # a candidate capacity of 8 slots in the PQ scan: the second (exact) pass of every block, behind the filtered coarse search
    with ctx.option("MDB_PQF_CAP", 8):
        assert_result_rows(g.search(q, k, P), want, b)
    # the unfused step takes its probes from the same two launches (IvfSet::coarse)
    with ctx.option("MDB_PQ_NO_FUSED", 1):
        assert_result_rows(g.search(q, k, P), want, b)
