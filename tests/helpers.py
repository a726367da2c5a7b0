"""Test-side index synthesis (numpy): deterministic k-means, PQ codebooks, IVF / SPANN /
multi-user segment assembly in the reference's on-disk formats (muopdb_amd.formats)."""
import numpy as np

from muopdb_amd import formats as F


def kmeans(x, k, iters=10, seed=0):
    """Plain seeded Lloyd (the reference's k-means is nondeterministic — centroids are inputs)."""
    rng = np.random.default_rng(seed)
    x = np.asarray(x, np.float32)
    n = x.shape[0]
    k = min(k, n)
    c = x[rng.choice(n, k, replace=False)].copy()
    for _ in range(iters):
        d = ((x[:, None, :].astype(np.float64) - c[None, :, :]) ** 2).sum(-1) if n * k * x.shape[1] < 5e7 else \
            (x ** 2).sum(1)[:, None] - 2 * x @ c.T + (c ** 2).sum(1)[None, :]
        a = d.argmin(1)
        for j in range(k):
            m = a == j
            if m.any():
                c[j] = x[m].mean(0)
    return c.astype(np.float32)


def assign(x, c):
    x = np.asarray(x, np.float32)
    d = (x ** 2).sum(1)[:, None] - 2 * x @ c.T + (c ** 2).sum(1)[None, :]
    return d.argmin(1)


def train_pq_codebook(x, subdim, num_bits, iters=8, seed=0):
    x = np.asarray(x, np.float32)
    d = x.shape[1]
    m, K = d // subdim, 1 << num_bits
    cb = np.zeros((m, K, subdim), np.float32)
    for s in range(m):
        sub = x[:, s * subdim:(s + 1) * subdim]
        c = kmeans(sub, K, iters, seed + s)
        if c.shape[0] < K:  # fewer distinct rows than K: pad by repeating
            c = np.concatenate([c, np.repeat(c[-1:], K - c.shape[0], 0)])
        cb[s] = c
    return cb.reshape(-1)


def build_ivf_files(vectors, doc_ids, centroids, quantize=None, clusters_per_vector=1):
    """Returns (index_bytes, vectors_bytes, posting_lists).  quantize: f32[n,d] -> u8[n,m] or None."""
    vectors = np.asarray(vectors, np.float32)
    centroids = np.asarray(centroids, np.float32)
    L = centroids.shape[0]
    d = (vectors ** 2).sum(1)[:, None] - 2 * vectors @ centroids.T + (centroids ** 2).sum(1)[None, :]
    order = np.argsort(d, axis=1, kind="stable")[:, :clusters_per_vector]
    pls = [[] for _ in range(L)]
    for pid in range(vectors.shape[0]):
        for c in order[pid]:
            pls[int(c)].append(pid)
    pls = [np.asarray(p, np.uint64) for p in pls]
    if quantize is not None:
        stored = quantize(vectors)
        qd = stored.shape[1]
    else:
        stored, qd = vectors, vectors.shape[1]
    index = F.write_ivf_index(centroids, doc_ids, pls, quantized_dimension=qd)
    return index, F.write_vector_file(stored), pls


def build_hnsw_files(orc, vectors, doc_ids, max_neighbors=10, max_layers=2, ef_construction=100, seed=1, metric=0):
    """HNSW over `vectors` with the oracle's reference-style builder -> (index_bytes, vector_bytes)."""
    vectors = np.asarray(vectors, np.float32)
    b = orc.HnswBuilder(vectors.shape[1], max_neighbors, max_layers, ef_construction, metric, seed)
    b.insert(vectors)
    layers = b.layers()
    eps = b.entry_points()
    # make the builder's entry point the FIRST point of the top layer (the reader's rule)
    if len(layers) > 1:
        top = layers[-1]
        ordered = {eps[0]: top[eps[0]]}
        for p, e in top.items():
            if p != eps[0]:
                ordered[p] = e
        layers[-1] = ordered
    index = F.write_hnsw_index(layers, doc_ids, vectors.shape[1])
    return index, F.write_vector_file(vectors)


def build_spann_files(orc, vectors, doc_ids, num_clusters, quantize=None, seed=0, centroids=None, **hnsw_kw):
    """One user's SPANN: HNSW over IVF centroids (doc id = centroid index) + IVF posting lists."""
    vectors = np.asarray(vectors, np.float32)
    if centroids is None:
        centroids = kmeans(vectors, num_clusters, seed=seed)
    iidx, ivec, pls = build_ivf_files(vectors, doc_ids, centroids, quantize)
    hidx, hvec = build_hnsw_files(orc, centroids, list(range(centroids.shape[0])), **hnsw_kw)
    return dict(hnsw_index=hidx, hnsw_vectors=hvec, ivf_index=iidx, ivf_vectors=ivec,
                ivf_raw_vectors=F.write_vector_file(vectors)), centroids, pls


def sift_like(n, d=128, n_clusters=64, sigma=20.0, seed=1):
    """SIFT-like synthetic rows: Gaussian clusters, clipped to [0,218], rounded."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 218, (n_clusters, d))
    a = rng.integers(0, n_clusters, n)
    x = centers[a] + rng.normal(0, sigma, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(np.float32)


def test_hdf5_like(n_per=1000, n_clusters=10, d=128, seed=42):
    """py/create_test_hdf5.py:8-37 semantics: 10 clusters x 1000 points, centre i*100, N(0,5^2), shuffled."""
    np.random.seed(seed)
    parts = []
    for i in range(n_clusters):
        center = np.ones(d) * i * 100
        parts.append(center + np.random.normal(0, 5, (n_per, d)))
    data = np.vstack(parts)
    np.random.shuffle(data)
    return data.astype(np.float32)


def golden_rows(g, prefix=""):
    """(doc ids per query, score bits per query) of a tests/golden/*.npz result block written by scripts/make_index_fixtures.py"""
    lo, hi, sb, cnt = g[prefix + "lo"], g[prefix + "hi"], g[prefix + "score_bits"], g[prefix + "counts"]
    docs = [[(int(hi[i, j]) << 64) | int(lo[i, j]) for j in range(int(cnt[i]))] for i in range(len(cnt))]
    bits = [[int(x) for x in sb[i, :int(cnt[i])]] for i in range(len(cnt))]
    return docs, bits


def result_rows(res, b):
    return ([res.doc_ids(i) for i in range(b)],
            [[int(x) for x in np.asarray(res.scores[i, :int(res.counts[i])], np.float32).view(np.uint32)] for i in range(b)])
