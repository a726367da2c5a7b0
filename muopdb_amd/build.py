"""Index construction through the native library (the callers / data producers either side of the
hot path).  Every distance-shaped step runs in libmuopdb_hip.so:

* kmeans_fit / kmeans      — KMeansBuilder::fit (rs/utils/src/kmeans_builder/kmeans_builder.rs:116-360) = mdb_kmeans_fit:
                             Lloyd with the size penalty and the empty-cluster repair, bit-identical to the reference's
                             run from the same initial points (the reference draws them with thread_rng, here a seeded
                             numpy generator does);
* train_pq_codebook        — ProductQuantizerBuilder's role (rs/quantization/src/pq/pq_builder.rs:43-102: one k-means
                             per subvector; the reference uses the third-party `kmeans` crate's minibatch variant, so this
                             is QUALITY parity by construction) on mdb_kmeans_fit;
* assign_nearest           — IvfBuilder::build_posting_lists' per-vector step (ivf/builder.rs:267-326) = mdb_ivf_assign;
* ivf_build_centroids      — IvfBuilder::build_centroids (ivf/builder.rs:460-541): sample -> k-means -> assign -> split the
                             longest posting list until none exceeds max_posting_list_size;
* insert_hnsw              — HnswBuilder::insert (hnsw/builder.rs:221-305) batched over the library's traversal kernels.

Inputs are numpy arrays (host) or torch CUDA tensors (device; torch is plumbing here: data_ptr() and slicing only).
The products are the reference's on-disk formats (muopdb_amd.formats), which the hot path then loads.
"""
import ctypes as C

import numpy as np

from . import formats as F
from . import lib as L


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _rows(x):
    """(pointer, n, d, mem, keepalive) of a [n][d] f32 matrix: numpy -> host, torch CUDA tensor -> device"""
    if _is_torch(x):
        import torch
        t = x.contiguous()
        if t.dtype != torch.float32:
            t = t.float()
        if t.is_cuda:
            return C.c_void_p(t.data_ptr()), t.shape[0], t.shape[1], L.MEM_DEVICE, t
        x = t.numpy()
    a = L.f32(x)
    a = a.reshape(-1, a.shape[-1])
    return L.ptr(a, C.c_float), a.shape[0], a.shape[1], L.MEM_HOST, a


def _take(x, idx):
    if _is_torch(x):
        import torch
        return x[torch.as_tensor(idx, device=x.device)]
    return np.asarray(x)[idx]


# ------------------------------------------------------------------------------------------ k-means
def kmeans_fit(ctx, x, num_clusters, max_iter=10, tolerance=0.0, init_ids=None, seed=0):
    """KMeansBuilder::fit: (centroids [k][d], assignments [n], error, iterations); k = min(num_clusters, n).
    Outputs live where x lives (numpy / torch CUDA).  init_ids = `cluster_init_values` (default: k distinct random points,
    like init_random_points' choose_multiple, from a seeded generator)."""
    p, n, d, mem, keep = _rows(x)
    k = min(int(num_clusters), n)
    if init_ids is None:
        init_ids = np.random.default_rng(seed).choice(n, size=k, replace=False)
    init = np.ascontiguousarray(init_ids, np.uint64)
    err, it = C.c_float(), C.c_uint32()
    if mem == L.MEM_DEVICE:
        import torch
        cent = torch.empty((k, d), dtype=torch.float32, device=keep.device)
        lab = torch.empty(n, dtype=torch.int32, device=keep.device)
        cp, lp = C.c_void_p(cent.data_ptr()), C.c_void_p(lab.data_ptr())
        torch.cuda.synchronize()  # x may still be in flight on torch's stream (the context runs on its own)
    else:
        cent = np.empty((k, d), np.float32)
        lab = np.empty(n, np.uint32)
        cp, lp = L.ptr(cent, C.c_float), L.ptr(lab, C.c_uint32)
    ctx.check(ctx.lib.mdb_kmeans_fit(ctx.h, p, C.c_size_t(n), C.c_size_t(d), C.c_size_t(num_clusters), C.c_size_t(max_iter),
                                     C.c_float(tolerance), L.ptr(init, C.c_uint64), C.c_size_t(init.size), C.c_int(mem), cp, lp,
                                     C.byref(err), C.byref(it)))
    return cent, lab, float(err.value), int(it.value)


def kmeans(ctx, x, k, iters=10, seed=0, sample=None, tolerance=0.0):
    """centroids [k][d] of a (sampled) Lloyd run — IvfBuilder's use of KMeansBuilder (ivf/builder.rs:470-500)."""
    n = x.shape[0]
    if sample is not None and sample < n:
        x = _take(x, np.sort(np.random.default_rng(seed + 7919).choice(n, size=sample, replace=False)))
    return kmeans_fit(ctx, x, k, max_iter=iters, tolerance=tolerance, seed=seed)[0]


def train_pq_codebook(ctx, x, subdim, num_bits, iters=8, seed=0, sample=100_000):
    """[m][2^num_bits][subdim] f32 codebook, flattened (the `codebook` file of pq/mod.rs:101-126): one k-means per subvector."""
    n, d = x.shape
    m, K = d // subdim, 1 << num_bits
    if sample is not None and sample < n:
        x = _take(x, np.sort(np.random.default_rng(seed + 104729).choice(n, size=sample, replace=False)))
    cb = np.empty((m, K, subdim), np.float32)
    for s in range(m):
        sub = x[:, s * subdim:(s + 1) * subdim]
        c = kmeans_fit(ctx, sub.contiguous() if _is_torch(sub) else np.ascontiguousarray(sub), K, max_iter=iters, seed=seed + s)[0]
        c = c.cpu().numpy() if _is_torch(c) else c
        if c.shape[0] < K:  # fewer points than codes: pad by repeating the last row
            c = np.concatenate([c, np.repeat(c[-1:], K - c.shape[0], 0)])
        cb[s] = c
    return cb.reshape(-1)


# ------------------------------------------------------------------------------------------ IVF
def assign_nearest(ctx, x, centroids, max_clusters_per_vector=1, distance_threshold=0.1):
    """nearest centroid(s) of every row by the reference's squared-L2 cascade (mdb_ivf_assign).  max_clusters_per_vector == 1:
    labels [n]; otherwise (ids [n][mc] UINT32_MAX padded, counts [n]).  Output lives where x lives."""
    p, n, d, mem, keep = _rows(x)
    if mem == L.MEM_DEVICE:
        import torch
        c = centroids if _is_torch(centroids) else torch.from_numpy(L.f32(centroids)).to(keep.device)
        c = c.contiguous().float()
        ids = torch.empty((n, max_clusters_per_vector), dtype=torch.int32, device=keep.device)
        cnt = torch.empty(n, dtype=torch.int32, device=keep.device)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.mdb_ivf_assign(ctx.h, C.c_void_p(c.data_ptr()), C.c_size_t(c.shape[0]), p, C.c_size_t(n), C.c_size_t(d),
                                         C.c_size_t(max_clusters_per_vector), C.c_float(distance_threshold), C.c_int(mem),
                                         C.c_void_p(ids.data_ptr()), C.c_void_p(cnt.data_ptr())))
        ctx.sync()
        return ids[:, 0].to(torch.int64) if max_clusters_per_vector == 1 else (ids, cnt)
    c = L.f32(centroids.cpu().numpy() if _is_torch(centroids) else centroids)
    ids = np.empty((n, max_clusters_per_vector), np.uint32)
    cnt = np.empty(n, np.uint32)
    ctx.check(ctx.lib.mdb_ivf_assign(ctx.h, L.ptr(c, C.c_float), C.c_size_t(c.shape[0]), p, C.c_size_t(n), C.c_size_t(d),
                                     C.c_size_t(max_clusters_per_vector), C.c_float(distance_threshold), C.c_int(mem),
                                     L.ptr(ids, C.c_uint32), L.ptr(cnt, C.c_uint32)))
    return ids[:, 0].astype(np.int64) if max_clusters_per_vector == 1 else (ids, cnt)


def posting_lists_from_assignment(assign, num_lists):
    """list of sorted u64 point-id arrays (IvfBuilder::build_posting_lists role)."""
    a = assign.cpu().numpy() if _is_torch(assign) else np.asarray(assign)
    order = np.argsort(a, kind="stable")
    bounds = np.searchsorted(a[order], np.arange(num_lists + 1))
    return [order[bounds[i]:bounds[i + 1]].astype(np.uint64) for i in range(num_lists)]




def ivf_build_centroids(ctx, x, num_clusters, max_posting_list_size, num_data_points_for_clustering=20_000, max_iteration=10,
                        tolerance=0.0, seed=0):
    """IvfBuilder::build_centroids (rs/index/src/ivf/builder.rs:460-541): first pass = k-means over a sample with
    compute_actual_num_clusters clusters and an assignment of every point; then the LONGEST posting list is re-clustered
    (cluster_docs :419-444: ceil(len / max) clusters from a sample of max(10 * clusters, num_data_points_for_clustering) of
    its points) until no list exceeds max_posting_list_size.  Returns (centroids [L][d] numpy, posting lists as sorted u64
    arrays); empty lists are dropped like the reference does (:529-534).  The reference's sampling is thread_rng, so the
    outcome is quality-parity; each k-means run inside is the bit-exact mdb_kmeans_fit."""
    import heapq
    n, d = x.shape
    rng = np.random.default_rng(seed)

    def ceil_div(a, b):
        return (a + b - 1) // b

    def cluster(point_ids, k, n_sample, s):
        pick = point_ids if len(point_ids) <= n_sample else np.sort(rng.choice(point_ids, size=n_sample, replace=False))
        cent = kmeans_fit(ctx, _take(x, pick), k, max_iter=max_iteration, tolerance=tolerance, seed=s)[0]
        lab = assign_nearest(ctx, _take(x, point_ids), cent)
        lab = lab.cpu().numpy() if _is_torch(lab) else lab
        cent = cent.cpu().numpy() if _is_torch(cent) else cent
        order = np.argsort(lab, kind="stable")
        bounds = np.searchsorted(lab[order], np.arange(cent.shape[0] + 1))
        return [(cent[i], point_ids[order[bounds[i]:bounds[i + 1]]]) for i in range(cent.shape[0])]

    per = ceil_div(n, num_clusters)
    k0 = ceil_div(n, min(per, max_posting_list_size))  # compute_actual_num_clusters :446-458
    heap, tick = [], 0
    for cen, pl in cluster(np.arange(n), k0, max(k0, num_data_points_for_clustering), seed):
        heapq.heappush(heap, (-len(pl), tick, cen, pl)); tick += 1
    while heap and -heap[0][0] > max_posting_list_size:
        _, _, _, pl = heapq.heappop(heap)
        k = ceil_div(len(pl), max_posting_list_size)
        for cen, sub in cluster(pl, k, max(k * 10, num_data_points_for_clustering), seed + tick):
            heapq.heappush(heap, (-len(sub), tick, cen, sub)); tick += 1
    kept = [(cen, pl) for _, _, cen, pl in sorted(heap, key=lambda t: t[1]) if len(pl)]
    return np.stack([c for c, _ in kept]).astype(np.float32), [np.sort(pl).astype(np.uint64) for _, pl in kept]
