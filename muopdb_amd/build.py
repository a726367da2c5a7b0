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




def reassigned_ids(posting_lists, num_vectors):
    """IvfBuilder::get_reassigned_ids (rs/index/src/ivf/builder.rs:596-676): new point ids that make every posting list a run of
    consecutive ids as far as vectors shared by several lists (max_clusters_per_vector > 1: the "stopping points") allow.
    Lists that share a vector are drained up to it, smallest shared vector first (ties: list contents), then the shared vector itself
    is numbered; whatever is left is numbered list by list.  Returns int64 [num_vectors], -1 for vectors in no list.
    Without shared vectors (max_clusters_per_vector == 1) the answer is the list order itself: vectorised."""
    import heapq
    lists = [np.asarray(pl, np.int64).reshape(-1) for pl in posting_lists]
    out = np.full(num_vectors, -1, np.int64)
    if not lists:
        return out
    allv = np.concatenate(lists) if len(lists) else np.zeros(0, np.int64)
    occ = np.bincount(allv, minlength=num_vectors) if allv.size else np.zeros(num_vectors, np.int64)
    cur = 0
    if allv.size and occ.max() > 1:
        shared = occ > 1
        heap = []
        for pl in lists:
            sp = np.unique(pl[shared[pl]])
            if sp.size:
                heap.append((int(sp[0]), tuple(int(v) for v in pl), tuple(int(v) for v in sp)))
        heapq.heapify(heap)
        while heap:
            stop = heap[0][0]
            work = []
            while heap and heap[0][0] == stop:     # every list whose next shared vector is `stop`, in heap order
                work.append(heapq.heappop(heap))
            for _, pl, sps in work:
                for i, v in enumerate(pl):
                    if v == stop:
                        if len(sps) > 1:
                            heapq.heappush(heap, (sps[1], pl[i + 1:], sps[1:]))
                        break
                    if out[v] >= 0:
                        raise ValueError("Vectors that come before a stopping point should not be reassigned")
                    out[v] = cur
                    cur += 1
            out[stop] = cur
            cur += 1
    # the rest, list by list (:663-674): first occurrence wins
    if allv.size:
        first = np.ones(allv.size, bool)
        order = np.argsort(allv, kind="stable")
        first[order[1:]] = allv[order[1:]] != allv[order[:-1]]
        todo = allv[first & (out[allv] < 0)]
        out[todo] = cur + np.arange(todo.size)
    return out


def reindex(posting_lists, doc_ids, vectors):
    """IvfBuilder::reindex (ivf/builder.rs:682-761): renumber the points with reassigned_ids and move doc ids and vectors to their
    new places.  Returns (posting lists in new ids — kept in list order, so NOT necessarily ascending when lists share vectors,
    doc ids [n'], vectors [n'], mapping u32 [n] old -> new: the bytes of the segment's `reassigned_mappings` file, 0xFFFFFFFF for a
    vector that is in no list).  n' = number of vectors that are in some list."""
    vectors = np.asarray(vectors)
    n = vectors.shape[0]
    ids = reassigned_ids(posting_lists, n)
    new_lists = [ids[np.asarray(pl, np.int64)].astype(np.uint64) for pl in posting_lists]
    valid = np.flatnonzero(ids >= 0)
    rev = np.empty(valid.size, np.int64)
    rev[ids[valid]] = valid
    docs = doc_ids if isinstance(doc_ids, np.ndarray) else np.asarray(list(doc_ids), dtype=object)
    return new_lists, docs[rev], vectors[rev], ids.astype(np.uint32)   # (-1 as u32 = 0xFFFFFFFF, `*x as u32` at :684)


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


# ------------------------------------------------------------------------------------------ HNSW construction
def select_neighbors(ctx, x, cand_ids, cand_dist, max_neighbors, metric=L.METRIC_L2):
    """HnswBuilder::select_neighbors_heuristic (hnsw/builder.rs:339-375) for many candidate lists at once
    (mdb_hnsw_select_neighbors).  cand_ids / cand_dist [rows][width] in pop order (distance ascending, larger id first among
    equals), UINT32_MAX padded.  Returns (ids [rows][M], dist [rows][M], counts [rows]) as numpy arrays."""
    p, n, d, mem, keep = _rows(x)
    ci = np.ascontiguousarray(cand_ids, np.uint32)
    cd = np.ascontiguousarray(cand_dist, np.float32)
    rows, width = ci.shape
    ids = np.empty((rows, max_neighbors), np.uint32)
    dist = np.empty((rows, max_neighbors), np.float32)
    cnt = np.zeros(rows, np.uint32)
    if mem == L.MEM_DEVICE:
        import torch
        torch.cuda.synchronize()
    ctx.check(ctx.lib.mdb_hnsw_select_neighbors(ctx.h, p, C.c_size_t(n), C.c_size_t(d), C.c_int(metric), C.c_int(mem), L.ptr(ci, C.c_uint32),
                                                L.ptr(cd, C.c_float), C.c_size_t(rows), C.c_size_t(width), C.c_size_t(max_neighbors),
                                                L.ptr(ids, C.c_uint32), L.ptr(dist, C.c_float), L.ptr(cnt, C.c_uint32)))
    return ids, dist, cnt


def _pop_order(ids, dist):
    """rows re-ordered into the heap's pop order: distance ascending, LARGER id first among equal distances; padding last"""
    pad = ids == 0xFFFFFFFF
    key_id = np.where(pad, -1, ids.astype(np.int64))
    o1 = np.argsort(-key_id, axis=1, kind="stable")
    ids, dist, pad = np.take_along_axis(ids, o1, 1), np.take_along_axis(dist, o1, 1), np.take_along_axis(pad, o1, 1)
    o2 = np.argsort(np.where(pad, np.inf, dist), axis=1, kind="stable")
    return np.take_along_axis(ids, o2, 1), np.take_along_axis(dist, o2, 1)


def insert_hnsw(ctx, x, max_neighbors=32, max_layers=8, ef_construction=100, seed=1, batch_frac=0.125, exact_below=2048, log=None):
    """HNSW construction with the reference's algorithm, batched for the GPU (HnswBuilder::insert, hnsw/builder.rs:221-305).

    Per new point the reference (i) draws a level (get_random_layer :332-337), (ii) finds ef_construction candidates on
    every layer <= its level with search_layer, (iii) keeps <= max_neighbors of them with select_neighbors_heuristic, links
    both ways and (iv) re-runs the heuristic on every neighbour whose edge list overflowed (:256-300).  Here a BATCH of
    points (batch_frac of the points already inserted) goes through (ii)-(iv) together: the layer-0 searches run through the
    library's traversal kernel on the graph built so far (written in the reference's on-disk format and loaded like any
    index), the selections and the trims through mdb_hnsw_select_neighbors.  Points of one batch do not see each other —
    the usual relaxation of parallel HNSW builds — so the graph is quality-parity with the sequential reference build
    (tests compare recall against the oracle's restatement of `insert`), not edge-identical.  Upper layers (1/M of the points)
    and the first `exact_below` points take their candidates from an exact scan (FlatIndex) instead of a graph search.
    The reference stores the heuristic's NEGATED distance on a kept edge (builder.rs:366-369, a sign slip that its later
    trims inherit); edges here carry the true distance.

    x: [n][d] numpy or torch CUDA tensor.  Returns (layers, levels): layers in formats.write_hnsw_index's CSR form
    (layer 0 first; points None for layer 0), point ids = row numbers."""
    from .index import BlockBasedHnsw, FlatIndex
    n, d = x.shape
    M, efc = int(max_neighbors), int(ef_construction)
    rng = np.random.default_rng(seed)
    u = 1.0 - rng.random(n)
    levels = np.minimum(np.floor(-np.log(u) / np.log(M)), max_layers).astype(np.int64)
    xh = x.cpu().numpy() if _is_torch(x) else L.f32(x)
    adj = np.full((n, M), 0xFFFFFFFF, np.uint32)       # layer 0 edge lists
    adjd = np.full((n, M), np.inf, np.float32)
    cnt = np.zeros(n, np.int64)
    upper = {}                                          # layer -> {point: [(id, dist), ...]}
    top = int(levels[0])
    for l in range(1, top + 1):
        upper.setdefault(l, {})[0] = []
    first_top = 0

    def csr_layers(cur):
        m = np.arange(M)[None, :] < cnt[:cur, None]
        indptr = np.zeros(cur + 1, np.uint64)
        indptr[1:] = np.cumsum(cnt[:cur])
        out = [(None, indptr, adj[:cur][m])]
        for l in range(1, top + 1):
            pts = sorted(upper.get(l, {}).keys())
            if l == top and first_top in pts:           # the reader's entry point = FIRST point of the top layer
                pts.remove(first_top)
                pts.insert(0, first_top)
            ip = np.zeros(len(pts) + 1, np.uint64)
            ed = []
            for i, pnt in enumerate(pts):
                ed += [e for e, _ in upper[l][pnt]]
                ip[i + 1] = len(ed)
            out.append((np.asarray(pts, np.uint32), ip, np.asarray(ed, np.uint32)))
        return out

    def link(layer_get, layer_put, p, sel_ids, sel_d):
        """forward + reverse edges of point p on one UPPER layer, trims through the heuristic (python: few points)"""
        layer_put(p, list(zip(sel_ids, sel_d)))
        over = []
        for e, de in zip(sel_ids, sel_d):
            lst = layer_get(e)
            lst.append((p, de))
            if len(lst) > M:
                over.append(e)
        return over

    cur = 1
    while cur < n:
        B = int(min(n - cur, max(1, cur * batch_frac)))
        new = np.arange(cur, cur + B)
        q = xh[cur:cur + B]
        # ---- (ii) layer-0 candidates of the whole batch
        k0 = min(efc, cur)
        if cur <= exact_below:
            fi = FlatIndex(ctx, xh[:cur])
            ids0, d0, _ = fi.search(q, k0)
            fi.close()
            ids0 = ids0.astype(np.uint32)
        else:
            idx = F.write_hnsw_index(csr_layers(cur), np.arange(cur, dtype=np.uint64), d)
            g = BlockBasedHnsw(ctx, idx, F.write_vector_file(xh[:cur]), d)
            res = g.ann_search(q, k0, efc)
            g.close()
            ids0 = np.where(np.arange(k0)[None, :] < res.counts[:, None], res.doc_lo[:, :k0], 0xFFFFFFFF).astype(np.uint32)
            d0 = np.where(ids0 == 0xFFFFFFFF, np.inf, res.scores[:, :k0]).astype(np.float32)
        ci, cd = _pop_order(ids0, d0.astype(np.float32))
        sel, seld, selc = select_neighbors(ctx, x if _is_torch(x) else xh, ci, cd, M)
        # ---- (iii) layer 0: forward edges, then reverse edges grouped by target
        adj[new, :] = sel
        adjd[new, :] = seld
        cnt[new] = selc
        valid = np.arange(M)[None, :] < selc[:, None]
        tgt = sel[valid].astype(np.int64)
        src = np.repeat(new, selc.astype(np.int64))
        rd = seld[valid]
        order = np.lexsort((rd, tgt))                   # per target: nearest new neighbours first
        tgt, src, rd = tgt[order], src[order], rd[order]
        ue, start, ncount = np.unique(tgt, return_index=True, return_counts=True)
        rank = np.arange(tgt.size) - np.repeat(start, ncount)
        fits = cnt[ue] + ncount <= M
        # targets with room: append in place
        fm = np.repeat(fits, ncount)
        adj[tgt[fm], cnt[tgt[fm]] + rank[fm]] = src[fm].astype(np.uint32)
        adjd[tgt[fm], cnt[tgt[fm]] + rank[fm]] = rd[fm]
        cnt[ue[fits]] += ncount[fits]
        # ---- (iv) overflowing targets: old edges + new ones through the heuristic again
        oe = ue[~fits]
        if oe.size:
            R = int(min(ncount[~fits].max(), 2 * M))    # the nearest <= 2M newcomers of a target compete with its M old edges
            W = M + R
            li = np.full((oe.size, W), 0xFFFFFFFF, np.uint32)
            ld = np.full((oe.size, W), np.inf, np.float32)
            li[:, :M] = adj[oe]
            ld[:, :M] = adjd[oe]
            om = (~fm) & (rank < R)
            rowof = np.searchsorted(oe, tgt[om])
            li[rowof, M + rank[om]] = src[om].astype(np.uint32)
            ld[rowof, M + rank[om]] = rd[om]
            li, ld = _pop_order(li, ld)
            t_ids, t_d, t_c = select_neighbors(ctx, x if _is_torch(x) else xh, li, ld, M)
            adj[oe] = t_ids
            adjd[oe] = t_d
            cnt[oe] = t_c
        # ---- upper layers of the batch (1/M of the points): exact candidates among the layer's members, python bookkeeping
        for l in range(1, int(levels[new].max()) + 1):
            pts_new = new[levels[new] >= l]
            members = np.array(sorted(upper.get(l, {}).keys()), np.int64)
            lay = upper.setdefault(l, {})
            if members.size and pts_new.size:
                fi = FlatIndex(ctx, xh[members])
                kk = min(efc, members.size)
                ui, ud, _ = fi.search(xh[pts_new], kk)
                fi.close()
                ci, cd = _pop_order(members[ui.astype(np.int64)].astype(np.uint32), ud.astype(np.float32))
                s_i, s_d, s_c = select_neighbors(ctx, x if _is_torch(x) else xh, ci, cd, M)
            over = []
            for r, pnt in enumerate(pts_new.tolist()):
                if members.size:
                    c_ = int(s_c[r])
                    over += link(lambda e: lay[e], lambda p_, v: lay.__setitem__(p_, v), pnt, s_i[r, :c_].tolist(), s_d[r, :c_].tolist())
                else:
                    lay[pnt] = []
            over = sorted(set(over))
            if over:
                W = max(len(lay[e]) for e in over)
                li = np.full((len(over), W), 0xFFFFFFFF, np.uint32)
                ld = np.full((len(over), W), np.inf, np.float32)
                for r, e in enumerate(over):
                    li[r, :len(lay[e])] = [a for a, _ in lay[e]]
                    ld[r, :len(lay[e])] = [b for _, b in lay[e]]
                li, ld = _pop_order(li, ld)
                t_i, t_d, t_c = select_neighbors(ctx, x if _is_torch(x) else xh, li, ld, M)
                for r, e in enumerate(over):
                    lay[e] = list(zip(t_i[r, :int(t_c[r])].tolist(), t_d[r, :int(t_c[r])].tolist()))
        hi = int(levels[new].max())
        if hi > top:                                    # a new top layer: its first point becomes the entry point (:301-303)
            first_top = int(new[levels[new] == hi][0])
            for l in range(top + 1, hi + 1):
                upper.setdefault(l, {})
                for pnt in new[levels[new] >= l].tolist():
                    upper[l].setdefault(pnt, [])
            top = hi
        cur += B
        if log:
            log("insert_hnsw: %d / %d points" % (cur, n))
    return csr_layers(n), levels


def hnsw_files_by_insertion(ctx, x, doc_ids=None, **kw):
    """(index_bytes, vector_bytes) in the reference's HNSW formats, graph built by insert_hnsw."""
    layers, _ = insert_hnsw(ctx, x, **kw)
    n, d = x.shape
    if doc_ids is None:
        doc_ids = np.arange(n, dtype=np.uint64)
    xh = x.cpu().numpy() if _is_torch(x) else L.f32(x)
    return F.write_hnsw_index(layers, doc_ids, d), F.write_vector_file(xh)
