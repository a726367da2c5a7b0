"""Workload SYNTHESIS for tests and bench.py — not the product's build path (that is muopdb_amd.build, native).

No dataset can be downloaded here, so the BASELINE workloads are synthesised: data generators (SiftLike, EmbedLike and
round 1's isotropic ones), float64 exact k-NN ground truth for recall, a FAST graph generator for million-point HNSW
benchmark indexes (exact k-NN graph + the reference's neighbour-selection heuristic: HNSW-FORMAT graphs, not the
reference's insertion algorithm — that is muopdb_amd.build.insert_hnsw), and the C5 shard builder.  PyTorch is used
freely here (GEMM-shaped bulk work on synthetic data); nothing in this module is on the search path.
"""
import math

import numpy as np
import torch

from . import formats as F


# ------------------------------------------------------------------------------------------ data
class SiftLike:
    """BASELINE.md C2/C3 synthetic "SIFT-1M" (no dataset can be downloaded): integer-valued f32 rows in [0, 218] with the
    two properties of real descriptors that decide what an index can do with them — a LOW INTRINSIC DIMENSION and
    per-subvector structure a product quantizer can code.  Every 8-float block s is a `latent_per_block`-dimensional
    latent mapped through a fixed non-negative 8 x r frame U_s; the d/8 * r latent coordinates come from a mixture of
    `n_clusters` Gaussians (centres uniform in the unit cube, spread `sigma`), plus isotropic full-rank noise, then
    clipped and rounded like SIFT:  x = clip(round(300 * (z U) + 20 + noise * eps), 0, 218).
    Round 1's generator (isotropic 128-d Gaussian clusters, `gaussian_clusters` below) has no such structure: its
    intra-cluster distances concentrate, so the reference's SYMMETRIC PQ distance cannot rank them (recall@10 0.11) —
    a property of the data, not of the scan.  With r = 2 (32 intrinsic dimensions) symmetric PQ m=16 reaches
    recall@10 ~0.82 and IVF coverage rises gradually with nprobe (512 broad components cut by 4096 lists: measured
    0.42 / 0.81 / 0.82 / 0.82 / 0.82 at nprobe 1 / 8 / 16 / 32 / 64), i.e. the QPS/recall sweep means something.
    Base rows and queries are independent draws (different seeds) of the same distribution."""

    def __init__(self, d=128, latent_per_block=2, n_clusters=512, sigma=0.2, noise=1.0, seed=1, device="cuda"):
        assert d % 8 == 0
        self.d, self.r, self.m, self.sigma, self.noise, self.device = d, latent_per_block, d // 8, sigma, noise, device
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        u = torch.rand((self.m, self.r, 8), generator=g) + 0.1
        self.frames = (u / u.norm(dim=2, keepdim=True)).to(device)
        self.centers = torch.rand((n_clusters, self.m * self.r), generator=g).to(device)

    def draw(self, n, seed, chunk=1 << 20):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        a = torch.randint(0, self.centers.shape[0], (n,), generator=g)
        out = torch.empty((n, self.d), dtype=torch.float32, device=self.device)
        gd = torch.Generator(device=self.device)
        gd.manual_seed(seed)
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            z = self.centers[a[s:e].to(self.device)] + self.sigma * torch.randn((e - s, self.m * self.r), generator=gd,
                                                                                device=self.device)
            x = torch.einsum("nmr,mre->nme", z.view(e - s, self.m, self.r), self.frames).reshape(e - s, self.d)
            x = x * 300.0 + 20.0 + self.noise * torch.randn((e - s, self.d), generator=gd, device=self.device)
            out[s:e] = torch.clamp(torch.round(x), 0, 218)
        return out


class EmbedLike:
    """BASELINE.md C4 synthetic sentence embeddings (py/embed_1m_sentences.py's nomic-embed role): unit-norm 768-d rows
    of LOW RANK + noise — a shared r x d map (the "model") applied to per-user latent mixtures.  Round 1's isotropic
    768-d Gaussian has no neighbourhood structure at all (every centroid is equally far: SPANN recall 0.32 whatever the
    probe count).  Here a user's rows are  normalise(z A + noise * eps),  z from `n_clusters` broad Gaussians."""

    def __init__(self, d=768, rank=48, n_clusters=8, sigma=1.0, noise=0.01, seed=3, device="cuda"):
        self.d, self.rank, self.ncl, self.sigma, self.noise, self.device = d, rank, n_clusters, sigma, noise, device
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        self.A = (torch.randn((rank, d), generator=g) / d ** 0.5).to(device)

    def user(self, user_seed):
        """the latent cluster centres of one user"""
        g = torch.Generator(device="cpu")
        g.manual_seed(1_000_003 * 7 + user_seed)
        return torch.randn((self.ncl, self.rank), generator=g).to(self.device)

    def draw(self, centers, n, seed):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        a = torch.randint(0, self.ncl, (n,), generator=g).to(self.device)
        gd = torch.Generator(device=self.device)
        gd.manual_seed(seed)
        z = centers[a] + self.sigma * torch.randn((n, self.rank), generator=gd, device=self.device)
        x = z @ self.A + self.noise * torch.randn((n, self.d), generator=gd, device=self.device)
        return x / x.norm(dim=1, keepdim=True)


def gaussian_clusters(n, d=128, n_clusters=4096, sigma=20.0, seed=1, device="cuda"):
    """Round 1's C2/C3 base (isotropic Gaussian clusters, clipped to [0,218], rounded); kept so that round-1 numbers can be
    reproduced (`bench.py --data legacy`)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    centers = torch.rand((n_clusters, d), generator=g) * 218.0
    assign = torch.randint(0, n_clusters, (n,), generator=g)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    centers = centers.to(device)
    chunk = 1 << 18
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        noise = torch.randn((e - s, d), generator=g) * sigma
        out[s:e] = torch.clamp(torch.round(centers[assign[s:e].to(device)] + noise.to(device)), 0, 218)
    return out


def unit_gaussian(n, d, seed, device="cuda"):
    """Round 1's C4 rows (isotropic Gaussian, normalised); structure-free, kept for `--data legacy`."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    chunk = 1 << 16
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.randn((e - s, d), generator=g).to(device)
        out[s:e] = x / x.norm(dim=1, keepdim=True)
    return out


# ------------------------------------------------------------------------------------------ exact k-NN
def exact_knn(x, k, queries=None, chunk=4096, f64=False, exclude_self=True):
    """k nearest rows of x (squared L2) for every row of `queries` (default x itself).
    Returns (idx int64 [nq,k], sqdist f32 [nq,k]) ascending.  f64=True gives the float64 ground
    truth used for recall."""
    self_q = queries is None
    q = x if self_q else queries
    dt = torch.float64 if f64 else torch.float32
    xn = (x.to(dt) ** 2).sum(1)
    xt = x.to(dt).t().contiguous() if f64 else x.t().contiguous()
    nq = q.shape[0]
    idx = torch.empty((nq, k), dtype=torch.int64, device=x.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=x.device)
    if f64:
        chunk = max(64, chunk // 8)
    for s in range(0, nq, chunk):
        e = min(nq, s + chunk)
        qc = q[s:e].to(dt)
        dd = (qc ** 2).sum(1, keepdim=True) + xn[None, :] - 2.0 * (qc @ xt)
        if self_q and exclude_self:
            dd[torch.arange(e - s, device=x.device), torch.arange(s, e, device=x.device)] = float("inf")
        v, i = torch.topk(dd, k, dim=1, largest=False, sorted=True)
        idx[s:e] = i
        dist[s:e] = v.clamp_min(0).to(torch.float32)
    return idx, dist


# ------------------------------------------------------------------------------------------ HNSW bulk build
def _heuristic_prune(x, node_ids, cand_idx, cand_sq, max_neighbors, chunk=8192):
    """select_neighbors_heuristic (rs/index/src/hnsw/builder.rs:339-375), batched: walk the
    candidates nearest-first, keep e unless an already kept x is closer to e than the node is."""
    n, K = cand_idx.shape
    keep = torch.zeros((n, K), dtype=torch.bool, device=x.device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ci = cand_idx[s:e]
        cv = x[node_ids[ci]]                                   # [C,K,d]
        sq = (cv ** 2).sum(-1)
        pair = sq[:, :, None] + sq[:, None, :] - 2.0 * torch.bmm(cv, cv.transpose(1, 2))  # [C,K,K]
        dq = cand_sq[s:e]
        sel = torch.zeros((e - s, K), dtype=torch.bool, device=x.device)
        cnt = torch.zeros(e - s, dtype=torch.int32, device=x.device)
        valid = torch.isfinite(dq)
        for i in range(K):
            bad = ((pair[:, i, :] < dq[:, i:i + 1]) & sel).any(dim=1)
            ok = (~bad) & (cnt < max_neighbors) & valid[:, i]
            sel[:, i] = ok
            cnt += ok.to(torch.int32)
        keep[s:e] = sel
    return keep


def _layer_graph(x, node_ids, max_neighbors, kcand):
    """Adjacency (CSR over local indices -> global ids) of one layer."""
    n = node_ids.shape[0]
    dev = x.device
    if n <= 1:
        return np.zeros(n + 1, np.uint64), np.zeros(0, np.uint32)
    k = min(kcand, n - 1)
    idx, sq = exact_knn(x[node_ids], k)
    keep = _heuristic_prune(x, node_ids, idx, sq, max_neighbors)
    src = torch.arange(n, device=dev)[:, None].expand(n, k)[keep]
    dst = idx[keep]
    dd = sq[keep]
    # add reverse edges, dedup (src,dst), keep the max_neighbors nearest per node
    s2 = torch.cat([src, dst])
    d2 = torch.cat([dst, src])
    w2 = torch.cat([dd, dd])
    key = s2 * n + d2
    key, order = torch.sort(key)
    w2 = w2[order]
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    key, w2 = key[first], w2[first]
    s2, d2 = key // n, key % n
    o1 = torch.sort(w2, stable=True).indices
    s2, d2 = s2[o1], d2[o1]
    o2 = torch.sort(s2, stable=True).indices
    s2, d2 = s2[o2], d2[o2]
    counts = torch.bincount(s2, minlength=n)
    starts = torch.cumsum(counts, 0) - counts
    pos = torch.arange(s2.shape[0], device=dev) - starts[s2]
    m = pos < max_neighbors
    s2, d2 = s2[m], d2[m]
    counts = torch.bincount(s2, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(counts, 0)
    edges = node_ids[d2]
    return indptr.cpu().numpy().astype(np.uint64), edges.cpu().numpy().astype(np.uint32)


def bulk_hnsw(x, max_neighbors=32, max_layers=8, kcand=64, seed=1):
    """Returns (layers, levels): `layers` in muopdb_amd.formats.write_hnsw_index's CSR form
    (layer 0 first; points None for layer 0), entry point = first point of the top layer."""
    n = x.shape[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    u = torch.rand(n, generator=g).clamp_min(1e-12)
    # get_random_layer (builder.rs:332-337): floor(-ln(u)/ln(max_neighbors)), capped
    lv = torch.floor(-torch.log(u) / math.log(max_neighbors)).to(torch.int64).clamp_max(max_layers)
    top = int(lv.max().item())
    lv = lv.to(x.device)
    layers = []
    for layer in range(top + 1):
        ids = torch.nonzero(lv >= layer, as_tuple=False).reshape(-1)
        indptr, edges = _layer_graph(x, ids, max_neighbors, kcand)
        layers.append((None if layer == 0 else ids.cpu().numpy().astype(np.uint32), indptr, edges))
    return layers, lv.cpu().numpy()


def hnsw_files(x, doc_ids=None, **kw):
    """(index_bytes, vector_bytes) in the reference's HNSW formats for device rows x."""
    layers, _ = bulk_hnsw(x, **kw)
    n, d = x.shape
    if doc_ids is None:
        doc_ids = np.arange(n, dtype=np.uint64)
    return F.write_hnsw_index(layers, doc_ids, d), F.write_vector_file(x.cpu().numpy())


# ------------------------------------------------------------------------------------------ BASELINE config C5, one GPU's shard
def c5_index(ctx, total=100_000_000, world=8, rank=0, nlist=65536, d=128, chunk=2_000_000, seed=4, log=None):
    """BASELINE config C5 (100M x 128 as 16-byte PQ codes, IVF nlist 65 536) — as ONE of `world` GPUs holds it with the posting lists
    sharded l % world (world = 8: ~total/8 vectors), or whole (world = 1: all 100 M codes, 1.6 GB — the N = 1 anchor of C5's strong
    scaling): the full coarse quantizer, the shared PQ codebook, and the posting lists this rank owns with their codes.  The rows
    are generated chunk by chunk (SiftLike), assigned to their nearest of the nlist centroids and quantized ON THE DEVICE
    (mdb_pq_quantize_mem: the f32 rows never leave HBM); only the rows of owned lists are kept — the other ranks' lists are EMPTY in
    the returned index file, so loading it unsharded reproduces this rank's work exactly.
    Returns dict(index, vectors, pq, codebook, gen, n, nlist, owned_lists, centroids, list_sizes)."""
    import torch
    from .index import ProductQuantizer
    from . import build as B
    gen = SiftLike(d, seed=seed)
    sample = gen.draw(min(total, 2_000_000), seed=seed * 100)
    cent = B.kmeans(ctx, sample, nlist, iters=2, seed=seed)          # native Lloyd (mdb_kmeans_fit)
    cb = B.train_pq_codebook(ctx, sample, 8, 8, iters=6, seed=seed + 1, sample=100_000)
    pq = ProductQuantizer(d, 8, 8, cb)
    del sample
    nlist = cent.shape[0]
    keep_codes, keep_list = [], []
    done = 0
    ci = 0
    while done < total:
        m = min(chunk, total - done)
        x = gen.draw(m, seed=seed * 1000 + ci)
        a = gemm_assign_nearest(x, cent, chunk=1 << 14)   # 100M x 65 536 bulk labelling of SYNTHETIC rows: GEMM form
        if world > 1:
            own = (a % world) == rank
            xo = x[own].contiguous()
            ao = a[own]
        else:
            xo, ao = x.contiguous(), a
        keep_list.append(ao.cpu().numpy().astype(np.int32))
        codes_d = torch.empty((xo.shape[0], d // 8), dtype=torch.uint8, device=xo.device)
        if xo.shape[0]:
            pq.quantize_device(ctx, xo.data_ptr(), xo.shape[0], codes_d.data_ptr())
        keep_codes.append(codes_d.cpu().numpy())
        del x, xo, codes_d
        done += m
        ci += 1
        if log and ci % 10 == 0:
            log("c5 index: %d / %d rows assigned" % (done, total))
    lists = np.concatenate(keep_list)
    codes = np.concatenate(keep_codes)
    del keep_list, keep_codes
    n = codes.shape[0]
    # point ids in list order (what IvfBuilder::reindex produces, ivf/builder.rs:682): list l's points are contiguous
    order = np.argsort(lists, kind="stable")
    codes = codes[order]
    bounds = np.searchsorted(lists[order], np.arange(nlist + 1))
    del order, lists
    pls = [np.arange(bounds[i], bounds[i + 1], dtype=np.uint64) for i in range(nlist)]
    # this rank's global doc ids: an arbitrary injective labelling (rank-strided)
    docs = np.arange(n, dtype=np.uint64) * np.uint64(world) + np.uint64(rank)
    index = F.write_ivf_index(cent.cpu().numpy(), docs, pls, quantized_dimension=d // 8)
    return dict(index=index, vectors=F.write_vector_file(codes), pq=pq, codebook=cb, gen=gen, n=n, nlist=nlist,
                owned_lists=int((np.diff(bounds) > 0).sum()), centroids=cent, list_sizes=np.diff(bounds).astype(np.int64))


c5_shard = c5_index   # rounds 1-3's name (world = 8: a rank's shard)


def gemm_assign_nearest(x, c, chunk=1 << 16):
    """argmin_c ||x - c||^2 in GEMM form (torch) — bulk labelling of synthetic rows only; index builds use
    muopdb_amd.build.assign_nearest (mdb_ivf_assign: the reference's exact squared-L2 cascade)."""
    chunk = max(1024, min(chunk, (1 << 29) // max(1, c.shape[0])))  # distance block of at most 2 GiB
    cn = (c ** 2).sum(1)
    ct = c.t().contiguous()
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for s in range(0, x.shape[0], chunk):
        e = min(x.shape[0], s + chunk)
        dd = cn[None, :] - 2.0 * (x[s:e] @ ct)
        out[s:e] = dd.argmin(1)
    return out
