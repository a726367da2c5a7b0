"""Host-side mirror of the reference's search surface over the C ABI (include/muopdb_hip.h).

Class and method names follow the reference (Rust, rs/index + rs/quantization) so that the
parity tests read like the reference's own tests:

    reference                                             here
    ----------------------------------------------------  ---------------------------------------
    L2DistanceCalculator / DotProductDistanceCalculator   L2DistanceCalculator / DotProduct...
    NoQuantizer / ProductQuantizer                         NoQuantizer / ProductQuantizer
    BlockBasedIvf::{find_nearest_centroids, search, ...}  BlockBasedIvf (batched: B queries)
    BlockBasedHnsw::ann_search                             BlockBasedHnsw.ann_search
    Spann::search, MultiSpannIndex::search_for_user       Spann.search, MultiSpannIndex.search_for_user
    SearchParams, SearchResult{IdWithScore}                SearchParams, SearchResult

Every search takes a BATCH of queries (the reference handles one query per call and loops,
rs/index/src/collection/snapshot.rs:49-58); row i of a result is what the reference returns
for query i.  All compute runs in libmuopdb_hip.so on the GPU; nothing here falls back to CPU.
"""
import ctypes as C

import numpy as np

from . import lib as L

INF_ID = (1 << 128) - 1


class SearchParams:
    """rs/config/src/search_params.rs:1-34"""

    def __init__(self, top_k, ef_construction, record_pages=False):
        self.top_k = top_k
        self.ef_construction = ef_construction
        self.record_pages = record_pages
        self.num_explored_centroids = None
        self.centroid_distance_ratio = 0.1

    def with_num_explored_centroids(self, n):
        self.num_explored_centroids = n
        return self

    def with_centroid_distance_ratio(self, r):
        self.centroid_distance_ratio = r
        return self

    def to_c(self):
        p = L.SearchParamsC()
        p.top_k, p.ef_construction, p.record_pages = self.top_k, self.ef_construction, int(self.record_pages)
        p.num_explored_centroids = -1 if self.num_explored_centroids is None else int(self.num_explored_centroids)
        p.centroid_distance_ratio = self.centroid_distance_ratio
        return p


class SearchResult:
    """Batched SearchResult: row i = Vec<IdWithScore> of query i (rs/index/src/utils.rs:152-176)."""

    def __init__(self, b, k, doc_lo, doc_hi, scores, counts, found=None):
        self.b, self.k = b, k
        self.doc_lo, self.doc_hi, self.scores, self.counts = doc_lo, doc_hi, scores, counts
        self.found = found if found is not None else np.ones(b, np.uint8)

    def doc_ids(self, qi):
        n = int(self.counts[qi])
        return [(int(self.doc_hi[qi, i]) << 64) | int(self.doc_lo[qi, i]) for i in range(n)]

    def id_with_scores(self, qi):
        n = int(self.counts[qi])
        return list(zip(self.doc_ids(qi), self.scores[qi, :n].tolist()))


class _OutBuf:
    def __init__(self, b, k):
        self.ids = np.empty((b, max(k, 1), 2), np.uint64)
        self.scores = np.empty((b, max(k, 1)), np.float32)
        self.counts = np.zeros(b, np.uint32)
        self.found = np.ones(b, np.uint8)
        self.b, self.k = b, k

    def args(self):
        return [self.ids.ctypes.data_as(C.POINTER(L.U128)), L.ptr(self.scores, C.c_float),
                L.ptr(self.counts, C.c_uint32)]

    def result(self):
        return SearchResult(self.b, self.k, self.ids[:, :, 0], self.ids[:, :, 1], self.scores, self.counts, self.found)


# ------------------------------------------------------------------------------------------ distances
class L2DistanceCalculator:
    """rs/utils/src/distance/l2.rs — batched pairs on the GPU."""

    @staticmethod
    def calculate(ctx, a, b):
        return ctx.l2_distance(a, b, squared=False)

    @staticmethod
    def calculate_squared(ctx, a, b):
        return ctx.l2_distance(a, b, squared=True)


class DotProductDistanceCalculator:
    """rs/utils/src/distance/dot_product.rs"""

    @staticmethod
    def calculate(ctx, a, b):
        return ctx.dot_distance(a, b)


class NoQuantizer:
    """rs/quantization/src/noq/mod.rs"""

    def __init__(self, dimension, metric=L.METRIC_L2):
        self.dimension, self.metric = dimension, metric

    def quantized_dimension(self):
        return self.dimension

    def desc(self):
        return L.quant_desc(L.QUANT_NONE, self.metric, self.dimension)


class ProductQuantizer:
    """rs/quantization/src/pq/mod.rs (query-time functions)."""

    def __init__(self, dimension, subvector_dimension, num_bits, codebook, metric=L.METRIC_L2):
        if subvector_dimension == 0 or dimension % subvector_dimension != 0:
            raise ValueError("Vector dimension needs to be divisible by the subvector dimension.")
        self.dimension, self.subvector_dimension, self.num_bits, self.metric = (
            dimension, subvector_dimension, num_bits, metric)
        self.codebook = L.f32(codebook).reshape(-1)

    def quantized_dimension(self):
        return self.dimension // self.subvector_dimension

    def desc(self):
        return L.quant_desc(L.QUANT_PQ, self.metric, self.dimension, self.subvector_dimension, self.num_bits,
                            self.codebook)

    def quantize(self, ctx, vectors):
        v = L.f32(vectors).reshape(-1, self.dimension)
        out = np.empty((v.shape[0], self.quantized_dimension()), np.uint8)
        q, keep = self.desc()
        ctx.check(ctx.lib.mdb_pq_quantize(ctx.h, C.byref(q), L.ptr(v, C.c_float), C.c_size_t(v.shape[0]),
                                          L.ptr(out, C.c_uint8)))
        return out

    def quantize_device(self, ctx, rows_ptr, n, codes_ptr):
        """quantize n device-resident f32 rows into device codes [n][m] (mdb_pq_quantize_mem): an index build's corpus never
        leaves HBM."""
        q, keep = self.desc()
        ctx.check(ctx.lib.mdb_pq_quantize_mem(ctx.h, C.byref(q), C.c_void_p(rows_ptr), C.c_size_t(n), C.c_int(L.MEM_DEVICE),
                                              C.c_void_p(codes_ptr)))

    def original_vector(self, ctx, codes):
        """ProductQuantizer::original_vector (pq/mod.rs:184-200): codes [n][m] -> reconstructed vectors [n][dimension]."""
        m = self.quantized_dimension()
        c = np.ascontiguousarray(codes, np.uint8).reshape(-1, m)
        out = np.empty((c.shape[0], self.dimension), np.float32)
        q, keep = self.desc()
        ctx.check(ctx.lib.mdb_pq_original_vector(ctx.h, C.byref(q), L.ptr(c, C.c_uint8), C.c_size_t(c.shape[0]),
                                                 L.ptr(out, C.c_float)))
        return out

    def distance(self, ctx, a, b, implem=L.IMPL_STREAMING_SIMD):
        m = self.quantized_dimension()
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, m)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, m)
        out = np.empty(a.shape[0], np.float32)
        q, keep = self.desc()
        ctx.check(ctx.lib.mdb_pq_distance(ctx.h, C.byref(q), L.ptr(a, C.c_uint8), L.ptr(b, C.c_uint8),
                                          C.c_size_t(a.shape[0]), C.c_int(implem), L.ptr(out, C.c_float)))
        return out


def _quant(q, dimension):
    q = q or NoQuantizer(dimension)
    return q.desc()


# ------------------------------------------------------------------------------------------ flat
class FlatIndex:
    """Brute-force scan (the build's formulation of BASELINE config C1; ordering rules of
    BlockBasedIvf::find_nearest_centroids, rs/index/src/ivf/block_based/index.rs:147-163)."""

    def __init__(self, ctx, base, metric=L.METRIC_L2, device_ptr=None, n=None, d=None):
        self.ctx = ctx
        h = C.c_void_p()
        if device_ptr is not None:
            self.n, self.d = n, d
            ctx.check(ctx.lib.mdb_flat_create(ctx.h, C.c_void_p(device_ptr), C.c_size_t(n), C.c_size_t(d),
                                              C.c_int(metric), C.c_int(L.MEM_DEVICE), C.byref(h)))
        else:
            base = L.f32(base)
            self.n, self.d = base.shape
            ctx.check(ctx.lib.mdb_flat_create(ctx.h, L.ptr(base, C.c_float), C.c_size_t(self.n), C.c_size_t(self.d),
                                              C.c_int(metric), C.c_int(L.MEM_HOST), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.mdb_flat_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, queries, k):
        q = L.f32(queries).reshape(-1, self.d)
        b = q.shape[0]
        ids = np.empty((b, max(k, 1)), np.uint32)
        dist = np.empty((b, max(k, 1)), np.float32)
        counts = np.zeros(b, np.uint32)
        self.ctx.check(self.ctx.lib.mdb_flat_search(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.c_size_t(k),
                                                    C.c_int(L.MEM_HOST), L.ptr(ids, C.c_uint32),
                                                    L.ptr(dist, C.c_float), L.ptr(counts, C.c_uint32)))
        return ids[:, :k], dist[:, :k], counts

    def search_device(self, q_ptr, b, k, ids_ptr, dist_ptr, counts_ptr=None):
        """Device-resident queries / outputs (torch tensors' data_ptr()); enqueue only."""
        self.ctx.check(self.ctx.lib.mdb_flat_search(self.h, C.c_void_p(q_ptr), C.c_size_t(b), C.c_size_t(k),
                                                    C.c_int(L.MEM_DEVICE), C.c_void_p(ids_ptr), C.c_void_p(dist_ptr),
                                                    C.c_void_p(counts_ptr) if counts_ptr else None))


# ------------------------------------------------------------------------------------------ IVF

def _planner_args(planner):
    """per-call planner filter -> (allow pointer, n_bitmaps, words_per_bitmap, keepalive)"""
    if planner is None:
        return None, C.c_size_t(0), C.c_size_t(0), None
    bm = np.ascontiguousarray(planner, dtype=np.uint32)
    nb, words = (1, bm.shape[0]) if bm.ndim == 1 else bm.shape
    return L.ptr(bm, C.c_uint32), C.c_size_t(nb), C.c_size_t(words), bm


class Pending:
    """Result of a *_submit call: `wait()` completes it (mdb_wait) and returns the SearchResult."""

    def __init__(self, ctx, out, keep):
        self.ctx, self.out, self.keep = ctx, out, keep

    def done(self):
        return bool(self.ctx.lib.mdb_poll(self.ctx.h))

    def wait(self):
        self.ctx.check(self.ctx.lib.mdb_wait(self.ctx.h))
        self.keep = None
        return self.out.result()


def allow_bitmap(point_ids, num_points):
    """uint32 bitmap with the bits of `point_ids` set (the planner's kept ids)."""
    bm = np.zeros((num_points + 31) // 32, np.uint32)
    p = np.asarray(point_ids, np.int64)
    np.bitwise_or.at(bm, p >> 5, (np.uint32(1) << (p & 31).astype(np.uint32)))
    return bm


def _merge_shards(ctx, call, blocks, b, k, with_found):
    """Host-side convenience over mdb_*_merge_shards (device buffers): uploads the gathered blocks with torch, merges on
    the GPU, returns a SearchResult.  Production ranks keep everything on the device (muopdb_amd.distributed.PointsGather)."""
    import torch
    world = len(blocks)
    dev = torch.device("cuda", torch.cuda.current_device())
    recv = torch.from_numpy(np.concatenate([np.ascontiguousarray(x, np.uint8) for x in blocks])).to(dev)
    ke = max(k, 1)
    docs = torch.zeros((b, ke, 2), dtype=torch.int64, device=dev)
    sc = torch.zeros((b, ke), dtype=torch.float32, device=dev)
    cn = torch.zeros(b, dtype=torch.int32, device=dev)
    fd = torch.ones(b, dtype=torch.uint8, device=dev)
    args = [C.c_void_p(recv.data_ptr()), C.c_size_t(world), C.c_size_t(b), C.c_size_t(k), C.c_void_p(docs.data_ptr()),
            C.c_void_p(sc.data_ptr()), C.c_void_p(cn.data_ptr())]
    if with_found:
        args.append(C.c_void_p(fd.data_ptr()))
    ctx.check(call(*args))
    ctx.sync()
    ids = docs.cpu().numpy().view(np.uint64)
    return SearchResult(b, k, ids[:, :, 0], ids[:, :, 1], sc.cpu().numpy(), cn.cpu().numpy().view(np.uint32), fd.cpu().numpy())


class BlockBasedIvf:
    """rs/index/src/ivf/block_based/index.rs"""

    def __init__(self, ctx, index_bytes, vectors_bytes, quantizer=None, index_offset=0, vector_offset=0,
                 shard_rank=0, shard_world=1):
        self.ctx = ctx
        ib, vb = L.u8buf(index_bytes), L.u8buf(vectors_bytes)
        nf = int(np.frombuffer(ib[index_offset + 1:index_offset + 5].tobytes(), np.uint32)[0]) if ib.size >= index_offset + 5 else 0
        q, keep = _quant(quantizer, nf)
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_ivf_load(ctx.h, L.ptr(ib, C.c_uint8), C.c_size_t(ib.size), C.c_size_t(index_offset),
                                       L.ptr(vb, C.c_uint8), C.c_size_t(vb.size), C.c_size_t(vector_offset),
                                       C.byref(q), C.c_uint32(shard_rank), C.c_uint32(shard_world), C.byref(h)))
        self.h = h
        self.num_features = int(ctx.lib.mdb_ivf_num_features(h))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.mdb_ivf_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_clusters(self):
        return int(self.ctx.lib.mdb_ivf_num_clusters(self.h))

    def num_vectors(self):
        return int(self.ctx.lib.mdb_ivf_num_vectors(self.h))

    def num_resident_vectors(self):
        """posting-list entries held by this handle (a shard's share)"""
        return int(self.ctx.lib.mdb_ivf_num_resident_vectors(self.h))

    def find_nearest_centroids(self, queries, num_probes):
        q = L.f32(queries).reshape(-1, self.num_features)
        out = np.empty((q.shape[0], max(num_probes, 1)), np.uint32)
        self.ctx.check(self.ctx.lib.mdb_ivf_find_nearest_centroids(self.h, L.ptr(q, C.c_float), C.c_size_t(q.shape[0]),
                                                                   C.c_size_t(num_probes), C.c_int(L.MEM_HOST),
                                                                   L.ptr(out, C.c_uint32)))
        return out[:, :num_probes]

    def coarse_keys(self, queries, num_probes, first, count):
        """The num_probes nearest among centroids [first, first + count) as (distance, id) u64 keys (sharded coarse search)."""
        q = L.f32(queries).reshape(-1, self.num_features)
        out = np.empty((q.shape[0], num_probes), np.uint64)
        self.ctx.check(self.ctx.lib.mdb_ivf_coarse_keys(self.h, L.ptr(q, C.c_float), C.c_size_t(q.shape[0]), C.c_size_t(num_probes),
                                                        C.c_size_t(first), C.c_size_t(count), C.c_int(L.MEM_HOST), L.ptr(out, C.c_uint64)))
        return out

    def merge_coarse_keys(self, keys, num_probes):
        """keys [b][parts][num_probes] (every shard's coarse_keys row) -> probe ids [b][num_probes]."""
        k = np.ascontiguousarray(keys, np.uint64)
        b, parts = k.shape[0], k.shape[1]
        out = np.empty((b, num_probes), np.uint32)
        self.ctx.check(self.ctx.lib.mdb_ivf_merge_coarse_keys(self.h, L.ptr(k, C.c_uint64), C.c_size_t(b), C.c_size_t(parts),
                                                              C.c_size_t(num_probes), C.c_int(L.MEM_HOST), L.ptr(out, C.c_uint32)))
        return out

    def attach(self, ctx):
        """A second handle over the same resident index, bound to `ctx` (own stream / scratch): mdb_ivf_attach."""
        other = BlockBasedIvf.__new__(BlockBasedIvf)
        other.ctx, other.num_features = ctx, self.num_features
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_ivf_attach(ctx.h, self.h, C.byref(h)))
        other.h = h
        return other

    def search(self, queries, k, num_probes, planner=None):
        """BlockBasedIvf::search (index.rs:396-413).  planner: per-call allow bitmaps (uint32 [words] shared or
        [b][words] one per query) — the `planner` argument of scan_posting_list (index.rs:175)."""
        return self._search(queries, k, None, num_probes, planner)

    def search_with_centroids_and_remap(self, queries, nearest_centroid_ids, k, planner=None):
        """index.rs:298-332; nearest_centroid_ids: [B][P]."""
        p = np.ascontiguousarray(nearest_centroid_ids, np.uint32)
        p = p.reshape(-1, p.shape[-1])
        return self._search(queries, k, p, p.shape[1], planner)

    def _search(self, queries, k, probes, num_probes, planner=None):
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, k)
        if planner is None:
            self.ctx.check(self.ctx.lib.mdb_ivf_search(self.h, L.ptr(q, C.c_float), C.c_size_t(b),
                                                       L.ptr(probes, C.c_uint32) if probes is not None else None,
                                                       C.c_size_t(num_probes), C.c_size_t(k), C.c_int(L.MEM_HOST),
                                                       *out.args()))
        else:
            ap, nb, words, keep = _planner_args(planner)
            self.ctx.check(self.ctx.lib.mdb_ivf_search_filtered(self.h, L.ptr(q, C.c_float), C.c_size_t(b),
                                                                L.ptr(probes, C.c_uint32) if probes is not None else None,
                                                                C.c_size_t(num_probes), C.c_size_t(k), C.c_int(L.MEM_HOST),
                                                                ap, nb, words, *out.args()))
        return out.result()

    def search_submit(self, queries, k, num_probes, planner=None):
        """mdb_ivf_search_submit: enqueue and return; Pending.wait() gives the SearchResult."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, k)
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_ivf_search_submit(self.h, L.ptr(q, C.c_float), C.c_size_t(b), None, C.c_size_t(num_probes),
                                                          C.c_size_t(k), ap, nb, words, *out.args()))
        return Pending(self.ctx, out, None)

    def search_points(self, queries, k, num_probes, probes=None):
        """search_with_centroids (index.rs:250-286): top-k by (distance, point id), no remap."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        ids = np.empty((b, max(k, 1)), np.uint32)
        sc = np.empty((b, max(k, 1)), np.float32)
        cn = np.zeros(b, np.uint32)
        if probes is not None:
            probes = np.ascontiguousarray(probes, np.uint32).reshape(b, -1)
            num_probes = probes.shape[1]
        self.ctx.check(self.ctx.lib.mdb_ivf_search_points(self.h, L.ptr(q, C.c_float), C.c_size_t(b),
                                                          L.ptr(probes, C.c_uint32) if probes is not None else None,
                                                          C.c_size_t(num_probes), C.c_size_t(k), C.c_int(L.MEM_HOST),
                                                          L.ptr(ids, C.c_uint32), L.ptr(sc, C.c_float),
                                                          L.ptr(cn, C.c_uint32)))
        return ids[:, :k], sc[:, :k], cn

    # ---- exact list-sharded search (include/muopdb_hip.h, "list-sharded search, EXACT")
    def search_shard(self, queries, k, num_probes=0, probes=None, planner=None):
        """This rank's search_with_centroids rows as a POINTS block (uint8 array of mdb_points_block_bytes(b, k))."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        if probes is not None:
            probes = np.ascontiguousarray(probes, np.uint32).reshape(b, -1)
            num_probes = probes.shape[1]
        blk = np.zeros(int(self.ctx.lib.mdb_points_block_bytes(C.c_size_t(b), C.c_size_t(k))), np.uint8)
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_ivf_search_shard(self.h, L.ptr(q, C.c_float), C.c_size_t(b),
                                                         L.ptr(probes, C.c_uint32) if probes is not None else None,
                                                         C.c_size_t(num_probes), C.c_size_t(k), C.c_int(L.MEM_HOST), ap, nb, words,
                                                         L.ptr(blk, C.c_uint8)))
        return blk

    def merge_shards(self, blocks, b, k):
        """`blocks`: the ranks' points blocks (list of uint8 arrays, rank order) -> SearchResult equal to the unsharded search."""
        return _merge_shards(self.ctx, lambda *a: self.ctx.lib.mdb_ivf_merge_shards(self.h, *a), blocks, b, k, False)

    def invalidate(self, doc_id):
        return bool(self.invalidate_batch([doc_id])[0])

    def invalidate_batch(self, doc_ids):
        flags = np.zeros(max(len(doc_ids), 1), np.uint8)
        arr = L.u128_array(doc_ids)
        self.ctx.check(self.ctx.lib.mdb_ivf_invalidate(self.h, arr, C.c_size_t(len(doc_ids)), L.ptr(flags, C.c_uint8)))
        return flags[:len(doc_ids)]

    def is_invalidated(self, doc_id):
        flags = np.zeros(1, np.uint8)
        arr = L.u128_array([doc_id])
        self.ctx.check(self.ctx.lib.mdb_ivf_is_invalidated(self.h, arr, C.c_size_t(1), L.ptr(flags, C.c_uint8)))
        return bool(flags[0])


# ------------------------------------------------------------------------------------------ HNSW
class BlockBasedHnsw:
    """rs/index/src/hnsw/block_based/index.rs"""

    def __init__(self, ctx, index_bytes, vectors_bytes, dimension, quantizer=None, index_offset=0, vector_offset=0):
        self.ctx, self.dimension = ctx, dimension
        ib, vb = L.u8buf(index_bytes), L.u8buf(vectors_bytes)
        q, keep = _quant(quantizer, dimension)
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_hnsw_load(ctx.h, L.ptr(ib, C.c_uint8), C.c_size_t(ib.size), C.c_size_t(index_offset),
                                        L.ptr(vb, C.c_uint8), C.c_size_t(vb.size), C.c_size_t(vector_offset),
                                        C.byref(q), C.byref(h)))
        self.h = h

    def attach(self, ctx):
        """A second handle over the same resident graph, bound to `ctx` (its own stream and scratch): searches
        through different handles overlap on the device (mdb_hnsw_attach)."""
        other = BlockBasedHnsw.__new__(BlockBasedHnsw)
        other.ctx, other.dimension = ctx, self.dimension
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_hnsw_attach(ctx.h, self.h, C.byref(h)))
        other.h = h
        return other

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.mdb_hnsw_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ann_search(self, queries, k, ef):
        q = L.f32(queries).reshape(-1, self.dimension)
        b = q.shape[0]
        out = _OutBuf(b, k)
        self.ctx.check(self.ctx.lib.mdb_hnsw_ann_search(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.c_size_t(k),
                                                        C.c_uint32(ef), C.c_int(L.MEM_HOST), *out.args()))
        return out.result()

    def ann_search_submit(self, queries, k, ef):
        """mdb_hnsw_ann_search_submit: enqueue and return; Pending.wait() gives the SearchResult."""
        q = L.f32(queries).reshape(-1, self.dimension)
        b = q.shape[0]
        out = _OutBuf(b, k)
        self.ctx.check(self.ctx.lib.mdb_hnsw_ann_search_submit(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.c_size_t(k),
                                                               C.c_uint32(ef), *out.args()))
        return Pending(self.ctx, out, None)

    def ann_search_device(self, q_ptr, b, k, ef, ids_ptr, scores_ptr, counts_ptr):
        self.ctx.check(self.ctx.lib.mdb_hnsw_ann_search(self.h, C.c_void_p(q_ptr), C.c_size_t(b), C.c_size_t(k),
                                                        C.c_uint32(ef), C.c_int(L.MEM_DEVICE), C.c_void_p(ids_ptr),
                                                        C.c_void_p(scores_ptr), C.c_void_p(counts_ptr)))


def ivf_assign(ctx, centroids, vectors, max_clusters_per_vector=1, distance_threshold=0.1):
    """IvfBuilder::build_posting_lists' assignment step (rs/index/src/ivf/builder.rs:267-326) on the GPU:
    (centroid ids [n][mc] UINT32_MAX padded, counts [n])."""
    c, v = L.f32(centroids), L.f32(vectors)
    c = c.reshape(-1, c.shape[-1]); v = v.reshape(-1, v.shape[-1])
    ids = np.empty((v.shape[0], max_clusters_per_vector), np.uint32)
    cnt = np.empty(v.shape[0], np.uint32)
    ctx.check(ctx.lib.mdb_ivf_assign(ctx.h, L.ptr(c, C.c_float), C.c_size_t(c.shape[0]), L.ptr(v, C.c_float), C.c_size_t(v.shape[0]),
                                     C.c_size_t(v.shape[1]), C.c_size_t(max_clusters_per_vector), C.c_float(distance_threshold),
                                     C.c_int(L.MEM_HOST), L.ptr(ids, C.c_uint32), L.ptr(cnt, C.c_uint32)))
    return ids, cnt


def posting_lists_from_assignment(ids, counts, num_lists):
    """IvfBuilder::build_posting_lists :328-343: per centroid the sorted point ids (u64)."""
    n, mc = ids.shape
    mask = np.arange(mc)[None, :] < counts[:, None]
    cent = ids[mask].astype(np.int64)
    pts = np.repeat(np.arange(n, dtype=np.uint64), counts)
    order = np.lexsort((pts, cent))
    cent, pts = cent[order], pts[order]
    bounds = np.searchsorted(cent, np.arange(num_lists + 1))
    return [pts[bounds[i]:bounds[i + 1]] for i in range(num_lists)]


# ------------------------------------------------------------------------------------------ SPANN
class Spann:
    """rs/index/src/spann/index.rs"""

    def __init__(self, ctx, hnsw_index, hnsw_vectors, ivf_index, ivf_vectors, quantizer=None, offsets=(0, 0, 0, 0)):
        self.ctx = ctx
        bufs = [L.u8buf(x) for x in (hnsw_index, hnsw_vectors, ivf_index, ivf_vectors)]
        off = offsets[2]
        self.num_features = int(np.frombuffer(bufs[2][off + 1:off + 5].tobytes(), np.uint32)[0])
        q, keep = _quant(quantizer, self.num_features)
        a = []
        for buf, o in zip(bufs, offsets):
            a += [L.ptr(buf, C.c_uint8), C.c_size_t(buf.size), C.c_size_t(o)]
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_spann_load(ctx.h, *a, C.byref(q), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.mdb_spann_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def attach(self, ctx):
        """A second handle over the same resident centroid graph + posting lists, bound to `ctx` (mdb_spann_attach)."""
        other = Spann.__new__(Spann)
        other.ctx, other.num_features = ctx, self.num_features
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_spann_attach(ctx.h, self.h, C.byref(h)))
        other.h = h
        return other

    def search(self, queries, params, planner=None):
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, params.top_k)
        p = params.to_c()
        if planner is None:
            self.ctx.check(self.ctx.lib.mdb_spann_search(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.byref(p),
                                                         C.c_int(L.MEM_HOST), *out.args(), L.ptr(out.found, C.c_uint8)))
        else:
            ap, nb, words, keep = _planner_args(planner)
            self.ctx.check(self.ctx.lib.mdb_spann_search_filtered(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.byref(p),
                                                                  C.c_int(L.MEM_HOST), ap, nb, words, *out.args(),
                                                                  L.ptr(out.found, C.c_uint8)))
        return out.result()

    def search_submit(self, queries, params, planner=None):
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, params.top_k)
        p = params.to_c()
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_spann_search_submit(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.byref(p), ap, nb, words,
                                                            *out.args(), L.ptr(out.found, C.c_uint8)))
        return Pending(self.ctx, out, None)

    def search_shard(self, queries, params, planner=None):
        """This rank's rows of Spann::search before the remap, as a POINTS block (mdb_spann_search_shard)."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        blk = np.zeros(int(self.ctx.lib.mdb_points_block_bytes(C.c_size_t(b), C.c_size_t(params.top_k))), np.uint8)
        p = params.to_c()
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_spann_search_shard(self.h, L.ptr(q, C.c_float), C.c_size_t(b), C.byref(p), C.c_int(L.MEM_HOST),
                                                           ap, nb, words, L.ptr(blk, C.c_uint8)))
        return blk

    def merge_shards(self, blocks, b, k):
        return _merge_shards(self.ctx, lambda *a: self.ctx.lib.mdb_spann_merge_shards(self.h, *a), blocks, b, k, True)

    def invalidate(self, doc_id):
        flags = np.zeros(1, np.uint8)
        self.ctx.check(self.ctx.lib.mdb_spann_invalidate(self.h, L.u128_array([doc_id]), C.c_size_t(1),
                                                         L.ptr(flags, C.c_uint8)))
        return bool(flags[0])

    def is_invalidated(self, doc_id):
        flags = np.zeros(1, np.uint8)
        self.ctx.check(self.ctx.lib.mdb_spann_is_invalidated(self.h, L.u128_array([doc_id]), C.c_size_t(1),
                                                             L.ptr(flags, C.c_uint8)))
        return bool(flags[0])


class MultiSpannIndex:
    """rs/index/src/multi_spann/index.rs — `user_table` = 112-byte UserIndexInfo records."""

    def __init__(self, ctx, user_table, num_features, hnsw_index, hnsw_vectors, ivf_index, ivf_vectors,
                 quantizer=None, shard_rank=0, shard_world=1):
        self.ctx, self.num_features = ctx, num_features
        ut = L.u8buf(user_table)
        n_users = ut.size // 112
        users = (L.UserIndexInfoC * max(n_users, 1))()
        C.memmove(users, ut.ctypes.data, n_users * 112)
        bufs = [L.u8buf(x) for x in (hnsw_index, hnsw_vectors, ivf_index, ivf_vectors)]
        q, keep = _quant(quantizer, num_features)
        a = []
        for buf in bufs:
            a += [L.ptr(buf, C.c_uint8), C.c_size_t(buf.size)]
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_multi_spann_load(ctx.h, users, C.c_size_t(n_users), C.c_uint32(num_features), *a,
                                               C.byref(q), C.c_uint32(shard_rank), C.c_uint32(shard_world),
                                               C.byref(h)))
        self.h = h
        self.invalidated_ids_storage = None

    @classmethod
    def open_segment(cls, ctx, directory, shard_rank=0, shard_world=1, user_slots=None):
        """MultiSpannReader::read (rs/index/src/multi_spann/reader.rs:35) + MultiSpannIndex::new (multi_spann/index.rs:44-79):
        open a segment DIRECTORY as the reference writes it — `user_index_info` is the odht table, parsed
        by the library (mdb_odht_user_table); the five data files are mapped and handed to mdb_multi_spann_load; the tombstone
        log under `invalidated_ids_storage/` is read (InvalidatedIdsStorage::read) and applied to the resident users
        (mdb_multi_spann_replay_invalidations: what get_or_create_index :121-124 does per user on first open), and later
        `invalidate` calls append to it as the reference's do.  user_slots: keep only these records of the (id-sorted) user
        table — by-user sharding; the log's records of the other users are skipped by the replay."""
        import mmap
        import os
        from . import formats as F

        def mapped(rel):
            with open(os.path.join(directory, rel), "rb") as f:
                size = os.fstat(f.fileno()).st_size
                return np.frombuffer(mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ), np.uint8) if size else np.zeros(0, np.uint8)

        raw = mapped("user_index_info")
        n = C.c_size_t()
        ctx.check(ctx.lib.mdb_odht_user_table(L.ptr(raw, C.c_uint8), C.c_size_t(raw.size), None, C.c_size_t(0), C.byref(n)))
        users = (L.UserIndexInfoC * max(n.value, 1))()
        ctx.check(ctx.lib.mdb_odht_user_table(L.ptr(raw, C.c_uint8), C.c_size_t(raw.size), users, C.c_size_t(n.value), C.byref(n)))
        table = C.string_at(users, n.value * 112)
        if user_slots is not None:
            table = b"".join(table[int(u) * 112:(int(u) + 1) * 112] for u in user_slots)
        with open(os.path.join(directory, "centroids/quantizer/no_op_quantizer_config.yaml")) as f:
            d = F.parse_simple_yaml(f.read())["dimension"]
        quant = None
        pq_cfg = os.path.join(directory, "ivf/quantizer/product_quantizer_config.yaml")
        if os.path.exists(pq_cfg):
            with open(pq_cfg) as f:
                y = F.parse_simple_yaml(f.read())
            quant = ProductQuantizer(y["dimension"], y["subvector_dimension"], y["num_bits"],
                                     np.frombuffer(mapped("ivf/quantizer/codebook").tobytes(), np.float32))
        ms = cls(ctx, table, d, mapped("centroids/hnsw/index"), mapped("centroids/hnsw/vector_storage"), mapped("ivf/index"),
                 mapped("ivf/vectors"), quant, shard_rank, shard_world)
        ms.invalidated_ids_storage = F.InvalidatedIdsStorage.read(os.path.join(directory, "invalidated_ids_storage"))
        ms.replayed_invalidations = ms.replay_invalidations(ms.invalidated_ids_storage.record_bytes())
        return ms

    def replay_invalidations(self, records):
        """pending_invalidations applied (multi_spann/index.rs:64-77, 121-124): records = 32-byte (user id, doc id) pairs in
        log order; returns the number of documents newly tombstoned."""
        rec = L.u8buf(records)
        if rec.size % 32:
            raise ValueError("Incomplete invalidation record at end of file")
        n = C.c_size_t()
        self.ctx.check(self.ctx.lib.mdb_multi_spann_replay_invalidations(self.h, L.ptr(rec, C.c_uint8) if rec.size else None,
                                                                         C.c_size_t(rec.size // 32), C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.mdb_multi_spann_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_users(self):
        return int(self.ctx.lib.mdb_multi_spann_num_users(self.h))

    def attach(self, ctx):
        """A second handle over the same resident users, bound to `ctx` (mdb_multi_spann_attach)."""
        other = MultiSpannIndex.__new__(MultiSpannIndex)
        other.ctx, other.num_features = ctx, self.num_features
        h = C.c_void_p()
        ctx.check(ctx.lib.mdb_multi_spann_attach(ctx.h, self.h, C.byref(h)))
        other.h = h
        return other

    def search_for_user(self, user_ids, queries, params, planner=None):
        """Batch of (user_id, query) pairs; found[i] == 0 mirrors `None`.  planner: per-call allow bitmaps over each
        query's user-local point ids."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, params.top_k)
        p = params.to_c()
        if planner is None:
            self.ctx.check(self.ctx.lib.mdb_multi_spann_search(self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float),
                                                               C.c_size_t(b), C.byref(p), C.c_int(L.MEM_HOST), *out.args(),
                                                               L.ptr(out.found, C.c_uint8)))
        else:
            ap, nb, words, keep = _planner_args(planner)
            self.ctx.check(self.ctx.lib.mdb_multi_spann_search_filtered(self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float),
                                                                        C.c_size_t(b), C.byref(p), C.c_int(L.MEM_HOST), ap, nb, words,
                                                                        *out.args(), L.ptr(out.found, C.c_uint8)))
        return out.result()

    def search_for_user_submit(self, user_ids, queries, params, planner=None):
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        out = _OutBuf(b, params.top_k)
        p = params.to_c()
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_multi_spann_search_submit(self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float),
                                                                  C.c_size_t(b), C.byref(p), ap, nb, words, *out.args(),
                                                                  L.ptr(out.found, C.c_uint8)))
        return Pending(self.ctx, out, None)

    def search_shard(self, user_ids, queries, params, planner=None):
        """This rank's rows of search_for_user before the remap, as a POINTS block (mdb_multi_spann_search_shard)."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        blk = np.zeros(int(self.ctx.lib.mdb_points_block_bytes(C.c_size_t(b), C.c_size_t(params.top_k))), np.uint8)
        p = params.to_c()
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_multi_spann_search_shard(self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float), C.c_size_t(b),
                                                                 C.byref(p), C.c_int(L.MEM_HOST), ap, nb, words, L.ptr(blk, C.c_uint8)))
        return blk

    def merge_shards(self, user_ids, blocks, b, k):
        uids = L.u128_array(list(user_ids))
        return _merge_shards(self.ctx, lambda *a: self.ctx.lib.mdb_multi_spann_merge_shards(self.h, uids, *a), blocks, b, k, True)

    def probes(self, user_ids, queries, params):
        """The centroid stage alone (mdb_multi_spann_probes): per (user, query) pair the posting lists the closure kept, nearest
        first, BEFORE any scan, as uint32 rows [b][2 + ne] = (count, found, list ids).  Same on every list shard (every rank holds
        every user's centroid graph), so under list sharding ONE rank runs it per pair and the others receive the row."""
        q = L.f32(queries).reshape(-1, self.num_features)
        p = params.to_c()
        rows = np.zeros((q.shape[0], int(self.ctx.lib.mdb_spann_probe_row_words(C.byref(p)))), np.uint32)
        if q.shape[0] == 0:
            return rows
        self.ctx.check(self.ctx.lib.mdb_multi_spann_probes(self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float), C.c_size_t(q.shape[0]),
                                                           C.byref(p), C.c_int(L.MEM_HOST), L.ptr(rows, C.c_uint32)))
        return rows

    def search_shard_probes(self, user_ids, queries, params, rows, planner=None):
        """search_shard with the centroid stage handed in (mdb_multi_spann_search_shard_probes): the POINTS block is byte-identical
        to search_shard's when `rows` came from probes() with the same params."""
        q = L.f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        p = params.to_c()
        rows = np.ascontiguousarray(rows, np.uint32).reshape(b, int(self.ctx.lib.mdb_spann_probe_row_words(C.byref(p))))
        blk = np.zeros(int(self.ctx.lib.mdb_points_block_bytes(C.c_size_t(b), C.c_size_t(params.top_k))), np.uint8)
        ap, nb, words, keep = _planner_args(planner)
        self.ctx.check(self.ctx.lib.mdb_multi_spann_search_shard_probes(
            self.h, L.u128_array(list(user_ids)), L.ptr(q, C.c_float), C.c_size_t(b), C.byref(p), C.c_int(L.MEM_HOST),
            L.ptr(rows, C.c_uint32), ap, nb, words, L.ptr(blk, C.c_uint8)))
        return blk

    def search_for_users(self, user_ids, query, params):
        """Snapshot::search_for_users (collection/snapshot.rs:39-66): one query fanned to several
        users, concatenated, sorted by (score, doc id), truncated to top_k."""
        res = self.search_for_user(list(user_ids), np.repeat(L.f32(query).reshape(1, -1), len(user_ids), 0), params)
        rows = []
        for i in range(len(user_ids)):
            if res.found[i]:
                rows += res.id_with_scores(i)
        rows.sort(key=lambda r: (np.isnan(r[1]), r[1], r[0]))
        return rows[:params.top_k]

    def invalidate(self, user_id, doc_id):
        """MultiSpannIndex::invalidate :166-180: an EFFECTIVE invalidation is appended to the segment's tombstone log (a handle
        opened from a directory has one), so that the next open replays it."""
        flags = np.zeros(1, np.uint8)
        self.ctx.check(self.ctx.lib.mdb_multi_spann_invalidate(self.h, L.u128_array([user_id]), L.u128_array([doc_id]),
                                                               C.c_size_t(1), L.ptr(flags, C.c_uint8)))
        if flags[0] and getattr(self, "invalidated_ids_storage", None) is not None:
            self.invalidated_ids_storage.invalidate(user_id, doc_id)
        return bool(flags[0])

    def invalidate_batch(self, user_to_doc_ids):
        """MultiSpannIndex::invalidate_batch :189-223: {user id: [doc ids]} -> number effectively invalidated; the effective
        pairs go to the log in one batch."""
        pairs = []
        for user_id, doc_ids in user_to_doc_ids.items():
            doc_ids = list(doc_ids)
            flags = np.zeros(max(len(doc_ids), 1), np.uint8)
            self.ctx.check(self.ctx.lib.mdb_multi_spann_invalidate(self.h, L.u128_array([user_id]), L.u128_array(doc_ids),
                                                                   C.c_size_t(len(doc_ids)), L.ptr(flags, C.c_uint8)))
            pairs += [(user_id, d) for d, f in zip(doc_ids, flags) if f]
        if pairs and getattr(self, "invalidated_ids_storage", None) is not None:
            self.invalidated_ids_storage.invalidate_batch(pairs)
        return len(pairs)

    def is_invalidated(self, user_id, doc_id):
        """MultiSpannIndex::is_invalidated :229-232 (an unknown user raises: Err("User not found"))"""
        flags = np.zeros(1, np.uint8)
        self.ctx.check(self.ctx.lib.mdb_multi_spann_is_invalidated(self.h, L.u128_array([user_id]), L.u128_array([doc_id]),
                                                                   C.c_size_t(1), L.ptr(flags, C.c_uint8)))
        return bool(flags[0])


# ------------------------------------------------------------------------------------------ segment fan-out (callers of the path)
def _id_with_score_key(row):
    """IdWithScore order (rs/index/src/utils.rs:95-114): score ascending, NaN last, then doc id"""
    return (np.isnan(row[1]), row[1], row[0])


class PendingSegment:
    """PendingSegment::search_with_id while the merged index is not built yet (rs/index/src/segment/pending_segment.rs:285-335):
    the inner segments are searched with an OVER-FETCH of top_k + len(invalidated ids of the user) and NO planner
    (`search_with_id(s, id, query, &adjusted_params, None)`, :316-323), the temporarily invalidated documents are dropped from
    every inner result and the rows are CONCATENATED — neither sorted nor truncated here: the caller
    (Snapshot::search_for_user, collection/snapshot.rs:69-110) sorts and truncates.  The result is Some(..) even when no inner
    segment knows the user (an empty list, :333).  `inner_segments`: MultiSpannIndex handles (resident on the GPU);
    `temp_invalidated_ids`: user id -> doc ids."""

    def __init__(self, inner_segments, temp_invalidated_ids=None):
        self.inner_segments = list(inner_segments)
        self.temp_invalidated_ids = {u: set(v) for u, v in (temp_invalidated_ids or {}).items()}

    def invalidate(self, user_id, doc_id):
        self.temp_invalidated_ids.setdefault(user_id, set()).add(doc_id)

    def search_with_id(self, user_id, query, params):
        dead = self.temp_invalidated_ids.get(user_id) or set()
        adjusted = SearchParams(params.top_k + len(dead), params.ef_construction, params.record_pages)
        adjusted.num_explored_centroids = params.num_explored_centroids
        adjusted.centroid_distance_ratio = params.centroid_distance_ratio
        rows = []
        for seg in self.inner_segments:
            res = seg.search_for_user([user_id], np.asarray(query, np.float32).reshape(1, -1), adjusted)
            if res.found[0]:
                rows += [r for r in res.id_with_scores(0) if r[0] not in dead]
        return rows


class Snapshot:
    """Snapshot::search_for_user / search_for_users (rs/index/src/collection/snapshot.rs:39-110): every segment of the
    snapshot is searched (immutable segments = MultiSpannIndex handles, pending ones = PendingSegment), the rows are
    concatenated, sorted by IdWithScore and truncated to top_k; search_for_users does the same over several users."""

    def __init__(self, segments):
        self.segments = list(segments)

    def _planner_for(self, planner, si, user_id):
        """The reference builds ONE Planner per (segment, user) from the call's DocumentFilter (snapshot.rs:82-95): its bitmap
        indexes that user's point ids inside that segment.  `planner` is therefore a callable (segment index, user id) ->
        allow bitmap (uint32 words) or None (the segment has no term index: no filter).  A bare bitmap is accepted only where
        the reference would build exactly one planner: one finalized segment in the snapshot."""
        if planner is None:
            return None
        if callable(planner):
            return planner(si, user_id)
        if sum(0 if isinstance(s, PendingSegment) else 1 for s in self.segments) > 1:
            raise ValueError("Snapshot: one allow bitmap for several finalized segments (pass a callable (segment, user) -> bitmap)")
        return planner

    def search_for_user(self, user_id, query, params, planner=None):
        rows = []
        for si, seg in enumerate(self.segments):
            if isinstance(seg, PendingSegment):   # BoxedImmutableSegment::PendingSegment passes no planner (segment/mod.rs)
                rows += seg.search_with_id(user_id, query, params)
            else:
                res = seg.search_for_user([user_id], np.asarray(query, np.float32).reshape(1, -1), params,
                                          planner=self._planner_for(planner, si, user_id))
                if res.found[0]:
                    rows += res.id_with_scores(0)
        rows.sort(key=_id_with_score_key)
        return rows[:params.top_k]

    def search_for_users(self, user_ids, query, params, planner=None):
        if planner is not None and not callable(planner) and len(set(user_ids)) > 1:
            raise ValueError("Snapshot.search_for_users: one allow bitmap for several users (pass a callable (segment, user) -> bitmap)")
        rows = []
        for u in user_ids:
            rows += self.search_for_user(u, query, params, planner=planner)
        rows.sort(key=_id_with_score_key)
        return rows[:params.top_k]
