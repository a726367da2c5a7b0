"""ctypes loader + thin numpy wrappers for oracle/libmuopdb_oracle.so (test infrastructure).

The C++ restatement cites the reference file:line per function; this file only marshals.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmuopdb_oracle.so")
_SRC = os.path.join(_HERE, "muopdb_oracle.cpp")

METRIC_L2, METRIC_DOT = 0, 1
QUANT_NONE, QUANT_PQ = 0, 1
PQ_SCALAR, PQ_SIMD, PQ_STREAMING = 0, 1, 2


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile).  No-op when the .so is up to date."""
    if (not force and os.path.exists(_SO) and os.path.exists(_SRC)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    if not os.path.exists(_SRC):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8buf(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _declare(L):
    f = C.c_float
    fp = C.POINTER(C.c_float)
    L.orc_lane_conforming.restype = C.c_float
    for name in ("orc_l2_squared", "orc_l2", "orc_l2_scalar", "orc_dot", "orc_dot_scalar"):
        fn = getattr(L, name)
        fn.restype = f
        fn.argtypes = [fp, fp, C.c_size_t]
    L.orc_pq_new.restype = C.c_void_p
    L.orc_ivf_open.restype = C.c_void_p
    L.orc_hnsw_open.restype = C.c_void_p
    L.orc_spann_open.restype = C.c_void_p
    L.orc_multi_spann_open.restype = C.c_void_p
    L.orc_hnsw_builder_new.restype = C.c_void_p
    L.orc_ef_encode.restype = C.c_long
    L.orc_ef_decode.restype = C.c_long
    L.orc_ivf_posting_list.restype = C.c_long
    L.orc_hnsw_edges.restype = C.c_long
    L.orc_hnsw_builder_num_layers.restype = C.c_uint32
    L.orc_hnsw_builder_entry_points.restype = C.c_uint32


# ---------------------------------------------------------------- distances
def l2_squared(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_l2_squared(_p(a, C.c_float), _p(b, C.c_float), a.size))


def l2(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_l2(_p(a, C.c_float), _p(b, C.c_float), a.size))


def l2_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_l2_scalar(_p(a, C.c_float), _p(b, C.c_float), a.size))


def dot(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot(_p(a, C.c_float), _p(b, C.c_float), a.size))


def dot_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot_scalar(_p(a, C.c_float), _p(b, C.c_float), a.size))


def lane_conforming(metric, lanes, a, b):
    """LaneConformingDistanceCalculator<LANES, D>::calculate_squared (lane_conforming.rs:22-26)."""
    a, b = _f32(a), _f32(b)
    return float(lib().orc_lane_conforming(C.c_int(metric), C.c_int(lanes), _p(a, C.c_float), _p(b, C.c_float), a.size))


def kmeans_fit(data, num_clusters, max_iter, tolerance, init_ids):
    """KMeansBuilder::fit with cluster_init_values (kmeans_builder.rs:116-360): (centroids [k][d], assignments [n], error, iterations)."""
    x = _f32(data)
    x = x.reshape(-1, x.shape[-1])
    n, d = x.shape
    k = min(num_clusters, n)
    init = np.ascontiguousarray(init_ids, np.uint64)
    if init.size != k:
        raise ValueError("init_ids must hold min(num_clusters, n) point ids")
    cent = np.empty((k, d), np.float32)
    lab = np.empty(n, np.uint32)
    err, it, kk = C.c_float(), C.c_uint32(), C.c_size_t()
    rc = lib().orc_kmeans_fit(_p(x, C.c_float), C.c_size_t(n), C.c_size_t(d), C.c_size_t(num_clusters), C.c_size_t(max_iter),
                              C.c_float(tolerance), _p(init, C.c_uint64), _p(cent, C.c_float), _p(lab, C.c_uint32), C.byref(err),
                              C.byref(it), C.byref(kk))
    if rc:
        raise ValueError("orc_kmeans_fit failed: %d" % rc)
    return cent, lab, float(err.value), int(it.value)


def ivf_assign(centroids, vectors, max_clusters_per_vector, distance_threshold):
    """IvfBuilder::build_posting_lists' assignment (ivf/builder.rs:267-326) -> (ids [n][mc] UINT32_MAX padded, counts [n])."""
    c, v = _f32(centroids), _f32(vectors)
    c = c.reshape(-1, c.shape[-1]); v = v.reshape(-1, v.shape[-1])
    ids = np.empty((v.shape[0], max_clusters_per_vector), np.uint32)
    cnt = np.empty(v.shape[0], np.uint32)
    rc = lib().orc_ivf_assign(_p(c, C.c_float), C.c_size_t(c.shape[0]), _p(v, C.c_float), C.c_size_t(v.shape[0]),
                              C.c_size_t(v.shape[1]), C.c_size_t(max_clusters_per_vector), C.c_float(distance_threshold),
                              _p(ids, C.c_uint32), _p(cnt, C.c_uint32))
    if rc == 2:
        raise IndexError("max_clusters_per_vector out of range")
    if rc:
        raise ValueError("NaN distance")
    return ids, cnt


def distance_many(metric, q, base):
    """metric: 0 sqrt-L2, 1 neg-dot, 2 squared L2.  Returns f32[n]."""
    q, base = _f32(q), _f32(base)
    n, d = base.shape
    out = np.empty(n, np.float32)
    lib().orc_distance_many(C.c_int(metric), _p(q, C.c_float), _p(base, C.c_float), C.c_size_t(n), C.c_size_t(d),
                            _p(out, C.c_float))
    return out


def flat_topk(metric, base, queries, k, threads=1):
    base, queries = _f32(base), _f32(queries)
    n, d = base.shape
    b = queries.shape[0]
    ids = np.empty((b, k), np.uint32)
    dist = np.empty((b, k), np.float32)
    rc = lib().orc_flat_topk(C.c_int(metric), _p(base, C.c_float), C.c_size_t(n), C.c_size_t(d),
                             _p(queries, C.c_float), C.c_size_t(b), C.c_size_t(k), _p(ids, C.c_uint32),
                             _p(dist, C.c_float), C.c_int(threads))
    if rc:
        raise ValueError("NaN distance (reference panics: NotNan::new().unwrap())")
    return ids, dist


# ---------------------------------------------------------------- PQ
class ProductQuantizer:
    """rs/quantization/src/pq/mod.rs (query-time functions)."""

    def __init__(self, dimension, subvector_dimension, num_bits, codebook, metric=METRIC_L2):
        self.dimension, self.subdim, self.num_bits, self.metric = dimension, subvector_dimension, num_bits, metric
        self.codebook = _f32(codebook).reshape(-1)
        self.h = lib().orc_pq_new(C.c_int(metric), C.c_size_t(dimension), C.c_size_t(subvector_dimension),
                                  C.c_uint32(num_bits), _p(self.codebook, C.c_float), C.c_size_t(self.codebook.size))
        if not self.h:
            raise ValueError("Vector dimension needs to be divisible by the subvector dimension.")
        self.m = dimension // subvector_dimension

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pq_free(C.c_void_p(self.h))
            self.h = None

    def quantize(self, v):
        v = _f32(v).reshape(-1, self.dimension)
        out = np.empty((v.shape[0], self.m), np.uint8)
        lib().orc_pq_quantize(C.c_void_p(self.h), _p(v, C.c_float), C.c_size_t(v.shape[0]), _p(out, C.c_uint8))
        return out

    def original_vector(self, codes):
        codes = np.ascontiguousarray(codes, np.uint8)
        out = np.empty(self.dimension, np.float32)
        lib().orc_pq_original_vector(C.c_void_p(self.h), _p(codes, C.c_uint8), _p(out, C.c_float))
        return out

    def distance(self, a, b, impl=PQ_STREAMING):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, self.m)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, self.m)
        out = np.empty(a.shape[0], np.float32)
        lib().orc_pq_distance(C.c_void_p(self.h), _p(a, C.c_uint8), _p(b, C.c_uint8), C.c_size_t(a.shape[0]),
                              C.c_int(impl), _p(out, C.c_float))
        return out


class Quant:
    """Quantizer descriptor handed to the index openers."""

    def __init__(self, kind=QUANT_NONE, metric=METRIC_L2, subdim=0, num_bits=0, codebook=None):
        self.kind, self.metric, self.subdim, self.num_bits = kind, metric, subdim, num_bits
        self.codebook = _f32(codebook).reshape(-1) if codebook is not None else np.zeros(0, np.float32)

    def args(self):
        return [C.c_int(self.kind), C.c_int(self.metric), C.c_size_t(self.subdim), C.c_uint32(self.num_bits),
                _p(self.codebook, C.c_float), C.c_size_t(self.codebook.size)]


# ---------------------------------------------------------------- Elias-Fano
def ef_encode(values, universe):
    """Returns (blob bytes, lower_bit_length, lower_bits_len, upper_bits_len)."""
    v = np.ascontiguousarray(values, np.uint64)
    L, lb, ub = C.c_uint64(), C.c_uint64(), C.c_uint64()
    need = lib().orc_ef_encode(_p(v, C.c_uint64), C.c_size_t(v.size), C.c_uint64(universe), None, C.c_size_t(0),
                               C.byref(L), C.byref(lb), C.byref(ub))
    if need < 0:
        raise ValueError("EF encode error (unsorted or > universe)")
    out = np.zeros(need, np.uint8)
    lib().orc_ef_encode(_p(v, C.c_uint64), C.c_size_t(v.size), C.c_uint64(universe), _p(out, C.c_uint8),
                        C.c_size_t(need), None, None, None)
    return out.tobytes(), L.value, lb.value, ub.value


def ef_decode(blob):
    b = _u8buf(blob)
    n = int(np.frombuffer(b[:8].tobytes(), np.uint64)[0]) if b.size >= 8 else 0
    out = np.empty(max(n, 1), np.uint64)
    r = lib().orc_ef_decode(_p(b, C.c_uint8), C.c_size_t(b.size), _p(out, C.c_uint64), C.c_size_t(out.size))
    if r < 0:
        raise ValueError("EF decode error")
    return out[:r].copy()


# ---------------------------------------------------------------- results
class _Res:
    def __init__(self, b, k):
        self.lo = np.empty((b, k), np.uint64)
        self.hi = np.empty((b, k), np.uint64)
        self.scores = np.empty((b, k), np.float32)
        self.counts = np.empty(b, np.uint32)
        self.found = np.ones(b, np.uint8)

    def args(self):
        return [_p(self.lo, C.c_uint64), _p(self.hi, C.c_uint64), _p(self.scores, C.c_float),
                _p(self.counts, C.c_uint32)]

    def doc_ids(self, qi):
        n = int(self.counts[qi])
        return [(int(self.hi[qi, i]) << 64) | int(self.lo[qi, i]) for i in range(n)]

    def id_with_scores(self, qi):
        """[(doc_id, score)] of row qi in IdWithScore order — the same accessor as muopdb_amd.index.SearchResult."""
        return [(d, float(self.scores[qi, i])) for i, d in enumerate(self.doc_ids(qi))]


def _split(doc_id):
    return C.c_uint64(doc_id & 0xFFFFFFFFFFFFFFFF), C.c_uint64(doc_id >> 64)


# ---------------------------------------------------------------- IVF
class BlockBasedIvf:
    """rs/index/src/ivf/block_based/index.rs"""

    def __init__(self, index_bytes, vectors_bytes, quant=None, index_offset=0, vector_offset=0):
        quant = quant or Quant()
        self._i, self._v, self._q = _u8buf(index_bytes), _u8buf(vectors_bytes), quant
        self.h = lib().orc_ivf_open(_p(self._i, C.c_uint8), C.c_size_t(self._i.size), C.c_size_t(index_offset),
                                    _p(self._v, C.c_uint8), C.c_size_t(self._v.size), C.c_size_t(vector_offset),
                                    *quant.args())
        if not self.h:
            raise ValueError("failed to open IVF index")
        hdr = np.zeros(8, np.uint64)
        lib().orc_ivf_header(C.c_void_p(self.h), _p(hdr, C.c_uint64))
        (self.num_features, self.quantized_dimension, self.num_clusters, self.num_vectors, self.doc_id_mapping_len,
         self.centroids_len, self.posting_lists_and_metadata_len, self.num_posting_lists) = [int(x) for x in hdr]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ivf_free(C.c_void_p(self.h))
            self.h = None

    def get_doc_id(self, idx):
        lo, hi = C.c_uint64(), C.c_uint64()
        if lib().orc_ivf_doc_id(C.c_void_p(self.h), C.c_size_t(idx), C.byref(lo), C.byref(hi)):
            raise IndexError("Index out of bound")
        return (hi.value << 64) | lo.value

    def get_centroid(self, idx):
        out = np.empty(self.num_features, np.float32)
        if lib().orc_ivf_centroid(C.c_void_p(self.h), C.c_size_t(idx), _p(out, C.c_float)):
            raise IndexError("Index out of bound")
        return out

    def get_posting_list(self, idx):
        out = np.empty(max(self.num_vectors * 4, 16), np.uint64)
        r = lib().orc_ivf_posting_list(C.c_void_p(self.h), C.c_size_t(idx), _p(out, C.c_uint64), C.c_size_t(out.size))
        if r < 0:
            raise IndexError("Index out of bound")
        return out[:r].copy()

    def find_nearest_centroids(self, queries, num_probes):
        q = _f32(queries).reshape(-1, self.num_features)
        out = np.empty((q.shape[0], num_probes), np.uint32)
        if lib().orc_ivf_find_nearest_centroids(C.c_void_p(self.h), _p(q, C.c_float), C.c_size_t(q.shape[0]),
                                                C.c_size_t(num_probes), _p(out, C.c_uint32)):
            raise ValueError("num_probes out of range (reference panics)")
        return out

    def search(self, queries, k, num_probes=None, probes=None, threads=1):
        q = _f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        res = _Res(b, k)
        if probes is not None:
            probes = np.ascontiguousarray(probes, np.uint32).reshape(b, -1)
            num_probes = probes.shape[1]
        rc = lib().orc_ivf_search(C.c_void_p(self.h), _p(q, C.c_float), C.c_size_t(b),
                                  _p(probes, C.c_uint32) if probes is not None else None, C.c_size_t(num_probes),
                                  C.c_size_t(k), *res.args(), C.c_int(threads))
        if rc:
            raise ValueError("IVF search error")
        return res

    def invalidate(self, doc_id):
        return bool(lib().orc_ivf_invalidate(C.c_void_p(self.h), *_split(doc_id)))

    def is_invalidated(self, doc_id):
        return bool(lib().orc_ivf_is_invalidated(C.c_void_p(self.h), *_split(doc_id)))


# ---------------------------------------------------------------- HNSW
class BlockBasedHnsw:
    """rs/index/src/hnsw/block_based/index.rs"""

    def __init__(self, index_bytes, vectors_bytes, dimension, quant=None, index_offset=0, vector_offset=0):
        quant = quant or Quant()
        self._i, self._v, self._q = _u8buf(index_bytes), _u8buf(vectors_bytes), quant
        self.dimension = dimension
        a = quant.args()
        self.h = lib().orc_hnsw_open(_p(self._i, C.c_uint8), C.c_size_t(self._i.size), C.c_size_t(index_offset),
                                     _p(self._v, C.c_uint8), C.c_size_t(self._v.size), C.c_size_t(vector_offset),
                                     a[0], a[1], C.c_size_t(dimension), a[2], a[3], a[4], a[5])
        if not self.h:
            raise ValueError("failed to open HNSW index")
        hdr = np.zeros(8, np.uint64)
        lib().orc_hnsw_header(C.c_void_p(self.h), _p(hdr, C.c_uint64))
        (self.quantized_dimension, self.num_layers, self.edges_len, self.points_len, self.edge_offsets_len,
         self.level_offsets_len, self.doc_id_mapping_len, self.entry_point) = [int(x) for x in hdr]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hnsw_free(C.c_void_p(self.h))
            self.h = None

    def get_edges_for_point(self, point, layer):
        out = np.empty(65536, np.uint32)
        r = lib().orc_hnsw_edges(C.c_void_p(self.h), C.c_uint32(point), C.c_uint32(layer), _p(out, C.c_uint32),
                                 C.c_size_t(out.size))
        return None if r < 0 else out[:r].copy()

    def ann_search(self, queries, k, ef, threads=1):
        q = _f32(queries).reshape(-1, self.dimension)
        b = q.shape[0]
        res = _Res(b, k)
        rc = lib().orc_hnsw_ann_search(C.c_void_p(self.h), _p(q, C.c_float), C.c_size_t(b), C.c_size_t(k),
                                       C.c_uint32(ef), *res.args(), C.c_int(threads))
        if rc:
            raise ValueError("HNSW search error")
        return res

    def stats(self, reset=True):
        e, x = C.c_uint64(), C.c_uint64()
        lib().orc_hnsw_stats(C.c_void_p(self.h), C.byref(e), C.byref(x), C.c_int(int(reset)))
        return e.value, x.value


# ---------------------------------------------------------------- SPANN
class SearchParams:
    """rs/config/src/search_params.rs"""

    def __init__(self, top_k, ef_construction, record_pages=False, num_explored_centroids=None,
                 centroid_distance_ratio=0.1):
        self.top_k, self.ef_construction, self.record_pages = top_k, ef_construction, record_pages
        self.num_explored_centroids, self.centroid_distance_ratio = num_explored_centroids, centroid_distance_ratio

    def args(self):
        n = -1 if self.num_explored_centroids is None else int(self.num_explored_centroids)
        return [C.c_size_t(self.top_k), C.c_uint32(self.ef_construction), C.c_int64(n),
                C.c_float(self.centroid_distance_ratio)]


class Spann:
    """rs/index/src/spann/index.rs"""

    def __init__(self, hnsw_index, hnsw_vectors, ivf_index, ivf_vectors, quant=None, offsets=(0, 0, 0, 0)):
        quant = quant or Quant()
        self._b = [_u8buf(x) for x in (hnsw_index, hnsw_vectors, ivf_index, ivf_vectors)]
        self._q = quant
        a = []
        for buf, off in zip(self._b, offsets):
            a += [_p(buf, C.c_uint8), C.c_size_t(buf.size), C.c_size_t(off)]
        self.h = lib().orc_spann_open(*a, *quant.args())
        if not self.h:
            raise ValueError("failed to open SPANN index")
        self.num_features = int(np.frombuffer(self._b[2][offsets[2] + 1: offsets[2] + 5].tobytes(), np.uint32)[0])

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_spann_free(C.c_void_p(self.h))
            self.h = None

    def search(self, queries, params, threads=1):
        q = _f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        res = _Res(b, params.top_k)
        rc = lib().orc_spann_search(C.c_void_p(self.h), _p(q, C.c_float), C.c_size_t(b), *params.args(), *res.args(),
                                    _p(res.found, C.c_uint8), C.c_int(threads))
        if rc:
            raise ValueError("SPANN search error")
        return res

    def invalidate(self, doc_id):
        return bool(lib().orc_spann_invalidate(C.c_void_p(self.h), *_split(doc_id)))

    def is_invalidated(self, doc_id):
        return bool(lib().orc_spann_is_invalidated(C.c_void_p(self.h), *_split(doc_id)))


class MultiSpannIndex:
    """rs/index/src/multi_spann/index.rs (user table = n x 112-byte UserIndexInfo records)."""

    def __init__(self, user_records, num_features, hnsw_index, hnsw_vectors, ivf_index, ivf_vectors, quant=None):
        quant = quant or Quant()
        self._u = _u8buf(user_records)
        self._b = [_u8buf(x) for x in (hnsw_index, hnsw_vectors, ivf_index, ivf_vectors)]
        self._q = quant
        self.num_features = num_features
        a = []
        for buf in self._b:
            a += [_p(buf, C.c_uint8), C.c_size_t(buf.size)]
        self.h = lib().orc_multi_spann_open(_p(self._u, C.c_uint8), C.c_size_t(self._u.size // 112),
                                            C.c_size_t(num_features), *a, *quant.args())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_multi_spann_free(C.c_void_p(self.h))
            self.h = None

    def search_for_user(self, user_ids, queries, params, threads=1):
        """Batch of (user_id, query) pairs -> _Res (found[qi]=0 means None)."""
        q = _f32(queries).reshape(-1, self.num_features)
        b = q.shape[0]
        ulo = np.array([u & 0xFFFFFFFFFFFFFFFF for u in user_ids], np.uint64)
        uhi = np.array([u >> 64 for u in user_ids], np.uint64)
        res = _Res(b, params.top_k)
        rc = lib().orc_multi_spann_search(C.c_void_p(self.h), _p(ulo, C.c_uint64), _p(uhi, C.c_uint64),
                                          _p(q, C.c_float), C.c_size_t(b), *params.args(), *res.args(),
                                          _p(res.found, C.c_uint8), C.c_int(threads))
        if rc:
            raise ValueError("multi-spann search error")
        return res

    def search_for_users(self, user_ids, query, params):
        q = _f32(query).reshape(-1)
        ulo = np.array([u & 0xFFFFFFFFFFFFFFFF for u in user_ids], np.uint64)
        uhi = np.array([u >> 64 for u in user_ids], np.uint64)
        res = _Res(1, params.top_k)
        rc = lib().orc_multi_spann_search_for_users(C.c_void_p(self.h), _p(ulo, C.c_uint64), _p(uhi, C.c_uint64),
                                                    C.c_size_t(len(user_ids)), _p(q, C.c_float), *params.args(),
                                                    *res.args())
        if rc:
            raise ValueError("multi-spann search error")
        return res

    def invalidate(self, user_id, doc_id):
        return bool(lib().orc_multi_spann_invalidate(C.c_void_p(self.h), *_split(user_id), *_split(doc_id)))

    def apply_pending_invalidations(self, directory):
        """MultiSpannIndex::new, multi_spann/index.rs:51-77: the segment's tombstone log is read into
        pending_invalidations (user id -> SET of doc ids); get_or_create_index :121-124 hands a user's set to
        Spann::invalidate_batch when that user's index is first opened — here at once, for the users the table holds
        (the others' records stay pending).  Returns the pending map."""
        pending = {}
        for user_id, doc_id in invalidated_ids_iter(directory):
            pending.setdefault(user_id, set()).add(doc_id)
        for user_id, docs in pending.items():
            for doc_id in docs:
                self.invalidate(user_id, doc_id)
        return pending


def invalidated_ids_iter(base_directory):
    """InvalidatedIdsStorage::read + ::iter, rs/index/src/ivf/files/invalidated_ids.rs:45-106, 183-256, restated on the
    standard library: the files named `invalidated_ids.bin.*` are COUNTED (read sorts them by numeric suffix only to size
    the last one), the iterator then opens `invalidated_ids.bin.<i>` for i < count — a name missing from that range is
    skipped (`.ok()`) — and yields (user id, doc id) from consecutive 32-byte little-endian records of each file; a file
    that ends inside a record panics ("Incomplete invalidation record at end of file")."""
    import os
    if not os.path.isdir(base_directory):
        return
    count = sum(1 for n in os.listdir(base_directory) if n.startswith("invalidated_ids.bin."))
    for i in range(count):
        path = os.path.join(base_directory, "invalidated_ids.bin.%d" % i)
        if not os.path.isfile(path):
            continue
        with open(path, "rb") as f:
            data = f.read()
        off = 0
        while off < len(data):
            if off > len(data) - 32:
                raise ValueError("Incomplete invalidation record at end of file")
            yield int.from_bytes(data[off:off + 16], "little"), int.from_bytes(data[off + 16:off + 32], "little")
            off += 32


# ---------------------------------------------------------------- ordering (K12)
class planner_filter:
    """`with planner_filter(bitmaps): ...` — the searches inside see per-query allow bitmaps over point ids
    (the Planner hook of scan_posting_list, ivf/block_based/index.rs:214-226).  bitmaps: uint32 [b][words] or
    [words] (shared by every query)."""

    def __init__(self, bitmaps):
        self.b = np.ascontiguousarray(bitmaps, dtype=np.uint32)

    def __enter__(self):
        stride = self.b.shape[-1] if self.b.ndim == 2 else 0
        lib().orc_set_filter(_p(self.b, C.c_uint32), C.c_size_t(stride))
        return self

    def __exit__(self, *exc):
        lib().orc_set_filter(None, C.c_size_t(0))
        return False


def sort_id_with_score(scores, doc_ids):
    s = _f32(scores)
    lo = np.array([d & 0xFFFFFFFFFFFFFFFF for d in doc_ids], np.uint64)
    hi = np.array([d >> 64 for d in doc_ids], np.uint64)
    perm = np.empty(len(doc_ids), np.uint32)
    lib().orc_sort_id_with_score(_p(s, C.c_float), _p(lo, C.c_uint64), _p(hi, C.c_uint64), C.c_size_t(len(doc_ids)),
                                 _p(perm, C.c_uint32))
    return perm


def heap_pop_order(dist, ids):
    d, i = _f32(dist), np.ascontiguousarray(ids, np.uint32)
    out = np.empty(i.size, np.uint32)
    lib().orc_heap_pop_order(_p(d, C.c_float), _p(i, C.c_uint32), C.c_size_t(i.size), _p(out, C.c_uint32))
    return out


# ---------------------------------------------------------------- HNSW builder (test-index synthesis)
class HnswBuilder:
    """rs/index/src/hnsw/builder.rs insert/select_neighbors_heuristic with a seeded rng."""

    def __init__(self, dim, max_neighbors, max_layers, ef_construction, metric=METRIC_L2, seed=1):
        self.dim = dim
        self.h = lib().orc_hnsw_builder_new(C.c_size_t(dim), C.c_size_t(max_neighbors), C.c_uint32(max_layers),
                                            C.c_uint32(ef_construction), C.c_int(metric), C.c_uint64(seed))
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hnsw_builder_free(C.c_void_p(self.h))
            self.h = None

    def insert(self, vectors):
        v = _f32(vectors).reshape(-1, self.dim)
        lib().orc_hnsw_builder_insert(C.c_void_p(self.h), _p(v, C.c_float), C.c_size_t(v.shape[0]))
        self.n += v.shape[0]

    def layers(self):
        """Returns [layer0, layer1, ...]; each layer = dict point -> np.uint32 edges (points ascending)."""
        out = []
        nl = lib().orc_hnsw_builder_num_layers(C.c_void_p(self.h))
        for l in range(nl):
            npnt, ne = C.c_uint64(), C.c_uint64()
            lib().orc_hnsw_builder_layer_size(C.c_void_p(self.h), C.c_uint32(l), C.byref(npnt), C.byref(ne))
            pts = np.empty(npnt.value, np.uint32)
            deg = np.empty(npnt.value, np.uint32)
            edg = np.empty(max(ne.value, 1), np.uint32)
            lib().orc_hnsw_builder_layer_export(C.c_void_p(self.h), C.c_uint32(l), _p(pts, C.c_uint32),
                                                _p(deg, C.c_uint32), _p(edg, C.c_uint32))
            d, o = {}, 0
            for p, g in zip(pts.tolist(), deg.tolist()):
                d[p] = edg[o:o + g].copy()
                o += g
            out.append(d)
        return out

    def entry_points(self):
        out = np.empty(max(self.n, 1), np.uint32)
        c = lib().orc_hnsw_builder_entry_points(C.c_void_p(self.h), _p(out, C.c_uint32), C.c_size_t(out.size))
        return out[:c].tolist()


def num_threads():
    return int(lib().orc_num_threads())


# --------------------------------------------------------------------------- IvfBuilder::reindex
def reassigned_ids(posting_lists, num_vectors):
    """IvfBuilder::get_reassigned_ids + assign_ids_until_last_stopping_point, rs/index/src/ivf/builder.rs:596-676, statement by
    statement (pure Python: test infrastructure for small cases).  The heap is BinaryHeap<Reverse<PostingListWithStoppingPoints>>:
    smallest first stopping point first, ties by the posting list's contents (:114-126)."""
    import heapq
    occurrence = {}
    for li, pl in enumerate(posting_lists):                      # build_posting_lists_with_stopping_points :557-593
        for v in pl:
            occurrence.setdefault(int(v), []).append(li)
    stops = [[] for _ in posting_lists]
    for v, where in occurrence.items():
        if len(where) > 1:
            for li in where:
                stops[li].append(v)
    heap = [(sorted(sp)[0], [int(v) for v in pl], sorted(sp)) for pl, sp in zip(posting_lists, stops) if sp]
    heapq.heapify(heap)
    assigned = [-1] * num_vectors
    cur = 0
    while heap:                                                  # :603-655
        first = heapq.heappop(heap)
        stop = first[2][0]
        working = [first]
        while heap and heap[0][2][0] == stop:
            working.append(heapq.heappop(heap))
        for _, pl, sp in working:
            for i, v in enumerate(pl):
                if v == stop:
                    if len(sp) > 1:
                        heapq.heappush(heap, (sp[1], pl[i + 1:], sp[1:]))
                    break
                if assigned[v] >= 0:
                    raise ValueError("Vectors that come before a stopping point should not be reassigned")
                assigned[v] = cur
                cur += 1
        assigned[stop] = cur
        cur += 1
    for pl in posting_lists:                                     # :663-674
        for v in pl:
            if assigned[int(v)] < 0:
                assigned[int(v)] = cur
                cur += 1
    return assigned


def reindex(posting_lists, doc_ids, vectors):
    """IvfBuilder::reindex :682-761 -> (posting lists, doc ids, vectors, mapping) in the new numbering."""
    ids = reassigned_ids(posting_lists, len(vectors))
    new_lists = [[ids[int(v)] for v in pl] for pl in posting_lists]
    n_valid = sum(1 for x in ids if x >= 0)
    docs, vecs = [None] * n_valid, [None] * n_valid
    for old, new in enumerate(ids):
        if new >= 0:
            docs[new], vecs[new] = doc_ids[old], vectors[old]
    return new_lists, docs, vecs, [x & 0xFFFFFFFF for x in ids]
