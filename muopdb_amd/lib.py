"""ctypes binding of libmuopdb_hip.so (include/muopdb_hip.h) — the only native entry into the
MI355X path.  There is NO CPU fallback: if the shared library is missing, or no HIP device is
present, every call raises (MuopdbError / OSError) — loudly, never silently degrading.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmuopdb_hip.so")

MDB_OK = 0
MDB_ERR_INVALID_ARG = 1
STATUS_NAMES = {0: "MDB_OK", 1: "MDB_ERR_INVALID_ARG", 2: "MDB_ERR_FORMAT", 3: "MDB_ERR_OOM", 4: "MDB_ERR_HIP",
                5: "MDB_ERR_NAN", 6: "MDB_ERR_NOT_FOUND", 7: "MDB_ERR_UNSUPPORTED", 8: "MDB_ERR_OUT_OF_RANGE"}
METRIC_L2, METRIC_DOT = 0, 1
QUANT_NONE, QUANT_PQ = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
IMPL_SCALAR, IMPL_SIMD, IMPL_STREAMING_SIMD = 0, 1, 2

EXPORTED_SYMBOLS = [
    "mdb_device_open", "mdb_device_close", "mdb_set_stream", "mdb_sync", "mdb_last_error", "mdb_get_stats",
    "mdb_version", "mdb_set_profiling", "mdb_get_profile", "mdb_l2_distance", "mdb_dot_distance", "mdb_lane_conforming_distance", "mdb_pq_quantize", "mdb_pq_quantize_mem", "mdb_pq_original_vector", "mdb_pq_distance", "mdb_ef_decode",
    "mdb_ivf_assign", "mdb_kmeans_fit", "mdb_flat_create", "mdb_flat_free", "mdb_flat_search", "mdb_flat_topk",
    "mdb_ivf_load", "mdb_ivf_free", "mdb_ivf_num_clusters", "mdb_ivf_num_vectors", "mdb_ivf_num_features", "mdb_ivf_num_resident_vectors",
    "mdb_ivf_find_nearest_centroids", "mdb_ivf_coarse_keys", "mdb_ivf_merge_coarse_keys", "mdb_ivf_search", "mdb_ivf_search_points", "mdb_ivf_invalidate",
    "mdb_ivf_is_invalidated",
    "mdb_hnsw_load", "mdb_hnsw_attach", "mdb_hnsw_free", "mdb_hnsw_num_vectors", "mdb_hnsw_ann_search",
    "mdb_spann_load", "mdb_spann_free", "mdb_spann_search", "mdb_spann_invalidate", "mdb_spann_is_invalidated",
    "mdb_multi_spann_load", "mdb_multi_spann_free", "mdb_multi_spann_num_users", "mdb_multi_spann_search",
    "mdb_multi_spann_invalidate", "mdb_multi_spann_is_invalidated", "mdb_multi_spann_replay_invalidations", "mdb_merge_shards",
    "mdb_shard_block_bytes", "mdb_shard_block_views", "mdb_merge_shards_packed", "mdb_allgather_merge",
    "mdb_odht_user_table", "mdb_hnsw_select_neighbors", "mdb_wait", "mdb_poll", "mdb_ivf_search_filtered", "mdb_ivf_attach", "mdb_ivf_search_submit", "mdb_hnsw_ann_search_submit",
    "mdb_spann_search_filtered", "mdb_spann_attach", "mdb_spann_search_submit",
    "mdb_multi_spann_search_filtered", "mdb_multi_spann_attach", "mdb_multi_spann_search_submit",
    "mdb_set_option", "mdb_get_option", "mdb_points_block_bytes", "mdb_points_block_views", "mdb_ivf_search_shard", "mdb_ivf_merge_shards",
    "mdb_spann_search_shard", "mdb_spann_merge_shards", "mdb_multi_spann_search_shard", "mdb_multi_spann_merge_shards", "mdb_spann_probe_row_words", "mdb_multi_spann_probes", "mdb_multi_spann_search_shard_probes",
    "mdb_allgather_blocks", "mdb_device_mem_info",
]


class MuopdbError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))


class U128(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


class QuantDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("metric", C.c_int), ("dimension", C.c_uint32),
                ("subvector_dimension", C.c_uint32), ("num_bits", C.c_uint32),
                ("codebook", C.POINTER(C.c_float)), ("codebook_len", C.c_size_t)]


class SearchParamsC(C.Structure):
    _fields_ = [("top_k", C.c_size_t), ("ef_construction", C.c_uint32), ("record_pages", C.c_int),
                ("num_explored_centroids", C.c_int64), ("centroid_distance_ratio", C.c_float)]


class UserIndexInfoC(C.Structure):
    _fields_ = [("user_id", U128)] + [(n, C.c_uint64) for n in (
        "centroid_vector_offset", "centroid_vector_len", "centroid_index_offset", "centroid_index_len",
        "ivf_vectors_offset", "ivf_vectors_len", "ivf_raw_vectors_offset", "ivf_raw_vectors_len",
        "ivf_index_offset", "ivf_index_len", "ivf_pq_codebook_offset", "ivf_pq_codebook_len")]


class Stats(C.Structure):
    _fields_ = [("scored_vectors", C.c_uint64), ("distance_evals", C.c_uint64), ("expanded_nodes", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64)]


_lib = None


def load():
    """dlopen libmuopdb_hip.so; raises OSError if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("libmuopdb_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as the
        # system one this library links).  If torch is importable, load it FIRST so both share
        # torch's runtime; the other order leaves torch with "No HIP GPUs are available".
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(LIB_PATH)
        _lib.mdb_last_error.restype = C.c_char_p
        _lib.mdb_version.restype = C.c_char_p
        for n in ("mdb_shard_block_bytes", "mdb_points_block_bytes", "mdb_spann_probe_row_words", "mdb_ivf_num_clusters", "mdb_ivf_num_vectors", "mdb_ivf_num_features", "mdb_ivf_num_resident_vectors", "mdb_hnsw_num_vectors",
                  "mdb_multi_spann_num_users"):
            if hasattr(_lib, n):
                getattr(_lib, n).restype = C.c_size_t
    return _lib


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def u8buf(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(b, dtype=np.uint8)


def u128_array(ids):
    arr = (U128 * max(len(ids), 1))()
    for i, d in enumerate(ids):
        arr[i].lo = d & 0xFFFFFFFFFFFFFFFF
        arr[i].hi = d >> 64
    return arr


class Context:
    """mdb_ctx: one per GPU / per process (one process per GPU)."""

    def __init__(self, gpu=0):
        self.lib = load()
        h = C.c_void_p()
        st = self.lib.mdb_device_open(C.c_int(gpu), C.byref(h))
        if st != MDB_OK:
            raise MuopdbError(st, "mdb_device_open(%d): no usable HIP device (the product path has no CPU fallback)" % gpu)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdb_device_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != MDB_OK:
            raise MuopdbError(st, (self.lib.mdb_last_error(self.h) or b"").decode())

    def set_option(self, name, value):
        """mdb_set_option: a tuning / test switch of THIS context (names: MDB_OPTIONS in csrc/mdb_common.h)."""
        self.check(self.lib.mdb_set_option(self.h, name.encode(), C.c_longlong(int(value))))

    def mem_info(self):
        """(free, total) bytes of the device after the stream drained: the drop of `free` across a load is the index's HBM footprint"""
        f, t = C.c_size_t(), C.c_size_t()
        self.check(self.lib.mdb_device_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def get_option(self, name):
        v = C.c_longlong()
        self.check(self.lib.mdb_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def option(self, name, value):
        """with ctx.option("MDB_FLAT_NO_MFMA", 1): ...  — set for the block, restored afterwards."""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.old = ctx.get_option(name)
                ctx.set_option(name, value)

            def __exit__(self_, *exc):
                ctx.set_option(name, self_.old)
        return _Scope()

    def set_stream(self, stream_ptr):
        self.check(self.lib.mdb_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self.check(self.lib.mdb_sync(self.h))

    def stats(self):
        s = Stats()
        self.check(self.lib.mdb_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def set_profiling(self, on):
        # True -> 3 (every bracketed kernel); 1 = scan kernels only, 2 = HNSW traversal only
        self.check(self.lib.mdb_set_profiling(self.h, C.c_int(3 if on is True else int(on))))

    def get_profile(self):
        """(summed dominant-kernel ms, launches) since the last call; synchronises."""
        ms, n = C.c_double(), C.c_uint64()
        self.check(self.lib.mdb_get_profile(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- unit seams
    def l2_distance(self, a, b, squared=False):
        a, b = f32(a), f32(b)
        a = a.reshape(-1, a.shape[-1]); b = b.reshape(-1, b.shape[-1])
        out = np.empty(a.shape[0], np.float32)
        self.check(self.lib.mdb_l2_distance(self.h, ptr(a, C.c_float), ptr(b, C.c_float), C.c_size_t(a.shape[0]),
                                            C.c_size_t(a.shape[1]), C.c_int(int(squared)), ptr(out, C.c_float)))
        return out

    def dot_distance(self, a, b):
        a, b = f32(a), f32(b)
        a = a.reshape(-1, a.shape[-1]); b = b.reshape(-1, b.shape[-1])
        out = np.empty(a.shape[0], np.float32)
        self.check(self.lib.mdb_dot_distance(self.h, ptr(a, C.c_float), ptr(b, C.c_float), C.c_size_t(a.shape[0]),
                                             C.c_size_t(a.shape[1]), ptr(out, C.c_float)))
        return out

    def lane_conforming_distance(self, a, b, lanes, metric=0):
        a, b = f32(a), f32(b)
        a = a.reshape(-1, a.shape[-1]); b = b.reshape(-1, b.shape[-1])
        out = np.empty(a.shape[0], np.float32)
        self.check(self.lib.mdb_lane_conforming_distance(self.h, ptr(a, C.c_float), ptr(b, C.c_float), C.c_size_t(a.shape[0]),
                                                         C.c_size_t(a.shape[1]), C.c_int(lanes), C.c_int(metric),
                                                         ptr(out, C.c_float)))
        return out

    def ef_decode(self, blob):
        b = u8buf(blob)
        n = C.c_size_t()
        cap = min(int(np.frombuffer(b[:8].tobytes(), np.uint64)[0]), b.size * 8) if b.size >= 8 else 0  # a list cannot hold more ids than bits
        out = np.empty(max(cap, 1), np.uint64)
        self.check(self.lib.mdb_ef_decode(self.h, ptr(b, C.c_uint8), C.c_size_t(b.size), ptr(out, C.c_uint64),
                                          C.c_size_t(cap), C.byref(n)))
        return out[:n.value].copy()


def quant_desc(kind=QUANT_NONE, metric=METRIC_L2, dimension=0, subvector_dimension=0, num_bits=0, codebook=None):
    """Returns (QuantDesc, keepalive)."""
    q = QuantDesc()
    q.kind, q.metric, q.dimension = kind, metric, dimension
    q.subvector_dimension, q.num_bits = subvector_dimension, num_bits
    cb = f32(codebook).reshape(-1) if codebook is not None else np.zeros(1, np.float32)
    q.codebook = ptr(cb, C.c_float)
    q.codebook_len = cb.size if codebook is not None else 0
    return q, cb
