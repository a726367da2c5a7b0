"""Producers for the reference's on-disk index formats (numpy, vectorised).

These write byte-for-byte what the reference writers emit, so that a segment built here
can be opened by the reference readers and — the use on this path — by
`libmuopdb_hip.so`'s loaders (`mdb_ivf_load`, `mdb_hnsw_load`, ...), which parse exactly
these layouts out of the mmapped files:

* vector file            rs/index/src/vector/file.rs:213-225, async_storage.rs:112-136
* Elias-Fano posting list rs/compression/src/elias_fano/ef.rs:34-71, 129-215
* IVF `index` container  rs/index/src/ivf/writer.rs:255-355, posting_list/combined_file.rs:16-25
* HNSW `index` container rs/index/src/hnsw/writer.rs:24-33, 43-265
* multi-user concat      rs/index/src/multi_spann/writer.rs:171-250, user_index_info.rs:26-42

The byte layouts are pinned by the reference's own writer tests (SURVEY.md §8c K1-K3, K5,
K7, K13), re-encoded in tests/test_oracle_kat.py and tests/test_segment_formats.py.
"""
import struct

import numpy as np

U64_MAX = (1 << 64) - 1


# --------------------------------------------------------------------------- helpers
def pad_to(n, alignment):
    """rs/utils/src/io.rs:48-60 write_pad: number of zero bytes to reach `alignment`."""
    r = n % alignment
    return 0 if r == 0 else alignment - r


def doc_ids_to_array(doc_ids):
    """Python ints / (n,2) u64 array -> (n,2) little-endian u64 [lo, hi]."""
    if isinstance(doc_ids, np.ndarray) and doc_ids.ndim == 2 and doc_ids.dtype == np.uint64:
        return np.ascontiguousarray(doc_ids)
    if isinstance(doc_ids, np.ndarray) and doc_ids.ndim == 1:
        out = np.zeros((doc_ids.size, 2), np.uint64)
        out[:, 0] = doc_ids.astype(np.uint64)
        return out
    out = np.zeros((len(doc_ids), 2), np.uint64)
    for i, d in enumerate(doc_ids):
        out[i, 0] = d & U64_MAX
        out[i, 1] = d >> 64
    return out


def u128_bytes(v):
    return struct.pack("<QQ", v & U64_MAX, v >> 64)


# --------------------------------------------------------------------------- vector file (V1)
def write_vector_file(vectors):
    """`u64 n` + row-major little-endian values (f32 or u8)."""
    v = np.ascontiguousarray(vectors)
    if v.dtype not in (np.float32, np.uint8):
        raise TypeError("vector files hold f32 or u8 (PQ codes)")
    return struct.pack("<Q", v.shape[0]) + v.tobytes()


# --------------------------------------------------------------------------- Elias-Fano (E1)
def ef_lower_bit_length(universe, num_elem):
    """ef.rs:37-42: msb(universe / n) if universe > n else 0."""
    if universe > num_elem and num_elem > 0:
        return int(universe // num_elem).bit_length() - 1
    return 0


def ef_encode(values, universe=None):
    """Serialized posting list: `u64 n, u64 L, u64 lower_words, u64 upper_words, words...`.

    `universe` defaults to the last element — what the IVF writer passes
    (rs/index/src/ivf/writer.rs:271 `posting_list.last().unwrap_or(0)`).
    """
    v = np.ascontiguousarray(values, dtype=np.uint64)
    n = int(v.size)
    if universe is None:
        universe = int(v[-1]) if n else 0
    if n and (np.any(v[1:] < v[:-1]) or int(v[-1]) > universe):
        raise ValueError("Sequence is not sorted / exceeds universe")
    L = ef_lower_bit_length(universe, n)
    # lower bits, Lsb0 inside u64 words
    if L > 0 and n > 0:
        bits = ((v[:, None] >> np.arange(L, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).reshape(-1)
        lower = np.packbits(bits, bitorder="little")
    else:
        lower = np.zeros(0, np.uint8)
    lower_words = (n * L + 63) // 64
    lower = np.concatenate([lower, np.zeros(lower_words * 8 - lower.size, np.uint8)])
    # upper bits: element i sets bit (high_i + i)
    if n > 0:
        high = (v >> np.uint64(L)) if L < 64 else np.zeros(n, np.uint64)
        pos = high + np.arange(n, dtype=np.uint64)
        nbits = int(pos[-1]) + 1
        ub = np.zeros(nbits, np.uint8)
        ub[pos.astype(np.int64)] = 1
        upper = np.packbits(ub, bitorder="little")
    else:
        nbits = 0
        upper = np.zeros(0, np.uint8)
    upper_words = (nbits + 63) // 64
    upper = np.concatenate([upper, np.zeros(upper_words * 8 - upper.size, np.uint8)])
    return struct.pack("<QQQQ", n, L, lower_words, upper_words) + lower.tobytes() + upper.tobytes()


def ef_bits(values, universe):
    """(L, lower bit list, upper bit list) — the encoder's raw bit vectors (K1)."""
    blob = ef_encode(values, universe)
    n, L, lw, uw = struct.unpack_from("<QQQQ", blob, 0)
    lower = np.unpackbits(np.frombuffer(blob, np.uint8, lw * 8, 32), bitorder="little")[: n * L]
    v = np.asarray(values, np.uint64)
    nup = int((int(v[-1]) >> L) + n) if n else 0
    upper = np.unpackbits(np.frombuffer(blob, np.uint8, uw * 8, 32 + lw * 8), bitorder="little")[:nup]
    return int(L), lower.tolist(), upper.tolist()


# --------------------------------------------------------------------------- IVF container (I1)
def write_ivf_header(num_features, quantized_dimension, num_clusters, num_vectors, doc_id_mapping_len,
                     centroids_len, posting_lists_and_metadata_len):
    """ivf/writer.rs:280-295 (45 bytes, version 0)."""
    return struct.pack("<BIIIQQQQ", 0, num_features, quantized_dimension, num_clusters, num_vectors,
                       doc_id_mapping_len, centroids_len, posting_lists_and_metadata_len)


def write_posting_lists_and_metadata(posting_lists):
    """ivf/writer.rs:255-297: (metadata bytes, posting-list bytes)."""
    meta = [struct.pack("<Q", len(posting_lists))]
    pls = []
    off = 0
    for pl in posting_lists:
        blob = ef_encode(pl)
        meta.append(struct.pack("<QQ", len(blob), off))
        pls.append(blob)
        off += len(blob)
    return b"".join(meta), b"".join(pls)


def write_ivf_index(centroids, doc_ids, posting_lists, quantized_dimension=None):
    """The combined IVF `index` file (ivf/writer.rs:300-355)."""
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    num_clusters, num_features = c.shape
    if len(posting_lists) != num_clusters:
        raise ValueError("Mismatch between number of clusters and number of posting lists")
    ids = doc_ids_to_array(doc_ids)
    n = ids.shape[0]
    qd = num_features if quantized_dimension is None else quantized_dimension
    doc_map = u128_bytes(n) + ids.tobytes()
    cent = struct.pack("<Q", num_clusters) + c.tobytes()
    meta, pls = write_posting_lists_and_metadata(posting_lists)
    out = bytearray(write_ivf_header(num_features, qd, num_clusters, n, len(doc_map), len(cent),
                                     len(meta) + len(pls)))
    out += b"\0" * pad_to(len(out), 16)
    out += doc_map
    out += cent
    out += b"\0" * pad_to(len(out), 8)
    out += meta
    out += pls
    return bytes(out)


# --------------------------------------------------------------------------- HNSW container (H1)
def write_hnsw_index(layers, doc_ids, quantized_dimension):
    """The HNSW `index` file (hnsw/writer.rs:43-265).

    layers[0] is the bottom layer, layers[-1] the top.  Each layer is either a dict
    {point_id: iterable of neighbour ids} or a CSR tuple (points, indptr, edges) where
    `points` is None for layer 0 (slot == point id).  Upper-layer points are written in the
    order given (the reference iterates a HashSet: arbitrary); the FIRST point of the top
    layer is the entry point (graph_storage.rs:552-557).
    """
    edges_parts, points_parts, flat, level_offsets = [], [], [], []
    num_edges = 0   # running index into the edges section
    count = 0       # running number of edge_offsets ENTRIES (what level_offsets index)
    for layer_idx in range(len(layers) - 1, -1, -1):   # top layer first (:93-150)
        level_offsets.append(count)
        layer = layers[layer_idx]
        if isinstance(layer, dict):
            if layer_idx > 0:
                pts = list(layer.keys())
                nbrs = [layer[p] for p in pts]
                points = np.asarray(pts, np.uint32)
            else:
                max_point = max(layer.keys()) if layer else 0   # :124
                nbrs = [layer.get(p, ()) for p in range(max_point + 1)]
                points = None
            lens = [len(x) for x in nbrs]
            edges = (np.concatenate([np.asarray(x, np.uint32) for x in nbrs])
                     if nbrs else np.zeros(0, np.uint32))
            indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        else:
            points, indptr, edges = layer
            indptr = np.asarray(indptr, np.uint64)
            edges = np.asarray(edges, np.uint32)
            points = None if points is None else np.asarray(points, np.uint32)
        flat.append(indptr[:-1] + np.uint64(num_edges))
        count += indptr.size - 1
        num_edges += int(indptr[-1])
        edges_parts.append(edges.astype("<u4").tobytes())
        if layer_idx > 0:
            points_parts.append(points.astype("<u4").tobytes())
        else:
            flat.append(np.array([num_edges], np.uint64))  # sentinel after layer 0 (:140-142)
            count += 1
            level_offsets.append(count)
    eo = np.concatenate(flat).astype("<u8") if flat else np.zeros(0, "<u8")
    lo = np.asarray(level_offsets, "<u8")
    ids = doc_ids_to_array(doc_ids)
    edges_b, points_b = b"".join(edges_parts), b"".join(points_parts)
    eo_b, lo_b, ids_b = eo.tobytes(), lo.tobytes(), ids.tobytes()
    out = bytearray(struct.pack("<BIIQQQQQ", 0, quantized_dimension, len(layers), len(edges_b), len(points_b),
                                len(eo_b), len(lo_b), len(ids_b)))
    out += b"\0" * pad_to(len(out), 4)
    out += edges_b
    out += points_b
    out += b"\0" * pad_to(len(out), 8)
    out += eo_b
    out += lo_b
    out += b"\0" * pad_to(len(out), 16)
    out += ids_b
    return bytes(out)


# --------------------------------------------------------------------------- multi-user concat (M1)
USER_INDEX_INFO_FIELDS = (
    "centroid_vector_offset", "centroid_vector_len", "centroid_index_offset", "centroid_index_len",
    "ivf_vectors_offset", "ivf_vectors_len", "ivf_raw_vectors_offset", "ivf_raw_vectors_len",
    "ivf_index_offset", "ivf_index_len", "ivf_pq_codebook_offset", "ivf_pq_codebook_len")


def pack_user_index_info(user_id, **fields):
    """112-byte LE record (multi_spann/user_index_info.rs:26-42)."""
    return u128_bytes(user_id) + struct.pack("<12Q", *[int(fields.get(f, 0)) for f in USER_INDEX_INFO_FIELDS])


def unpack_user_index_info(rec):
    lo, hi = struct.unpack_from("<QQ", rec, 0)
    vals = struct.unpack_from("<12Q", rec, 16)
    d = dict(zip(USER_INDEX_INFO_FIELDS, vals))
    d["user_id"] = (hi << 64) | lo
    return d


def concat_multi_spann(users):
    """Concatenate per-user SPANN files into the 5 shared files (multi_spann/writer.rs:171-250).

    users: dict user_id -> dict(hnsw_index, hnsw_vectors, ivf_index, ivf_vectors,
                                 [ivf_raw_vectors], [codebook])  (bytes each)
    Returns dict(hnsw_index, hnsw_vectors, ivf_index, ivf_vectors, ivf_raw_vectors, codebook,
                 user_table) where user_table is this build's flat table: the 112-byte
    records sorted by user id (the reference stores the same records in an `odht` 0.3.1
    hash table whose byte format is not under /root/reference — SURVEY.md §8c).
    """
    out = {k: bytearray() for k in ("hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors", "ivf_raw_vectors",
                                    "codebook")}
    recs = []
    for uid in sorted(users.keys()):
        u = users[uid]
        f = {}
        out["hnsw_index"] += b"\0" * pad_to(len(out["hnsw_index"]), 16)
        f["centroid_index_offset"] = len(out["hnsw_index"])
        out["hnsw_index"] += u["hnsw_index"]
        f["centroid_index_len"] = len(u["hnsw_index"])
        out["hnsw_vectors"] += b"\0" * pad_to(len(out["hnsw_vectors"]), 8)
        f["centroid_vector_offset"] = len(out["hnsw_vectors"])
        out["hnsw_vectors"] += u["hnsw_vectors"]
        f["centroid_vector_len"] = len(u["hnsw_vectors"])
        out["ivf_index"] += b"\0" * pad_to(len(out["ivf_index"]), 16)
        f["ivf_index_offset"] = len(out["ivf_index"])
        out["ivf_index"] += u["ivf_index"]
        f["ivf_index_len"] = len(u["ivf_index"])
        out["ivf_vectors"] += b"\0" * pad_to(len(out["ivf_vectors"]), 8)
        f["ivf_vectors_offset"] = len(out["ivf_vectors"])
        out["ivf_vectors"] += u["ivf_vectors"]
        f["ivf_vectors_len"] = len(u["ivf_vectors"])
        raw = u.get("ivf_raw_vectors", b"")
        f["ivf_raw_vectors_offset"] = len(out["ivf_raw_vectors"])  # no pad (:230-235)
        out["ivf_raw_vectors"] += raw
        f["ivf_raw_vectors_len"] = len(raw)
        cb = u.get("codebook")
        if cb:
            out["codebook"] += b"\0" * pad_to(len(out["codebook"]), 8)
            f["ivf_pq_codebook_offset"] = len(out["codebook"])
            out["codebook"] += cb
            f["ivf_pq_codebook_len"] = len(cb)
        recs.append(pack_user_index_info(uid, **f))
    res = {k: bytes(v) for k, v in out.items()}
    res["user_table"] = b"".join(recs)
    return res


# --------------------------------------------------------------------------- quantizer configs
def no_op_quantizer_config_yaml(dimension):
    """rs/quantization/src/noq/mod.rs:64-73 (`no_op_quantizer_config.yaml`)."""
    return "dimension: %d\n" % dimension


def product_quantizer_config_yaml(dimension, subvector_dimension, num_bits):
    """rs/quantization/src/pq/mod.rs:34-39 (`product_quantizer_config.yaml`)."""
    return "dimension: %d\nsubvector_dimension: %d\nnum_bits: %d\n" % (dimension, subvector_dimension, num_bits)


def parse_simple_yaml(text):
    """The quantizer configs are flat `key: int` maps."""
    out = {}
    for line in text.splitlines():
        line = line.strip()
        if not line or line.startswith("#") or line == "---":
            continue
        k, v = line.split(":", 1)
        out[k.strip()] = int(v.strip())
    return out


# --------------------------------------------------------------------------- odht 0.3.1 table (`user_index_info`)
# The reference keeps the UserIndexInfo records in an `odht` on-disk hash table (multi_spann/writer.rs:253-259:
# HashTableOwned::<HashConfig>::with_capacity(n_users, 90), insert per user, raw_bytes(); read back by
# HashTable::from_raw_bytes, multi_spann/index.rs:50).  The crate (Cargo.lock: odht 0.3.1) is NOT under /root/reference, so
# this is a restatement of its PUBLISHED layout — a SwissTable-style open-addressing table — and is PARITY-UNPINNED
# against the crate itself (no crate, no golden file here); tests pin it with a hand-computed table and by round trip:
#   header, 32 bytes:  tag "ODHT" | size_of_metadata u8 = 1 | size_of_key u8 = 16 | size_of_value u8 = 112 |
#                      size_of_header u8 = 32 | item_count u64 LE | slot_count u64 LE | file_format_version [0,0,0,2] |
#                      max_load_factor u16 LE (percent * 65535 / 100) | 2 bytes padding
#   entries:           slot_count x { key [16] | value [112] }   (empty slots are all zero)
#   metadata:          slot_count + 16 control bytes: 0xFF = empty, else h2 = top 7 bits of the key's hash; the first 16
#                      bytes are mirrored after the end so that an unaligned 16-byte group read never wraps
#   hash:              FxHashFn over the encoded key (HashConfig::H, user_index_info.rs:84-90): for every LE u32 word w:
#                      h = (rotl(h, 5) ^ w) * 0x9e3779b9 (mod 2^32)
#   probing:           groups of 16 control bytes starting at h & (slot_count - 1), then triangular steps of 16
#                      (index += stride, stride += 16); a key goes into the first empty slot of the first group that has one
#   slot_count:        max(16, next_power_of_two(ceil(n * 65535 / factor)))
ODHT_GROUP = 16
ODHT_EMPTY = 0xFF   # what the writer fills empty control bytes with; READERS test bit 7 (h2 is 7 bits: odht's group query takes
                    # the movemask of the control bytes), so a table whose writer used 0x80 reads the same


def _odht_is_empty(ctrl):
    return (ctrl & 0x80) != 0


def fx_hash32(data):
    """odht::FxHashFn::hash — u32 words, then a u16, then a u8 tail."""
    h = 0
    i, n = 0, len(data)

    def add(hv, v):
        return ((((hv << 5) | (hv >> 27)) & 0xFFFFFFFF) ^ v) * 0x9E3779B9 & 0xFFFFFFFF

    while n - i >= 4:
        h = add(h, int.from_bytes(data[i:i + 4], "little"))
        i += 4
    if n - i >= 2:
        h = add(h, int.from_bytes(data[i:i + 2], "little"))
        i += 2
    if n - i >= 1:
        h = add(h, data[i])
    return h


def odht_slots_needed(item_count, max_load_factor_percent=90):
    factor = (0xFFFF * max_load_factor_percent) // 100
    need = (item_count * 0xFFFF + factor - 1) // factor
    p = 1
    while p < need:
        p <<= 1
    return max(p, ODHT_GROUP), factor


def _odht_probe(h, mask):
    index, stride = h & mask, 0
    while True:
        yield index
        stride += ODHT_GROUP
        index = (index + stride) & mask


def odht_table(entries, key_size=16, value_size=112, max_load_factor_percent=90):
    """raw_bytes() of an odht table holding `entries` = [(key bytes, value bytes)] inserted in that order."""
    slots, factor = odht_slots_needed(len(entries), max_load_factor_percent)
    mask = slots - 1
    esz = key_size + value_size
    data = bytearray(slots * esz)
    meta = bytearray([ODHT_EMPTY]) * (slots + ODHT_GROUP)
    count = 0
    for key, value in entries:
        assert len(key) == key_size and len(value) == value_size
        h = fx_hash32(key)
        h2 = h >> 25
        done = False
        for start in _odht_probe(h, mask):
            group = [(start + j) & mask for j in range(ODHT_GROUP)]
            for idx in group:                       # an equal key already present: the value is replaced
                if meta[idx] == h2 and bytes(data[idx * esz:idx * esz + key_size]) == key:
                    data[idx * esz + key_size:(idx + 1) * esz] = value
                    done = True
                    break
            if done:
                break
            empty = next((idx for idx in group if _odht_is_empty(meta[idx])), None)
            if empty is not None:
                data[empty * esz:(empty + 1) * esz] = key + value
                meta[empty] = h2
                if empty < ODHT_GROUP:
                    meta[slots + empty] = h2
                count += 1
                break
    header = b"ODHT" + bytes([1, key_size, value_size, 32]) + struct.pack("<QQ", count, slots) + bytes([0, 0, 0, 2]) + \
        struct.pack("<H", factor) + b"\0\0"
    return header + bytes(data) + bytes(meta)


def odht_entries(raw, key_size=16, value_size=112):
    """[(key, value)] of every occupied slot of an odht table, in slot order (what HashTable::iter yields)."""
    if len(raw) < 32 or raw[:4] != b"ODHT":
        raise ValueError("not an odht table")
    if raw[4] != 1 or raw[5] != key_size or raw[6] != value_size or raw[7] != 32 or bytes(raw[24:28]) != bytes([0, 0, 0, 2]):
        raise ValueError("odht header does not describe a %d/%d-byte table of format version 2" % (key_size, value_size))
    count, slots = struct.unpack_from("<QQ", raw, 8)
    esz = key_size + value_size
    if slots & (slots - 1) or len(raw) != 32 + slots * esz + slots + ODHT_GROUP:
        raise ValueError("odht table size does not match its slot count")
    meta = raw[32 + slots * esz:]
    out = []
    for i in range(slots):
        if not _odht_is_empty(meta[i]):
            e = raw[32 + i * esz:32 + (i + 1) * esz]
            out.append((bytes(e[:key_size]), bytes(e[key_size:])))
    if len(out) != count:
        raise ValueError("odht item_count %d != %d occupied slots" % (count, len(out)))
    return out


def odht_get(raw, key, key_size=16, value_size=112):
    """HashTable::get: the value stored under `key`, or None."""
    slots = struct.unpack_from("<Q", raw, 16)[0]
    mask, esz = slots - 1, key_size + value_size
    meta = raw[32 + slots * esz:]
    h = fx_hash32(key)
    h2 = h >> 25
    for start in _odht_probe(h, mask):
        group = [(start + j) & mask for j in range(ODHT_GROUP)]
        for idx in group:
            if meta[idx] == h2 and bytes(raw[32 + idx * esz:32 + idx * esz + key_size]) == key:
                return bytes(raw[32 + idx * esz + key_size:32 + (idx + 1) * esz])
        if any(_odht_is_empty(meta[idx]) for idx in group):
            return None


def user_index_info_table(user_table):
    """flat 112-byte records -> the `user_index_info` file (odht: key = user id u128 LE, value = the 112-byte record)."""
    recs = [bytes(user_table[i:i + 112]) for i in range(0, len(user_table), 112)]
    return odht_table([(r[:16], r) for r in recs])


def user_table_from_odht(raw):
    """`user_index_info` file -> flat 112-byte records sorted by user id (mdb_multi_spann_load's `users`)."""
    vals = [v for _, v in odht_entries(raw)]
    vals.sort(key=lambda r: int.from_bytes(r[:16], "little"))
    return b"".join(vals)


# --------------------------------------------------------------------------- tombstone log (delete path, read at open)
BYTES_PER_INVALIDATION = 32


class InvalidatedIdsStorage:
    """rs/index/src/ivf/files/invalidated_ids.rs:9-215 — the append log of (user id, doc id) tombstones a segment keeps
    under `invalidated_ids_storage/`: files `invalidated_ids.bin.<i>` of 32-byte records (u128 LE user id, u128 LE doc id),
    every file but the last `backing_file_size` bytes long (rounded DOWN to whole records, :33-35).  MultiSpannIndex::new
    reads it into `pending_invalidations` and every user's index is tombstoned from there when it is first opened
    (multi_spann/index.rs:51-77, 121-124); here all users are resident from the load on, so the open applies the whole log
    (MultiSpannIndex.open_segment -> mdb_multi_spann_replay_invalidations)."""

    DEFAULT_BACKING_FILE_SIZE = 8192

    def __init__(self, base_directory, backing_file_size=DEFAULT_BACKING_FILE_SIZE):
        """InvalidatedIdsStorage::new :32-43 (no file is created before the first record)."""
        self.base_directory = base_directory
        self.backing_file_size = backing_file_size // BYTES_PER_INVALIDATION * BYTES_PER_INVALIDATION
        self.num_files = 0
        self.current_backing_id = -1
        self.current_offset = self.backing_file_size

    @staticmethod
    def _suffix(name):
        """the sort key of :73-79: the text behind the last dot parsed as u32, 0 when it does not parse"""
        tail = name.rsplit(".", 1)[1] if "." in name else ""
        if tail[:1] == "+":                       # u32::from_str accepts one leading '+'
            tail = tail[1:]
        if tail.isascii() and tail.isdigit() and int(tail) < (1 << 32):
            return int(tail)
        return 0

    @classmethod
    def read(cls, base_directory):
        """InvalidatedIdsStorage::read :45-106: a missing directory is created; the files are ordered by numeric suffix
        (a stable sort over the directory listing — sorted by name here so that the result does not depend on the file
        system's order); one file => backing size max(8192, its size), several => the first file's size."""
        import os
        if not os.path.isdir(base_directory):
            os.makedirs(base_directory, exist_ok=True)
            return cls(base_directory)
        names = sorted(n for n in os.listdir(base_directory) if n.startswith("invalidated_ids.bin."))
        if not names:
            return cls(base_directory)
        names.sort(key=cls._suffix)
        size = lambda n: os.path.getsize(os.path.join(base_directory, n))
        first = size(names[0])
        st = cls(base_directory, max(cls.DEFAULT_BACKING_FILE_SIZE, first) if len(names) == 1 else first)
        st.num_files = len(names)
        st.current_backing_id = len(names) - 1
        st.current_offset = size(names[-1])
        return st

    def _path(self, i):
        import os
        return os.path.join(self.base_directory, "invalidated_ids.bin.%d" % i)

    def _new_backing_file(self):
        self.current_backing_id += 1
        open(self._path(self.current_backing_id), "ab").close()
        self.num_files += 1
        self.current_offset = 0

    def invalidate(self, user_id, doc_id):
        """:120-143"""
        self.invalidate_batch([(user_id, doc_id)])

    def invalidate_batch(self, pairs):
        """:147-181: records fill the current file to exactly backing_file_size, then the next file is opened."""
        buf = bytearray()
        def flush():
            if buf:
                with open(self._path(self.current_backing_id), "ab") as f:
                    f.write(buf)
                buf.clear()
        for user_id, doc_id in pairs:
            if self.current_offset == self.backing_file_size:
                flush()
                self._new_backing_file()
            buf += u128_bytes(int(user_id)) + u128_bytes(int(doc_id))
            self.current_offset += BYTES_PER_INVALIDATION
        flush()

    def record_bytes(self):
        """What InvalidatedIdsIterator yields (:183-196, 217-256), as one byte string of 32-byte records: files
        `invalidated_ids.bin.0 .. num_files-1` BY INDEX (a missing one is skipped, `.ok()`), each read to its end; a file
        that ends inside a record is the iterator's panic ("Incomplete invalidation record at end of file")."""
        import os
        out = bytearray()
        for i in range(self.num_files):
            if not os.path.isfile(self._path(i)):
                continue
            with open(self._path(i), "rb") as f:
                data = f.read()
            if len(data) % BYTES_PER_INVALIDATION:
                raise ValueError("Incomplete invalidation record at end of file %s" % self._path(i))
            out += data
        return bytes(out)

    def __iter__(self):
        raw = self.record_bytes()
        for o in range(0, len(raw), BYTES_PER_INVALIDATION):
            yield (int.from_bytes(raw[o:o + 16], "little"), int.from_bytes(raw[o + 16:o + 32], "little"))

    def num_entries(self):
        """:202-210"""
        if self.current_backing_id == -1:
            return 0
        return (self.current_offset + self.current_backing_id * self.backing_file_size) // BYTES_PER_INVALIDATION


# --------------------------------------------------------------------------- segment directory
def write_segment(directory, cat, num_features, pq=None, reassigned=None, invalidated=None, backing_file_size=None):
    """One (multi-user) SPANN segment as the reference lays it out on disk (multi_spann/writer.rs:82-298; SURVEY.md
    Appendix A) from concat_multi_spann's result: the reference's readers (MultiSpannReader::read, multi_spann/reader.rs:35)
    open this tree.  pq = (dimension, subvector_dimension, num_bits) when the posting lists hold PQ codes (cat["codebook"]).
    reassigned = {user_id: u32 [n] old -> new point id} for users whose IVF was reindexed (muopdb_amd.build.reindex): written as
    `reassigned_mappings.<user_id>`, 4 little-endian bytes per vector (ivf/writer.rs:52-66, moved to the top level by
    multi_spann/writer.rs:264-273).  invalidated = [(user_id, doc_id)] already deleted from the segment: appended to the
    tombstone log `invalidated_ids_storage/` (InvalidatedIdsStorage, files of `backing_file_size` bytes — 8192 by default) which
    the open replays; without it the directory is created empty, as is bloom_filter/ (delete path only)."""
    import os
    def put(rel, data):
        path = os.path.join(directory, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(data if isinstance(data, (bytes, bytearray)) else data.encode())
    put("user_index_info", user_index_info_table(cat["user_table"]))
    put("centroids/quantizer/no_op_quantizer_config.yaml", no_op_quantizer_config_yaml(num_features))
    put("centroids/hnsw/index", cat["hnsw_index"])
    put("centroids/hnsw/vector_storage", cat["hnsw_vectors"])
    if pq is None:
        put("ivf/quantizer/no_op_quantizer_config.yaml", no_op_quantizer_config_yaml(num_features))
    else:
        put("ivf/quantizer/product_quantizer_config.yaml", product_quantizer_config_yaml(*pq))
        put("ivf/quantizer/codebook", cat.get("codebook", b""))
    put("ivf/index", cat["ivf_index"])
    put("ivf/vectors", cat["ivf_vectors"])
    put("ivf/raw_vectors", cat.get("ivf_raw_vectors", b""))
    for user_id, mapping in (reassigned or {}).items():
        put("reassigned_mappings.%d" % int(user_id), np.ascontiguousarray(mapping, dtype="<u4").tobytes())
    for sub in ("bloom_filter", "invalidated_ids_storage"):
        os.makedirs(os.path.join(directory, sub), exist_ok=True)
    if invalidated:
        log = InvalidatedIdsStorage(os.path.join(directory, "invalidated_ids_storage"),
                                    backing_file_size or InvalidatedIdsStorage.DEFAULT_BACKING_FILE_SIZE)
        log.invalidate_batch(invalidated)


def read_segment(directory):
    """The files of a segment directory as write_segment / the reference's MultiSpannWriter leave them:
    dict(user_table (flat records), hnsw_index, hnsw_vectors, ivf_index, ivf_vectors, num_features, pq, codebook, reassigned,
    invalidated = [(user_id, doc_id)] of the tombstone log in replay order)."""
    import os
    def get(rel):
        with open(os.path.join(directory, rel), "rb") as f:
            return f.read()
    out = dict(user_table=user_table_from_odht(get("user_index_info")), hnsw_index=get("centroids/hnsw/index"),
               hnsw_vectors=get("centroids/hnsw/vector_storage"), ivf_index=get("ivf/index"), ivf_vectors=get("ivf/vectors"))
    out["num_features"] = parse_simple_yaml(get("centroids/quantizer/no_op_quantizer_config.yaml").decode())["dimension"]
    pq_path = os.path.join(directory, "ivf/quantizer/product_quantizer_config.yaml")
    out["pq"], out["codebook"] = None, None
    if os.path.exists(pq_path):
        y = parse_simple_yaml(get("ivf/quantizer/product_quantizer_config.yaml").decode())
        out["pq"] = (y["dimension"], y["subvector_dimension"], y["num_bits"])
        out["codebook"] = np.frombuffer(get("ivf/quantizer/codebook"), np.float32)
    out["reassigned"] = {int(name.split(".", 1)[1]): np.frombuffer(get(name), "<u4")
                         for name in os.listdir(directory) if name.startswith("reassigned_mappings.")}
    out["invalidated"] = list(InvalidatedIdsStorage.read(os.path.join(directory, "invalidated_ids_storage")))
    return out
