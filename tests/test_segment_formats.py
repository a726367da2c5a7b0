"""CPU tests of the segment-level producers: the odht `user_index_info` table and the on-disk
segment tree of Appendix A.

odht 0.3.1 is a third-party crate ABSENT from /root/reference (Cargo.lock pins it; multi_spann/writer.rs:253-259 and
user_index_info.rs:84-140 are its only call sites, and the reference's tests at that boundary are end to end: K9 / K10).
muopdb_amd.formats restates its published layout; these tests pin the restatement against a HAND-computed table and against
the library's own reader (mdb_odht_user_table) — they cannot pin it against the crate itself: **parity unpinned** there."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from muopdb_amd import formats as F
from muopdb_amd import lib as L

K = 0x9E3779B9
M32 = 0xFFFFFFFF


def test_fx_hash_by_hand():
    # FxHashFn over a u128 key = four LE u32 words; h <- (rotl(h, 5) ^ w) * 0x9e3779b9 mod 2^32
    assert F.fx_hash32((0).to_bytes(16, "little")) == 0                       # (0 ^ 0) * K four times
    h = (0 ^ 1) * K & M32                                                     # word 0 = 1
    assert h == 0x9E3779B9
    for _ in range(3):                                                        # words 1..3 = 0
        h = ((((h << 5) | (h >> 27)) & M32) ^ 0) * K & M32
    assert h == 0x4D286184 == F.fx_hash32((1).to_bytes(16, "little"))
    # a 7-byte key: one u32 word (1), then a u16 (2), then a u8 (3)
    rot = lambda v: ((v << 5) | (v >> 27)) & M32
    h = (rot(0) ^ 1) * K & M32
    h = (rot(h) ^ 2) * K & M32
    h = (rot(h) ^ 3) * K & M32
    assert F.fx_hash32(b"\x01\x00\x00\x00\x02\x00\x03") == h


def test_odht_two_entry_table_by_hand():
    """users 0 and 1, with_capacity(2, 90): factor = 65535 * 90 / 100 = 58981, ceil(2 * 65535 / 58981) = 3 -> 4 -> max(.., 16)
    = 16 slots.  key 0: hash 0 -> control byte 0, slot 0 (mirrored at 16).  key 1: hash 0x4d286184 -> control byte
    0x4d286184 >> 25 = 38, slot 0x4d286184 & 15 = 4 (mirrored at 20)."""
    r0 = F.pack_user_index_info(0, centroid_vector_offset=7, ivf_index_len=99)
    r1 = F.pack_user_index_info(1, centroid_vector_offset=8, ivf_pq_codebook_len=5)
    t = F.user_index_info_table(r0 + r1)
    assert len(t) == 32 + 16 * 128 + 16 + 16
    assert t[:32] == b"ODHT" + bytes([1, 16, 112, 32]) + struct.pack("<QQ", 2, 16) + bytes([0, 0, 0, 2]) + struct.pack("<H", 58981) + b"\0\0"
    entries, meta = t[32:32 + 16 * 128], t[32 + 16 * 128:]
    assert entries[0:16] == (0).to_bytes(16, "little") and entries[16:128] == r0
    assert entries[4 * 128:4 * 128 + 16] == (1).to_bytes(16, "little") and entries[4 * 128 + 16:5 * 128] == r1
    assert all(b == 0 for i in range(16) if i not in (0, 4) for b in entries[i * 128:(i + 1) * 128])
    want = [0xFF] * 32
    want[0] = want[16] = 0
    want[4] = want[20] = 38
    assert list(meta) == want
    assert F.user_table_from_odht(t) == r0 + r1 and F.odht_get(t, (1).to_bytes(16, "little")) == r1
    assert F.odht_get(t, (2).to_bytes(16, "little")) is None


def test_odht_three_entry_collision_chain_by_hand():
    """users 5, 27 and 28 all hash into start slot 15 of a 16-slot table (0x6337b94f, 0x73c076ff, 0xeb7546ff: & 15 = 15) — one
    collision chain.  odht's insert (raw_table.rs, RawTableMut::insert -> find_or_insert) scans the 16-byte GROUP that starts at the
    key's slot — bytes 15, 16, .. 30 of the metadata array, i.e. slot 15 followed by the MIRROR of slots 0 .. 14 — and takes the
    first control byte with bit 7 set: 15, then 0 (seen through mirror byte 16), then 1.  Control bytes = hash >> 25 = 49, 57, 117;
    every slot below 16 is written twice (slot and mirror).  Published semantics this pins (the crate itself is absent here:
    parity unpinned against it — see the module docstring):
        | what                               | value                                                                        |
        | empty control byte                 | bit 7 set: lookups take the group's sign mask (`match_empty` = movemask), so  |
        |                                    | 0x80 and 0xFF both read as empty; an occupied slot holds h2 = top 7 hash bits |
        | group                              | 16 consecutive control bytes from the key's slot, no alignment, no wrap: the  |
        |                                    | first 16 bytes are repeated after the last slot                               |
        | what from_raw_bytes checks         | tag "ODHT", the four size bytes against the Config, format version [0,0,0,2], |
        |                                    | slot_count a power of two, len == 32 + slots * 128 + slots + 16 — never a     |
        |                                    | control byte: a table the WRITER below emits is accepted whichever of the two |
        |                                    | empty encodings the crate itself would have chosen                            |"""
    recs = [F.pack_user_index_info(u, centroid_vector_offset=u) for u in (5, 27, 28)]
    for u, h in zip((5, 27, 28), (0x6337B94F, 0x73C076FF, 0xEB7546FF)):
        assert F.fx_hash32(u.to_bytes(16, "little")) == h and h & 15 == 15
    t = F.user_index_info_table(b"".join(recs))
    assert len(t) == 32 + 16 * 128 + 32 and struct.unpack_from("<QQ", t, 8) == (3, 16)
    entries, meta = t[32:32 + 16 * 128], t[32 + 16 * 128:]
    for slot, u, rec in ((15, 5, recs[0]), (0, 27, recs[1]), (1, 28, recs[2])):
        assert entries[slot * 128:slot * 128 + 16] == u.to_bytes(16, "little") and entries[slot * 128 + 16:(slot + 1) * 128] == rec
    want = [0xFF] * 32
    want[15] = want[31] = 0x6337B94F >> 25
    want[0] = want[16] = 0x73C076FF >> 25
    want[1] = want[17] = 0xEB7546FF >> 25
    assert list(meta) == want and (want[15], want[0], want[1]) == (49, 57, 117)
    for u, rec in zip((5, 27, 28), recs):                                     # the chain is walked on lookup
        assert F.odht_get(t, u.to_bytes(16, "little")) == rec
    assert F.odht_get(t, (53).to_bytes(16, "little")) is None                 # same start slot, not present: stops at the first empty
    # a writer that fills with 0x80 instead (the other encoding a bit-7 reader accepts) reads the same — here and in the library
    t80 = bytearray(t)
    for i, b in enumerate(meta):
        if b == 0xFF:
            t80[32 + 16 * 128 + i] = 0x80
    assert F.user_table_from_odht(bytes(t80)) == F.user_table_from_odht(t)
    lib = L.load()
    for raw in (t, bytes(t80)):
        buf = np.frombuffer(raw, np.uint8)
        cnt = C.c_size_t()
        out = np.zeros(3 * 112, np.uint8)
        assert lib.mdb_odht_user_table(L.ptr(buf, C.c_uint8), C.c_size_t(buf.size), L.ptr(out, C.c_uint8), C.c_size_t(3), C.byref(cnt)) == 0
        assert cnt.value == 3 and out.tobytes() == F.user_table_from_odht(t)


def test_odht_probing_wrap_and_round_trip_and_native_reader():
    rng = np.random.default_rng(4)
    for n in (1, 3, 14, 15, 33, 200, 1024):
        ids = sorted({int(x) for x in rng.integers(0, 1 << 62, n)} | ({(1 << 100) + 5} if n > 2 else set()))
        recs = b"".join(F.pack_user_index_info(u, centroid_index_offset=i * 16, ivf_vectors_len=i) for i, u in enumerate(ids))
        t = F.user_index_info_table(recs)
        slots, _ = F.odht_slots_needed(len(ids))
        assert len(t) == 32 + slots * 129 + 16 and slots >= 16 and slots & (slots - 1) == 0 and len(ids) * 100 <= slots * 90 + 99
        assert t[32 + slots * 128:32 + slots * 128 + 16] == t[32 + slots * 129:]           # the mirrored first group
        assert F.user_table_from_odht(t) == recs
        for i, u in enumerate(ids):
            assert F.odht_get(t, u.to_bytes(16, "little")) == recs[i * 112:(i + 1) * 112]
        # the library's reader (host-only entry point of the C ABI)
        lib = L.load()
        buf = np.frombuffer(t, np.uint8)
        cnt = C.c_size_t()
        assert lib.mdb_odht_user_table(L.ptr(buf, C.c_uint8), C.c_size_t(buf.size), None, C.c_size_t(0), C.byref(cnt)) == 0
        assert cnt.value == len(ids)
        users = (L.UserIndexInfoC * len(ids))()
        assert lib.mdb_odht_user_table(L.ptr(buf, C.c_uint8), C.c_size_t(buf.size), users, C.c_size_t(len(ids)), C.byref(cnt)) == 0
        assert C.string_at(users, len(ids) * 112) == recs
        # empty control bytes are recognised by bit 7 (odht's group query = movemask of the control bytes): the same table with
        # its empties written as 0x80 instead of 0xFF reads identically through both readers
        alt = bytearray(t)
        for i in range(32 + slots * 128, len(alt)):
            if alt[i] == 0xFF:
                alt[i] = 0x80
        alt = bytes(alt)
        assert F.user_table_from_odht(alt) == recs and F.odht_get(alt, ids[0].to_bytes(16, "little")) == recs[:112]
        buf2 = np.frombuffer(alt, np.uint8)
        assert lib.mdb_odht_user_table(L.ptr(buf2, C.c_uint8), C.c_size_t(buf2.size), users, C.c_size_t(len(ids)), C.byref(cnt)) == 0
        assert C.string_at(users, len(ids) * 112) == recs
    # a full group forces the triangular probe: 17 keys whose hashes share the low 4 bits land in two groups
    same = []
    x = 0
    while len(same) < 17:
        x += 1
        if F.fx_hash32(x.to_bytes(16, "little")) & 31 == 7:
            same.append(x)
    recs = b"".join(F.pack_user_index_info(u, ivf_index_len=u) for u in same)
    t = F.user_index_info_table(recs)
    assert F.odht_slots_needed(17)[0] == 32
    assert F.user_table_from_odht(t) == b"".join(sorted((recs[i:i + 112] for i in range(0, len(recs), 112)), key=lambda r: int.from_bytes(r[:16], "little")))
    for u in same:
        assert F.odht_get(t, u.to_bytes(16, "little"))[:16] == u.to_bytes(16, "little")
    for bad in (t[:31], b"XDHT" + t[4:], t[:-1], t[:5] + bytes([8]) + t[6:], t[:8] + struct.pack("<Q", 3) + t[16:]):
        buf = np.frombuffer(bad, np.uint8)
        assert lib.mdb_odht_user_table(L.ptr(buf, C.c_uint8), C.c_size_t(buf.size), None, C.c_size_t(0), C.byref(cnt)) == 2  # MDB_ERR_FORMAT


def test_write_segment_tree_and_read_back(tmp_path):
    """the files a MultiSpannReader opens, at the paths MultiSpannWriter writes them."""
    users = {}
    for u in (3, 9, (1 << 90) + 2):
        v = np.arange(20 * 4, dtype=np.float32).reshape(20, 4) + (u & 0xFF)
        cent = v[:2].copy()
        ivf = F.write_ivf_index(cent, list(range(20)), [np.arange(0, 10, dtype=np.uint64), np.arange(10, 20, dtype=np.uint64)])
        hn = F.write_hnsw_index([{0: [1], 1: [0]}], [0, 1], 4)
        users[u] = dict(hnsw_index=hn, hnsw_vectors=F.write_vector_file(cent), ivf_index=ivf, ivf_vectors=F.write_vector_file(v),
                        ivf_raw_vectors=F.write_vector_file(v))
    cat = F.concat_multi_spann(users)
    seg = str(tmp_path / "segment")
    mapping = {3: np.asarray([2, 0, 1] + list(range(3, 20)), np.uint32), (1 << 90) + 2: np.arange(20, dtype=np.uint32)[::-1]}
    F.write_segment(seg, cat, 4, reassigned=mapping)
    for rel in ("user_index_info", "centroids/quantizer/no_op_quantizer_config.yaml", "centroids/hnsw/index", "centroids/hnsw/vector_storage",
                "ivf/quantizer/no_op_quantizer_config.yaml", "ivf/index", "ivf/vectors", "ivf/raw_vectors"):
        assert os.path.isfile(os.path.join(seg, rel)), rel
    assert open(os.path.join(seg, "centroids/quantizer/no_op_quantizer_config.yaml")).read() == "dimension: 4\n"
    back = F.read_segment(seg)
    assert back["user_table"] == cat["user_table"] and back["num_features"] == 4 and back["pq"] is None
    for kf in ("hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors"):
        assert back[kf] == cat[kf]
    # reassigned_mappings.<user_id>: 4 LE bytes per vector at the segment's top level (ivf/writer.rs:52-66, multi_spann/writer.rs:264-273)
    assert open(os.path.join(seg, "reassigned_mappings.3"), "rb").read()[:12] == struct.pack("<III", 2, 0, 1)
    assert sorted(back["reassigned"]) == sorted(mapping)
    for u, m in mapping.items():
        assert np.array_equal(back["reassigned"][u], m)
    # PQ variant: product_quantizer_config.yaml + codebook instead of the no-op config
    cat["codebook"] = np.arange(2 * 4 * 2, dtype=np.float32).tobytes()
    seg2 = str(tmp_path / "segment_pq")
    F.write_segment(seg2, cat, 4, pq=(4, 2, 2))
    back = F.read_segment(seg2)
    assert back["pq"] == (4, 2, 2) and back["codebook"].tobytes() == cat["codebook"]
    assert not os.path.exists(os.path.join(seg2, "ivf/quantizer/no_op_quantizer_config.yaml"))


# --------------------------------------------------------------------------------------- tombstone log (invalidated_ids.rs)
def _oracle_pairs(directory):
    from oracle.oracle import invalidated_ids_iter
    return list(invalidated_ids_iter(directory))


def test_invalidated_ids_storage_naming_rounding_and_rollover(tmp_path):
    """InvalidatedIdsStorage (rs/index/src/ivf/files/invalidated_ids.rs): 32-byte LE records, the backing size rounded DOWN to
    whole records (:33-35; the reference's own test_invalidate uses 1024), a new file exactly when the current one is full
    (:121-123), names `invalidated_ids.bin.<i>`; the product's reader/writer and the oracle's independent restatement of
    ::read + ::iter agree on the bytes."""
    d = str(tmp_path / "log")
    st = F.InvalidatedIdsStorage(d, 100)                      # 100 -> 96 bytes = 3 records per file
    os.makedirs(d)
    assert st.backing_file_size == 96 and st.num_entries() == 0
    big = (1 << 100) + 7
    pairs = [(0, 5), (big, (1 << 127) + 3), (7, 8), (7, 9), (big, 1), (0, 5), (3, 3)]
    st.invalidate(*pairs[0])
    st.invalidate_batch(pairs[1:])
    assert sorted(os.listdir(d)) == ["invalidated_ids.bin.0", "invalidated_ids.bin.1", "invalidated_ids.bin.2"]
    assert [os.path.getsize(os.path.join(d, "invalidated_ids.bin.%d" % i)) for i in range(3)] == [96, 96, 32]
    raw = open(os.path.join(d, "invalidated_ids.bin.0"), "rb").read()
    assert raw[:32] == struct.pack("<QQQQ", 0, 0, 5, 0)       # u128 LE user id, u128 LE doc id
    assert raw[32:64] == struct.pack("<QQQQ", 7, 1 << 36, 3, 1 << 63)
    assert st.num_entries() == 7 and list(st) == pairs == _oracle_pairs(d)
    # ::read on a directory of several files: backing size = the FIRST file's size, offset = the last file's size
    rd = F.InvalidatedIdsStorage.read(d)
    assert (rd.backing_file_size, rd.current_backing_id, rd.current_offset, rd.num_entries()) == (96, 2, 32, 7)
    rd.invalidate_batch([(1, 1), (1, 2), (1, 3)])             # fills file 2, then opens file 3
    assert [os.path.getsize(os.path.join(d, "invalidated_ids.bin.%d" % i)) for i in range(4)] == [96, 96, 96, 32]
    assert list(F.InvalidatedIdsStorage.read(d)) == pairs + [(1, 1), (1, 2), (1, 3)] == _oracle_pairs(d)
    # numeric, not lexicographic, order of the suffixes: 12 files
    d2 = str(tmp_path / "log2")
    os.makedirs(d2)
    st2 = F.InvalidatedIdsStorage(d2, 32)
    many = [(i, 1000 - i) for i in range(12)]
    st2.invalidate_batch(many)
    rd2 = F.InvalidatedIdsStorage.read(d2)
    assert rd2.num_files == 12 and rd2.current_backing_id == 11 and list(rd2) == many == _oracle_pairs(d2)


def test_invalidated_ids_storage_read_edge_cases(tmp_path):
    """::read :45-106 — missing directory is created and empty; ONE file: backing size = max(8192, its size); a file that ends
    inside a record is the iterator's panic; a hole in the numbering is skipped by ::iter (it opens files by INDEX)."""
    d = str(tmp_path / "absent" / "invalidated_ids_storage")
    st = F.InvalidatedIdsStorage.read(d)
    assert os.path.isdir(d) and st.num_entries() == 0 and list(st) == [] == _oracle_pairs(d)
    st.invalidate(9, 10)
    assert os.listdir(d) == ["invalidated_ids.bin.0"] and F.InvalidatedIdsStorage.read(d).backing_file_size == 8192
    with open(os.path.join(d, "invalidated_ids.bin.0"), "ab") as f:
        f.write(b"\0" * (32 * 300))                            # 9632 bytes > 8192: the single file's own size wins
    one = F.InvalidatedIdsStorage.read(d)
    assert one.backing_file_size == 9632 and one.num_entries() == 301
    with open(os.path.join(d, "invalidated_ids.bin.0"), "ab") as f:
        f.write(b"\1" * 5)
    with pytest.raises(ValueError, match="Incomplete invalidation record"):
        list(F.InvalidatedIdsStorage.read(d))
    with pytest.raises(ValueError, match="Incomplete invalidation record"):
        _oracle_pairs(d)
    d3 = str(tmp_path / "holes")
    os.makedirs(d3)
    for i, rec in ((0, (1, 2)), (2, (3, 4))):                 # files .0 and .2: two files => ::iter opens .0 and .1 only
        with open(os.path.join(d3, "invalidated_ids.bin.%d" % i), "wb") as f:
            f.write(F.u128_bytes(rec[0]) + F.u128_bytes(rec[1]))
    assert list(F.InvalidatedIdsStorage.read(d3)) == [(1, 2)] == _oracle_pairs(d3)


def test_write_segment_with_invalidated_ids_round_trips(tmp_path):
    users = {3: dict(hnsw_index=b"H" * 40, hnsw_vectors=b"V" * 24, ivf_index=b"I" * 64, ivf_vectors=b"W" * 16)}
    cat = F.concat_multi_spann(users)
    seg = str(tmp_path / "seg")
    dead = [(3, 11), (99, 1), (3, 11), (3, 12)]
    F.write_segment(seg, cat, 4, invalidated=dead, backing_file_size=64)
    assert sorted(os.listdir(os.path.join(seg, "invalidated_ids_storage"))) == ["invalidated_ids.bin.0", "invalidated_ids.bin.1"]
    assert F.read_segment(seg)["invalidated"] == dead
