"""GPU tests (-m gpu) of the boundary semantics the reference's callers rely on: the planner filter as
a PER-CALL argument (ivf/block_based/index.rs:175-237 takes `planner` per call), ONE resident index shared by concurrent
callers (segment/mod.rs:273-274: immutable index, `Quantizer: Send + Sync`, one query per tokio task), tombstones shared
by every handle over an index (`invalid_point_ids: DashSet`, index.rs:30), asynchronous submit / wait for host-buffer
callers, and malformed inputs refused before a kernel can read out of bounds."""
import struct
import threading

import numpy as np
import pytest

from muopdb_amd import formats as F
from tests import helpers as H
from tests.test_gpu_parity import assert_result_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from muopdb_amd import lib as L
    c = L.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def case(oracle):
    """one SPANN-shaped index (PQ posting lists) + its multi-user concatenation, the oracle twins, queries, two filters"""
    from muopdb_amd.index import ProductQuantizer, allow_bitmap
    rng = np.random.default_rng(77)
    n, d = 4000, 32
    v = H.sift_like(n, d, n_clusters=30, seed=16)
    cb = H.train_pq_codebook(v[:1500], 8, 6, iters=3)
    opq = oracle.ProductQuantizer(d, 8, 6, cb)
    files, cent, _ = H.build_spann_files(oracle, v, list(range(n)), 40, quantize=opq.quantize, max_neighbors=8, max_layers=3,
                                         ef_construction=50)
    q = (v[rng.integers(0, n, 24)] + rng.normal(0, 3, (24, d))).astype(np.float32)
    even = allow_bitmap(np.arange(0, n, 2), n)
    per_query = np.stack([allow_bitmap(rng.choice(n, size=int(rng.integers(1, n)), replace=False), n) for _ in range(24)])
    return dict(n=n, d=d, files=files, q=q, even=even, per_query=per_query, quant=ProductQuantizer(d, 8, 6, cb),
                oquant=oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 6, cb))


def test_per_call_filter_equals_oracle_and_leaves_no_state(ctx, oracle, case):
    from muopdb_amd.index import BlockBasedIvf, MultiSpannIndex, SearchParams, Spann
    f, q = case["files"], case["q"]
    g = BlockBasedIvf(ctx, f["ivf_index"], f["ivf_vectors"], case["quant"])
    o = oracle.BlockBasedIvf(f["ivf_index"], f["ivf_vectors"], case["oquant"])
    plain = o.search(q, 10, num_probes=12)
    for bm in (case["even"], case["per_query"]):
        with oracle.planner_filter(bm):
            want = o.search(q, 10, num_probes=12)
        assert_result_rows(g.search(q, 10, 12, planner=bm), want, len(q))
        assert_result_rows(g.search(q, 10, 12), plain, len(q))          # the filter was an argument, not state
    probes = g.find_nearest_centroids(q, 12)
    with oracle.planner_filter(case["even"]):
        want = o.search(q, 10, num_probes=12)
    assert_result_rows(g.search_with_centroids_and_remap(q, probes, 10, planner=case["even"]), want, len(q))
    sp = Spann(ctx, f["hnsw_index"], f["hnsw_vectors"], f["ivf_index"], f["ivf_vectors"], case["quant"])
    osp = oracle.Spann(f["hnsw_index"], f["hnsw_vectors"], f["ivf_index"], f["ivf_vectors"], case["oquant"])
    p, op = SearchParams(10, 50).with_num_explored_centroids(6), oracle.SearchParams(10, 50, num_explored_centroids=6)
    with oracle.planner_filter(case["per_query"]):
        want = osp.search(q, op)
    assert_result_rows(sp.search(q, p, planner=case["per_query"]), want, len(q))
    assert_result_rows(sp.search(q, p), osp.search(q, op), len(q))
    cat = F.concat_multi_spann({5: f})
    margs = (cat["user_table"], case["d"], cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    ms = MultiSpannIndex(ctx, *margs, case["quant"])
    oms = oracle.MultiSpannIndex(*margs, case["oquant"])
    with oracle.planner_filter(case["even"]):
        want = oms.search_for_user([5] * len(q), q, op)
    got = ms.search_for_user([5] * len(q), q, p, planner=case["even"])
    assert_result_rows(got, want, len(q))
    assert all(x % 2 == 0 for i in range(len(q)) for x in got.doc_ids(i))


def test_filter_bitmaps_are_validated(ctx, case):
    """a bitmap shorter than ceil(num_vectors / 32) words, or fewer per-query bitmaps than queries, used to be
    read out of bounds by the scan; now both are MDB_ERR_INVALID_ARG (the planner is a per-call argument: the stateful
    mdb_*_set_filter entries were removed in round 4)."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf
    f, q = case["files"], case["q"]
    g = BlockBasedIvf(ctx, f["ivf_index"], f["ivf_vectors"], case["quant"])
    short = case["even"][:-1]
    for bad in (short, case["per_query"][:5], np.zeros((0,), np.uint32)):
        with pytest.raises(L.MuopdbError) as e:
            g.search(q, 10, 12, planner=bad)
        assert e.value.status == L.MDB_ERR_INVALID_ARG
    g.search(q[:5], 10, 12, planner=case["per_query"][:5])   # a batch the bitmaps cover is served
    g.search(q, 10, 12, planner=case["per_query"])


def test_two_threads_with_different_filters_on_one_resident_index(ctx, oracle, case):
    """two host threads, DIFFERENT planner filters, ONE resident index (attached handles, each
    on its own context / stream): no handle-global state to race on, every row equals the oracle's for that thread's filter."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf
    f, q = case["files"], case["q"]
    g = BlockBasedIvf(ctx, f["ivf_index"], f["ivf_vectors"], case["quant"])
    o = oracle.BlockBasedIvf(f["ivf_index"], f["ivf_vectors"], case["oquant"])
    filters = [case["even"], case["per_query"], None]
    wants = []
    for bm in filters:
        if bm is None:
            wants.append(o.search(q, 10, num_probes=12))
        else:
            with oracle.planner_filter(bm):
                wants.append(o.search(q, 10, num_probes=12))
    ctxs = [L.Context(0) for _ in filters]
    handles = [g.attach(c) for c in ctxs]
    errors = []

    def worker(i):
        try:
            for _ in range(25):
                assert_result_rows(handles[i].search(q, 10, 12, planner=filters[i]), wants[i], len(q))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(filters))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # the SAME handle from two threads with different filters: calls serialise on the context, filters still do not mix
    errors.clear()
    handles[2] = handles[0]
    ts = [threading.Thread(target=worker, args=(i,)) for i in (0, 2)]
    filters[2], wants[2] = filters[1], wants[1]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for h in handles[:2]:
        h.close()
    for c in ctxs:
        c.close()


def test_attached_handles_share_index_and_tombstones(ctx, oracle, case):
    """mdb_{ivf,spann,multi_spann}_attach: same rows through every handle; a tombstone set through one handle is seen by all
    (one `invalid_point_ids` per index); the memory outlives the first handle (freed in any order)."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf, MultiSpannIndex, SearchParams, Spann
    f, q = case["files"], case["q"]
    c2, c3 = L.Context(0), L.Context(0)
    g = BlockBasedIvf(ctx, f["ivf_index"], f["ivf_vectors"], case["quant"])
    o = oracle.BlockBasedIvf(f["ivf_index"], f["ivf_vectors"], case["oquant"])
    a, b = g.attach(c2), g.attach(c3)
    want = o.search(q, 10, num_probes=12)
    for h in (g, a, b):
        assert_result_rows(h.search(q, 10, 12), want, len(q))
    victim = want.doc_ids(0)[0]
    assert a.invalidate(victim) and o.invalidate(victim)       # through an ATTACHED handle
    assert g.is_invalidated(victim) and b.is_invalidated(victim) and not b.invalidate(victim)
    want = o.search(q, 10, num_probes=12)
    g.close()                                                   # the owner first: the arrays live on with the views
    for h in (a, b):
        assert_result_rows(h.search(q, 10, 12), want, len(q))
    a.close(); b.close()
    sp = Spann(ctx, f["hnsw_index"], f["hnsw_vectors"], f["ivf_index"], f["ivf_vectors"], case["quant"])
    osp = oracle.Spann(f["hnsw_index"], f["hnsw_vectors"], f["ivf_index"], f["ivf_vectors"], case["oquant"])
    p, op = SearchParams(10, 50).with_num_explored_centroids(6), oracle.SearchParams(10, 50, num_explored_centroids=6)
    sa = sp.attach(c2)
    assert_result_rows(sa.search(q, p), osp.search(q, op), len(q))
    assert sa.invalidate(osp.search(q, op).doc_ids(1)[0]) and osp.invalidate(osp.search(q, op).doc_ids(1)[0])
    assert_result_rows(sp.search(q, p), osp.search(q, op), len(q))
    cat = F.concat_multi_spann({5: f, 9: f})
    margs = (cat["user_table"], case["d"], cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
    ms = MultiSpannIndex(ctx, *margs, case["quant"])
    oms = oracle.MultiSpannIndex(*margs, case["oquant"])
    ma = ms.attach(c3)
    users = [5 if i % 2 else 9 for i in range(len(q))]
    assert_result_rows(ma.search_for_user(users, q, p), oms.search_for_user(users, q, op), len(q))
    ms.close()
    assert_result_rows(ma.search_for_user(users, q, p), oms.search_for_user(users, q, op), len(q))
    ma.close(); sa.close(); sp.close()
    c2.close(); c3.close()


def test_submit_wait_equals_synchronous_calls(ctx, oracle, case):
    """mdb_*_search_submit / mdb_wait: the call returns after enqueueing (inputs are reusable at once), the outputs are
    filled by mdb_wait; several batches in flight = one context + attached handle each."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedHnsw, BlockBasedIvf, MultiSpannIndex, SearchParams
    f, q = case["files"], case["q"]
    g = BlockBasedIvf(ctx, f["ivf_index"], f["ivf_vectors"], case["quant"])
    want = g.search(q, 10, 12, planner=case["even"])
    ctxs = [L.Context(0) for _ in range(3)]
    hs = [g.attach(c) for c in ctxs]
    qs = [q.copy() for _ in hs]
    pend = [h.search_submit(qq, 10, 12, planner=case["even"]) for h, qq in zip(hs, qs)]
    for qq in qs:
        qq[:] = 0                                   # the inputs were staged at submit time
    for p_ in pend:
        assert_result_rows(p_.wait(), want, len(q))
    assert pend[0].done()
    with pytest.raises(L.MuopdbError):              # one pending call per context
        pd = hs[0].search_submit(q, 10, 12)
        try:
            hs[0].search(q, 10, 12)
        finally:
            pd.wait()
    v = np.random.default_rng(3).standard_normal((1200, 24)).astype(np.float32)
    hidx, hvec = H.build_hnsw_files(oracle, v, list(range(1200)), max_neighbors=10, max_layers=3, ef_construction=60)
    hn = BlockBasedHnsw(ctx, hidx, hvec, 24)
    hq = v[:17] + 0.01
    assert_result_rows(hn.ann_search_submit(hq, 5, 40).wait(), hn.ann_search(hq, 5, 40), len(hq))
    bad = hq.copy(); bad[3, 2] = np.nan             # a deferred error surfaces at wait()
    pd = hn.ann_search_submit(bad, 5, 40)
    with pytest.raises(L.MuopdbError) as e:
        pd.wait()
    assert e.value.status == 5  # MDB_ERR_NAN
    assert_result_rows(hn.ann_search(hq, 5, 40), hn.ann_search_submit(hq, 5, 40).wait(), len(hq))
    cat = F.concat_multi_spann({5: f})
    ms = MultiSpannIndex(ctx, cat["user_table"], case["d"], cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"],
                         case["quant"])
    p = SearchParams(10, 50).with_num_explored_centroids(6)
    assert_result_rows(ms.search_for_user_submit([5] * len(q), q, p, planner=case["per_query"]).wait(),
                       ms.search_for_user([5] * len(q), q, p, planner=case["per_query"]), len(q))
    for h in hs:
        h.close()
    for c in ctxs:
        c.close()


def test_corrupt_posting_list_headers_are_refused(ctx, case):
    """the Elias-Fano decoder trusted the on-disk list header (lower_bit_length, word counts); a corrupt or
    truncated index made the device read past the blob.  Now validated on the host before upload: MDB_ERR_FORMAT."""
    from muopdb_amd import lib as L
    from muopdb_amd.index import BlockBasedIvf
    good = F.ef_encode(np.array([5, 8, 8, 15, 32, 1000, 5000], np.uint64))
    assert list(ctx.ef_decode(good)) == [5, 8, 8, 15, 32, 1000, 5000]
    n, Lb, lw, uw = struct.unpack("<QQQQ", good[:32])

    def hdr(n_=n, L_=Lb, lw_=lw, uw_=uw):
        return struct.pack("<QQQQ", n_, L_, lw_, uw_) + good[32:]

    for blob in (hdr(L_=64), hdr(L_=200), hdr(lw_=0) if Lb else hdr(uw_=0), hdr(uw_=0), hdr(lw_=(1 << 61)), hdr(uw_=(1 << 62)),
                 hdr(n_=1 << 40), good[:40], good[:16]):
        with pytest.raises(L.MuopdbError) as e:
            ctx.ef_decode(blob)
        assert e.value.status == 2, blob[:32]
    # the same inside an IVF index file: corrupt one list's header in place
    f = case["files"]
    idx = bytearray(f["ivf_index"])
    nf, qd, ncl = struct.unpack_from("<III", idx, 1)
    nv, doc_len, cent_len = struct.unpack_from("<QQQ", idx, 13)
    meta = (((48 + doc_len + 7) // 8 * 8) + cent_len + 7) // 8 * 8
    npl = struct.unpack_from("<Q", idx, meta)[0]
    pl0 = meta + 8 + npl * 16 + struct.unpack_from("<Q", idx, meta + 8 + 8)[0]
    for field, value in ((8, 77), (16, 1 << 60), (24, 0), (0, 1 << 33)):
        bad = bytearray(idx)
        struct.pack_into("<Q", bad, pl0 + field, value)
        with pytest.raises(L.MuopdbError) as e:
            BlockBasedIvf(ctx, bytes(bad), f["ivf_vectors"], case["quant"])
        assert e.value.status == 2
    bad = bytearray(idx)
    struct.pack_into("<Q", bad, 21, (1 << 63))          # doc_id_mapping_len that wraps the section offsets
    with pytest.raises(L.MuopdbError) as e:
        BlockBasedIvf(ctx, bytes(bad), f["ivf_vectors"], case["quant"])
    assert e.value.status == 2


def test_pending_segment_over_fetch_and_snapshot_merge(ctx, oracle, case):
    """PendingSegment::search_with_id's over-fetch (segment/pending_segment.rs:300-318: top_k + len(invalidated)) and
    Snapshot::search_for_user's cross-segment merge (collection/snapshot.rs:69-110) over GPU-resident segments, against
    the same procedure over the oracle's segments."""
    from muopdb_amd.index import MultiSpannIndex, PendingSegment, SearchParams, Snapshot
    f, q = case["files"], case["q"]
    n = case["n"]
    v2 = H.sift_like(1500, case["d"], n_clusters=12, seed=77)
    opq = oracle.ProductQuantizer(case["d"], 8, 6, case["quant"].codebook)
    f2, _, _ = H.build_spann_files(oracle, v2, list(range(100_000, 101_500)), 15, quantize=opq.quantize, max_neighbors=8, max_layers=3,
                                   ef_construction=50)
    cats = [F.concat_multi_spann({7: f}), F.concat_multi_spann({7: f2, 8: f2})]
    segs, osegs = [], []
    for cat in cats:
        a = (cat["user_table"], case["d"], cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
        segs.append(MultiSpannIndex(ctx, *a, case["quant"]))
        osegs.append(oracle.MultiSpannIndex(*a, case["oquant"]))
    p = SearchParams(5, 50).with_num_explored_centroids(6)
    first = PendingSegment(segs[:1]).search_with_id(7, q[0], p)
    dead = {7: [first[0][0], first[2][0]]}                       # two of the top documents are temporarily invalidated
    pend = PendingSegment(segs[:1], dead)
    got = pend.search_with_id(7, q[0], p)
    op = oracle.SearchParams(5 + 2, 50, num_explored_centroids=6)
    ores = osegs[0].search_for_user([7], q[:1], op)
    want = [r for r in ores.id_with_scores(0) if r[0] not in dead[7]]   # concatenated, neither re-sorted nor truncated (:324-333)
    assert got == want and len(got) == 5 and all(r[0] not in dead[7] for r in got)
    assert got[:3] == [r for r in first if r[0] not in dead[7]][:3]     # the over-fetch refills the row instead of shortening it
    assert PendingSegment(segs[:1]).search_with_id(12345, q[0], p) == []   # Some(empty): no inner segment knows the user (:333)
    snap = Snapshot([pend, segs[1]])
    rows = snap.search_for_user(7, q[1], p)
    o1 = osegs[0].search_for_user([7], q[1:2], oracle.SearchParams(7, 50, num_explored_centroids=6))
    o2 = osegs[1].search_for_user([7], q[1:2], oracle.SearchParams(5, 50, num_explored_centroids=6))
    merged = sorted([r for r in o1.id_with_scores(0) if r[0] not in dead[7]] + o2.id_with_scores(0), key=lambda r: (r[1], r[0]))[:5]
    assert rows == merged
    many = snap.search_for_users([7, 8, 999], q[1], p)
    o3 = osegs[1].search_for_user([8], q[1:2], oracle.SearchParams(5, 50, num_explored_centroids=6))
    assert many == sorted(merged + o3.id_with_scores(0), key=lambda r: (r[1], r[0]))[:5]


def test_cpp_pending_segment_and_snapshot_match_python(ctx, oracle, tmp_path):
    """include/muopdb_host.hpp's PendingSegment / Snapshot (the compiled host logic a Rust or C++ host links; segment/
    pending_segment.rs:285-335, collection/snapshot.rs:39-110) through examples/host_mirror_demo.cpp `segments`: its rows must
    equal the Python mirror's — which the test above checks against the oracle — incl. the over-fetch around two temporarily
    invalidated documents, a user one segment does not know and an unknown user in search_for_users."""
    import os
    import struct
    import subprocess
    from muopdb_amd.index import MultiSpannIndex, PendingSegment, SearchParams, Snapshot
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "muopdb_amd", "host_mirror_demo")
    assert os.path.exists(exe), "host_mirror_demo not built (run __graft_entry__.build())"
    d = 32
    rng = np.random.default_rng(31)
    v1 = H.sift_like(3000, d, n_clusters=20, seed=41)
    v2 = H.sift_like(1500, d, n_clusters=12, seed=42)
    f1, _, _ = H.build_spann_files(oracle, v1, list(range(3000)), 30, max_neighbors=8, max_layers=3, ef_construction=50)
    f2, _, _ = H.build_spann_files(oracle, v2, list(range(100_000, 101_500)), 15, max_neighbors=8, max_layers=3, ef_construction=50)
    cats = [F.concat_multi_spann({7: f1}), F.concat_multi_spann({7: f2, 8: f2})]
    q = (v1[rng.integers(0, 3000, 6)] + rng.normal(0, 3, (6, d))).astype(np.float32)
    segs = []
    for s, cat in enumerate(cats):
        sd = tmp_path / ("seg%d" % s)
        sd.mkdir()
        for name in ("user_table", "hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors"):
            (sd / name).write_bytes(bytes(cat[name]))
        segs.append(MultiSpannIndex(ctx, cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"]))
    p = SearchParams(5, 50).with_num_explored_centroids(6).with_centroid_distance_ratio(0.3)
    first = PendingSegment(segs[:1]).search_with_id(7, q[0], p)
    dead = {7: [first[0][0], first[2][0]]}
    (tmp_path / "dead.txt").write_text("".join("7 %d\n" % doc for doc in dead[7]))
    (tmp_path / "users.txt").write_text("7\n8\n999\n")
    (tmp_path / "queries.f32").write_bytes(q.tobytes())
    # the finalized segment had deletes: its tombstone log (3 files of 64 bytes; a user it does not hold; a pair twice) is
    # replayed by InvalidatedIdsStorage::read + MultiSpannIndex::open_invalidated_ids of the C++ mirror and by the Python one
    top8 = segs[1].search_for_user([8], q[1:2], p).doc_ids(0)
    log = F.InvalidatedIdsStorage(str(tmp_path / "seg1" / "invalidated_ids_storage"), 64)
    os.makedirs(log.base_directory)
    log.invalidate_batch([(8, top8[0]), (999, 5), (8, top8[1]), (8, top8[0]), (7, 100_000)])
    assert segs[1].replay_invalidations(F.InvalidatedIdsStorage.read(log.base_directory).record_bytes()) == 3
    out = subprocess.run([exe, "segments", str(tmp_path), str(d), "5", "50", "6", "0.3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "replayed 1 3" in out.stdout.splitlines()
    got = {}
    for line in out.stdout.splitlines():
        t = line.split()
        if t[0] == "replayed":
            continue
        got[(t[0], int(t[1]))] = None if t[2] == "none" else [(int(x.split(":")[0]), int(x.split(":")[1], 16)) for x in t[3:]]

    def bits(rows):
        return [(int(i), struct.unpack("<I", struct.pack("<f", float(s)))[0]) for i, s in rows]
    pend = PendingSegment(segs[:1], dead)
    snap = Snapshot([pend, segs[1]])
    for i in range(len(q)):
        assert got[("pending", i)] == bits(pend.search_with_id(7, q[i], p)), i
        assert got[("snap_user", i)] == bits(snap.search_for_user(7, q[i], p)), i
        assert got[("snap_users", i)] == bits(snap.search_for_users([7, 8, 999], q[i], p)), i
    assert all(doc not in [r[0] for r in got[("pending", 0)]] for doc in dead[7]) and len(got[("pending", 0)]) == 5


def test_snapshot_planner_is_per_segment_and_user(ctx, oracle, tmp_path):
    """the reference builds ONE Planner per (segment, user) (collection/snapshot.rs:82-95) — its bitmap
    indexes THAT user's point ids in THAT segment.  Two finalized segments of different sizes, two users, four different
    bitmaps: Snapshot::search_for_user / search_for_users (Python mirror and include/muopdb_host.hpp through the demo) must
    equal the per-segment oracle searches under their own filters, merged by IdWithScore; a bare bitmap for several finalized
    segments (or several users) is refused instead of silently filtering the wrong points."""
    import os
    import struct
    import subprocess
    from muopdb_amd.index import MultiSpannIndex, SearchParams, Snapshot, allow_bitmap
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "muopdb_amd", "host_mirror_demo")
    assert os.path.exists(exe), "host_mirror_demo not built (run __graft_entry__.build())"
    d = 32
    rng = np.random.default_rng(53)
    sizes = {0: {7: 3000, 8: 900}, 1: {7: 1500, 8: 2100}}
    cats, plans = [], {}
    for si, users in sizes.items():
        files = {}
        for u, n in users.items():
            v = H.sift_like(n, d, n_clusters=12, seed=100 + 10 * si + u)
            base = 1_000_000 * si + 100_000 * u
            files[u], _, _ = H.build_spann_files(oracle, v, list(range(base, base + n)), max(8, n // 100), max_neighbors=8, max_layers=3,
                                                 ef_construction=50)
            plans[(si, u)] = allow_bitmap(rng.choice(n, size=n // 3, replace=False), n)
        cats.append(F.concat_multi_spann(files))
    q = H.sift_like(6, d, n_clusters=12, seed=107).astype(np.float32)
    segs, osegs = [], []
    for s, cat in enumerate(cats):
        sd = tmp_path / ("seg%d" % s)
        sd.mkdir()
        for name in ("user_table", "hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors"):
            (sd / name).write_bytes(bytes(cat[name]))
        a = (cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
        segs.append(MultiSpannIndex(ctx, *a))
        osegs.append(oracle.MultiSpannIndex(*a))
    p = SearchParams(5, 50).with_num_explored_centroids(6).with_centroid_distance_ratio(0.3)
    op = oracle.SearchParams(5, 50, num_explored_centroids=6, centroid_distance_ratio=0.3)
    snap = Snapshot(segs)
    planner = lambda si, u: plans.get((si, u))       # user 999 / unknown pairs: no planner

    def want_user(u, qi):
        rows = []
        for si in range(2):
            bm = plans.get((si, u))
            if bm is None:
                res = osegs[si].search_for_user([u], q[qi:qi + 1], op)
            else:
                with oracle.planner_filter(bm):
                    res = osegs[si].search_for_user([u], q[qi:qi + 1], op)
            if res.found[0]:
                rows += res.id_with_scores(0)
        return rows
    key = lambda r: (r[1], r[0])
    for qi in range(len(q)):
        w7 = sorted(want_user(7, qi), key=key)[:5]
        assert snap.search_for_user(7, q[qi], p, planner=planner) == w7, qi
        many = sorted(w7 + sorted(want_user(8, qi), key=key)[:5], key=key)[:5]
        assert snap.search_for_users([7, 8, 999], q[qi], p, planner=planner) == many, qi
        assert len(w7) == 5
    # the filters bite and differ per segment: the unfiltered snapshot returns other rows somewhere
    assert any(snap.search_for_user(7, q[qi], p) != snap.search_for_user(7, q[qi], p, planner=planner) for qi in range(len(q)))
    with pytest.raises(ValueError):
        snap.search_for_user(7, q[0], p, planner=plans[(0, 7)])              # one bitmap, two finalized segments
    with pytest.raises(ValueError):
        Snapshot(segs[:1]).search_for_users([7, 8], q[0], p, planner=plans[(0, 7)])   # one bitmap, two users
    assert Snapshot(segs[:1]).search_for_user(7, q[0], p, planner=plans[(0, 7)]) == \
        Snapshot(segs[:1]).search_for_user(7, q[0], p, planner=lambda si, u: plans[(0, 7)])
    # the compiled mirror
    (tmp_path / "dead.txt").write_text("")
    (tmp_path / "users.txt").write_text("7\n8\n999\n")
    (tmp_path / "queries.f32").write_bytes(q.tobytes())
    (tmp_path / "planners.txt").write_text("".join("%d %d %s\n" % (si, u, " ".join(str(int(w)) for w in bm)) for (si, u), bm in plans.items()))
    out = subprocess.run([exe, "segments", str(tmp_path), str(d), "5", "50", "6", "0.3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = {}
    for line in out.stdout.splitlines():
        t = line.split()
        got[(t[0], int(t[1]))] = None if t[2] == "none" else [(int(x.split(":")[0]), int(x.split(":")[1], 16)) for x in t[3:]]

    def bits(rows):
        return [(int(i), struct.unpack("<I", struct.pack("<f", float(s)))[0]) for i, s in rows]
    for qi in range(len(q)):
        assert got[("plan_user", qi)] == bits(snap.search_for_user(7, q[qi], p, planner=planner)), qi
        assert got[("plan_users", qi)] == bits(snap.search_for_users([7, 8, 999], q[qi], p, planner=planner)), qi


def test_environment_option_words_and_typos():
    """MDB_* environment values are integers or on/off/true/false/yes/no; anything else keeps the option's DEFAULT and
    is reported through mdb_last_error — MDB_FLAT_ROWS=off used to ENABLE the row copy, MDB_FLAT_BLOCKS= meant one block."""
    import os
    import subprocess
    import sys
    code = ("from muopdb_amd import lib as L\n"
            "c = L.Context(0)\n"
            "print([c.get_option(n) for n in ('MDB_FLAT_ROWS', 'MDB_BF_X1', 'MDB_PQ_SDC_MAX_MB', 'MDB_FLAT_BLOCKS', 'MDB_HNSW_NO_DENSE', 'MDB_MF_SAMPLE_DIV')])\n"
            "print((c.lib.mdb_last_error(c.h) or b'').decode())\n")
    env = dict(os.environ, MDB_FLAT_ROWS="off", MDB_BF_X1="False", MDB_PQ_SDC_MAX_MB="none", MDB_FLAT_BLOCKS="", MDB_HNSW_NO_DENSE="YES",
               MDB_MF_SAMPLE_DIV=" 16 ")
    base = subprocess.run([sys.executable, "-c", code.replace("print((c.lib", "#")], capture_output=True, text=True, timeout=300,
                          env={k_: v for k_, v in os.environ.items() if not k_.startswith("MDB_")}, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and base.returncode == 0, out.stderr + base.stderr
    dflt = eval(base.stdout.splitlines()[0])
    got = eval(out.stdout.splitlines()[0])
    assert got == [0, 0, dflt[2], dflt[3], 1, 16], (got, dflt)
    msg = out.stdout.splitlines()[1]
    assert "MDB_PQ_SDC_MAX_MB" in msg and "MDB_FLAT_BLOCKS" in msg and "MDB_FLAT_ROWS" not in msg


def test_resident_bytes_of_a_flat_index_and_a_large_coarse_quantizer(ctx):
    """what an index keeps resident in HBM is visible (mdb_device_mem_info around the load) and bounded: a flat
    L2 index = tiles 1 x + bf16 hi fragments 0.5 x + the row-major copy 1 x + sample (no lo fragments: the default filter never reads
    them) <= 2.7 x its f32 rows (2.53 x at 1 M rows); without the copy (stores above MDB_FLAT_ROWS_MAX_MB) <= 1.65 x; the lo halves come back (+0.5 x) only
    for a store loaded under MDB_BF_X1=0.  Results do not depend on any of it."""
    from muopdb_amd.index import FlatIndex
    n, d = 300_000, 128
    rows = n * d * 4
    rng = np.random.default_rng(8)
    base = H.sift_like(n, d, n_clusters=40, seed=12)
    q = (base[rng.integers(0, n, 40)] + rng.normal(0, 10, (40, d))).astype(np.float32)

    def load(**opts):
        import contextlib
        with contextlib.ExitStack() as st:
            for k_, v in opts.items():
                st.enter_context(ctx.option(k_, v))
            f0 = ctx.mem_info()[0]
            idx = FlatIndex(ctx, base, 0)
            used = f0 - ctx.mem_info()[0]
        return idx, used / rows
    a, ra = load()
    want = a.search(q, 10)
    b_, rb = load(MDB_FLAT_ROWS_MAX_MB=1)
    c_, rc = load(MDB_BF_X1=0)
    for other in (b_, c_):
        got = other.search(q, 10)
        assert all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y) for x, y in zip(got, want))
    # (300 k rows: the sample's floor of 256 tiles is 5 % here, 1/32 at 1 M rows)
    assert 2.3 < ra <= 2.7 and 1.4 < rb <= 1.65 and ra + 0.4 < rc <= 3.2, (ra, rb, rc)
