#!/usr/bin/env python
"""Writes tests/golden/{hnsw_small,ivfpq_small,mspann_small}.npz: small index FILES in the reference's on-disk formats
(built with the oracle's HnswBuilder restatement / muopdb_amd.formats), seeded queries and the CPU oracle's answers —
doc ids (u128 as lo/hi words), f32 score bits, and for HNSW the traversal counters.  tests/test_oracle_kat.py checks that the
oracle still reproduces them (a regression pin of the restatement); tests/test_gpu_parity.py feeds the same files through
the HIP path.  The fixtures are data (inputs + expected outputs), a few tens of KB each."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from muopdb_amd import formats as F  # noqa: E402
from tests import helpers as H  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rows(res, b, k):
    lo = np.full((b, k), np.iinfo(np.uint64).max, np.uint64)
    hi = np.full((b, k), np.iinfo(np.uint64).max, np.uint64)
    sc = np.zeros((b, k), np.float32)
    cnt = np.zeros(b, np.uint32)
    for i in range(b):
        ids = res.doc_ids(i)
        cnt[i] = len(ids)
        for j, d in enumerate(ids):
            lo[i, j], hi[i, j] = d & 0xFFFFFFFFFFFFFFFF, d >> 64
        sc[i, :len(ids)] = res.scores[i, :len(ids)]
    return dict(lo=lo, hi=hi, score_bits=sc.view(np.uint32), counts=cnt)


def u8(b):
    return np.frombuffer(bytes(b), np.uint8)


def save(name, **kw):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes")


rng = np.random.default_rng(2026)

# ---- HNSW: 500 x 12, M=8, 4 layers, doc ids with high words, k=6, ef in {40 (beam kernel), 600 (closure: n <= ef)}
v = H.sift_like(500, 12, n_clusters=9, seed=11)
doc = [7 * i + 3 + ((i % 3) << 70) for i in range(500)]
hidx, hvec = H.build_hnsw_files(oracle, v, doc, max_neighbors=8, max_layers=4, ef_construction=40, seed=5)
q = (v[rng.integers(0, 500, 10)] + rng.normal(0, 3, (10, 12))).astype(np.float32)
o = oracle.BlockBasedHnsw(hidx, hvec, 12)
out = dict(index=u8(hidx), vectors=u8(hvec), queries=q, dimension=np.int32(12), k=np.int32(6))
for ef in (40, 600):
    o.stats()
    r = o.ann_search(q, 6, ef)
    ev, ex = o.stats()
    out.update({"ef%d_%s" % (ef, kk): vv for kk, vv in rows(r, 10, 6).items()})
    out["ef%d_counters" % ef] = np.array([ev, ex], np.uint64)
save("hnsw_small.npz", **out)

# ---- IVF-PQ: 1500 x 32, 12 lists, PQ subdim 8 x 5 bits, k=7, 4 probes, then two tombstones
v = H.sift_like(1500, 32, n_clusters=10, seed=12)
doc = [100 + 3 * i + ((i % 7) << 65) for i in range(1500)]
cent = H.kmeans(v, 12, iters=4, seed=3)
cb = H.train_pq_codebook(v[:1000], 8, 5, iters=3)
opq = oracle.ProductQuantizer(32, 8, 5, cb)
index, vec, _ = H.build_ivf_files(v, doc, cent, quantize=opq.quantize)
q = (v[rng.integers(0, 1500, 12)] + rng.normal(0, 2, (12, 32))).astype(np.float32)
o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 5, cb))
out = dict(index=u8(index), vectors=u8(vec), codebook=np.asarray(cb, np.float32), queries=q, k=np.int32(7), nprobe=np.int32(4),
           probes=o.find_nearest_centroids(q, 4).astype(np.uint32))
r = o.search(q, 7, num_probes=4)
out.update({"a_" + kk: vv for kk, vv in rows(r, 12, 7).items()})
dead = [r.doc_ids(0)[0], r.doc_ids(5)[1]]
for d in dead:
    assert o.invalidate(d)
out["dead_lo"] = np.array([d & 0xFFFFFFFFFFFFFFFF for d in dead], np.uint64)
out["dead_hi"] = np.array([d >> 64 for d in dead], np.uint64)
out.update({"b_" + kk: vv for kk, vv in rows(o.search(q, 7, num_probes=4), 12, 7).items()})
save("ivfpq_small.npz", **out)

# ---- multi-user SPANN: 3 users (400 / 60 / 900 x 8), queries incl. an unknown user, ratio filter 0.3
users, per = {}, []
for u, n in enumerate((400, 60, 900)):
    x = H.sift_like(n, 8, n_clusters=5, seed=20 + u)
    files, _, _ = H.build_spann_files(oracle, x, [1000 * u + i for i in range(n)], max(2, n // 40), max_neighbors=6, max_layers=3,
                                      ef_construction=30, seed=30 + u)
    users[11 + 5 * u] = files
    per.append(x)
cat = F.concat_multi_spann(users)
uids = [11, 16, 21, 999, 21, 11]
q = np.stack([per[min((u - 11) // 5, 2) if u != 999 else 0][rng.integers(0, 60)] + rng.normal(0, 1, 8) for u in uids]).astype(np.float32)
o = oracle.MultiSpannIndex(cat["user_table"], 8, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
p = oracle.SearchParams(5, 50, num_explored_centroids=4, centroid_distance_ratio=0.3)
r = o.search_for_user(uids, q, p)
out = {kk: u8(cat[kk]) for kk in ("hnsw_index", "hnsw_vectors", "ivf_index", "ivf_vectors")}
out["user_table"] = np.frombuffer(bytes(cat["user_table"]), np.uint8) if not isinstance(cat["user_table"], np.ndarray) else cat["user_table"]
out.update(queries=q, user_ids=np.array(uids, np.uint64), found=np.array([bool(f) for f in r.found], np.uint8))
out.update(rows(r, len(uids), 5))
save("mspann_small.npz", **out)
