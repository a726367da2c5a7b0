#!/usr/bin/env python
"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random shapes, metrics, quantizers, k / ef /
probe counts, duplicates and tombstones for every index type, for a time budget.  Any mismatch prints the
configuration (reproducible from its seed) and exits non-zero.

    python scripts/stress_parity.py --seconds 300 [--seed 0] [--only flat,ivf,hnsw,spann]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import oracle
from muopdb_amd import formats as F
from muopdb_amd import lib as L
from muopdb_amd.index import (BlockBasedHnsw, BlockBasedIvf, FlatIndex, MultiSpannIndex, NoQuantizer, ProductQuantizer,
                              SearchParams, Spann)
from tests import helpers as H


def rows_equal(g, o, nq):
    for i in range(nq):
        if bool(g.found[i]) != bool(o.found[i]):
            return "found flag of query %d" % i
        if not g.found[i]:
            continue
        if g.doc_ids(i) != o.doc_ids(i):
            return "doc ids of query %d: %s vs %s" % (i, g.doc_ids(i)[:6], o.doc_ids(i)[:6])
        n = int(g.counts[i])
        if not np.array_equal(np.asarray(g.scores[i, :n], np.float32).view(np.uint32), np.asarray(o.scores[i, :n], np.float32).view(np.uint32)):
            return "score bits of query %d" % i
    return None


def data(rng, n, d, kind):
    if kind == 0:
        return H.sift_like(n, d, n_clusters=int(rng.integers(2, 40)), seed=int(rng.integers(1 << 30)))
    if kind == 1:
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == 2:  # heavy duplicates / ties
        base = rng.integers(0, 4, (max(n // 8, 1), d)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    if kind == 3:
        v = rng.standard_normal((n, d)).astype(np.float32)
        return (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-6)).astype(np.float32)
    # a few infinite components (legal: NotNan accepts inf; distances become inf, never NaN for L2 vs finite
    # queries as long as no row mixes +inf and -inf against an infinite query)
    v = rng.standard_normal((n, d)).astype(np.float32) * 10
    rows = rng.integers(0, n, max(1, n // 50))
    v[rows, rng.integers(0, d, len(rows))] = np.inf
    return v


def case_flat(ctx, rng):
    n = int(rng.choice([1, 7, 64, 65, 1000, 5000, 70000, 140000]))
    d = int(rng.choice([1, 3, 4, 16, 17, 30, 100, 128, 200, 768])) if n < 60000 else int(rng.choice([4, 16, 30, 128]))
    b = int(rng.choice([1, 2, 3, 5, 8, 33, 64, 100]))
    k = int(rng.choice([1, 3, 10, 50, 200]))
    metric = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 5))
    if kind == 4:
        metric = 0  # dot with infinite components yields NaN (inf * 0, inf - inf): both sides error out
    cfg = dict(n=n, d=d, b=b, k=k, metric=metric, kind=kind)
    base = data(rng, n, d, kind)
    q = (base[rng.integers(0, n, b)] + rng.normal(0, 1, (b, d))).astype(np.float32)
    q = np.where(np.isfinite(q), q, np.float32(0))  # finite queries: inf - inf would be a NaN distance on both sides
    small = int(rng.choice([0, 0, 1, 2, 4]))   # the small-base flat kernels' forms (bases <= 1024 tiles, batches <= 4): two launches (default) / off / unordered groups / one launch
    cfg.update(small=small)
    with ctx.option("MDB_FLAT_NO_SMALL", small):
        ids, dist, cnt = FlatIndex(ctx, base, metric).search(q, k)
    oids, odist = oracle.flat_topk(metric, base, q, k)
    kk = min(k, n)
    if not np.array_equal(ids[:, :kk], oids[:, :kk]):
        return cfg, "ids"
    if not np.array_equal(dist[:, :kk].view(np.uint32), odist[:, :kk].view(np.uint32)):
        return cfg, "score bits"
    return cfg, None


def pick_pq(rng, d):
    subs = [s for s in (1, 2, 3, 4, 5, 6, 8, 16, 32) if d % s == 0 and d // s <= 64]
    sub = int(rng.choice(subs))
    bits = int(rng.integers(1, 9))
    return sub, bits


def case_ivf(ctx, rng):
    n = int(rng.choice([50, 700, 3000, 9000]))
    d = int(rng.choice([4, 8, 16, 24, 32, 64, 128]))
    nl = int(rng.choice([1, 3, 17, 64, 300]))
    nl = min(nl, n)
    P = int(rng.integers(1, nl + 1))
    k = int(rng.choice([1, 5, 10, 64, 100]))
    cpv = int(rng.choice([1, 1, 2]))
    kind = int(rng.integers(0, 3))
    usepq = bool(rng.integers(0, 2))
    metric = int(rng.integers(0, 2)) if usepq else 0
    cfg = dict(n=n, d=d, nl=nl, P=P, k=k, cpv=cpv, kind=kind, pq=usepq, metric=metric)
    v = data(rng, n, d, kind)
    cent = H.kmeans(v, nl, iters=3, seed=int(rng.integers(1 << 30)))
    doc = [int(x) for x in rng.permutation(n * 3)[:n]]
    if usepq:
        sub, bits = pick_pq(rng, d)
        cfg.update(sub=sub, bits=bits)
        cb = H.train_pq_codebook(v[: min(n, 1500)], sub, bits, iters=2)
        opq = oracle.ProductQuantizer(d, sub, bits, cb, metric)
        index, vec, _ = H.build_ivf_files(v, doc, cent, quantize=opq.quantize, clusters_per_vector=cpv)
        g = BlockBasedIvf(ctx, index, vec, ProductQuantizer(d, sub, bits, cb, metric))
        o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, metric, sub, bits, cb))
    else:
        index, vec, _ = H.build_ivf_files(v, doc, cent, clusters_per_vector=cpv)
        g, o = BlockBasedIvf(ctx, index, vec), oracle.BlockBasedIvf(index, vec)
    b = int(rng.choice([1, 4, 19, 64, 300, 530]))   # 300 / 530: one one-phase block per query / the two-phase PQ scan's range
    if b > 64 and n > 3000:
        b = 64
    q = (v[rng.integers(0, n, b)] + rng.normal(0, 1, (b, d))).astype(np.float32)
    for step in range(2):
        err = rows_equal(g.search(q, k, P), o.search(q, k, num_probes=P), b)
        if err:
            return cfg, "search pass %d: %s" % (step, err)
        for dd in rng.choice(doc, size=min(5, n), replace=False):
            if g.invalidate(int(dd)) != o.invalidate(int(dd)):
                return cfg, "invalidate flag"
    return cfg, None


def case_ivf_fused(ctx, rng):
    """shapes inside the range of the fused small-batch step (ivf_prep_kernel + ivf_pq_fused_kernel: L2, 8-bit codes in whole
    4-byte words, k and probes <= 64, batches < 512), with the library's own coarse search or caller-given probes, tombstones,
    duplicate assignments, low-entropy data (masses of exact ties) and shrunk candidate lists (the in-kernel overflow pass)"""
    n = int(rng.choice([200, 1500, 6000, 20000]))
    sub = int(rng.choice([4, 8, 16, 32]))
    mw = int(rng.choice([1, 2, 4, 8]))
    d = sub * 4 * mw
    if d > 256:
        d, mw = sub * 4, 1
    nl = int(rng.choice([1, 5, 40, 64, 130, 700]))
    nl = min(nl, n)
    P = int(rng.integers(1, min(nl, 64) + 1))
    k = int(rng.choice([1, 3, 10, 33, 64]))
    cpv = int(rng.choice([1, 1, 2]))
    kind = int(rng.integers(0, 3))
    cfg = dict(n=n, d=d, sub=sub, nl=nl, P=P, k=k, cpv=cpv, kind=kind)
    v = data(rng, n, d, kind)
    cent = H.kmeans(v, nl, iters=3, seed=int(rng.integers(1 << 30)))
    doc = [int(x) for x in rng.permutation(n * 3)[:n]]
    cb = H.train_pq_codebook(v[: min(n, 1500)], sub, 8, iters=2)
    opq = oracle.ProductQuantizer(d, sub, 8, cb, 0)
    index, vec, _ = H.build_ivf_files(v, doc, cent, quantize=opq.quantize, clusters_per_vector=cpv)
    g = BlockBasedIvf(ctx, index, vec, ProductQuantizer(d, sub, 8, cb, 0))
    o = oracle.BlockBasedIvf(index, vec, oracle.Quant(oracle.QUANT_PQ, 0, sub, 8, cb))
    b = int(rng.choice([1, 7, 64, 256, 300]))
    if n > 6000:
        b = min(b, 64)
    q = (v[rng.integers(0, n, b)] + rng.normal(0, 1, (b, d))).astype(np.float32)
    cap = int(rng.choice([2048, 2048, 16]))
    cfg.update(b=b, cap=cap)
    with ctx.option("MDB_PQF_CAP", cap):
        for step in range(2):
            want = o.search(q, k, num_probes=P)
            err = rows_equal(g.search(q, k, P), want, b)
            if err:
                return cfg, "search pass %d: %s" % (step, err)
            probes = o.find_nearest_centroids(q, P)
            err = rows_equal(g.search_with_centroids_and_remap(q, probes, k), want, b)
            if err:
                return cfg, "given probes, pass %d: %s" % (step, err)
            for dd in rng.choice(doc, size=min(5, n), replace=False):
                if g.invalidate(int(dd)) != o.invalidate(int(dd)):
                    return cfg, "invalidate flag"
    return cfg, None


def case_hnsw(ctx, rng):
    n = int(rng.choice([1, 2, 40, 600, 2500]))
    d = int(rng.choice([3, 4, 16, 30, 48, 128]))
    M = int(rng.choice([4, 8, 16, 32, 40]))   # 40: rows longer than 64 edges (the multi-chunk beam kernel)
    layers = int(rng.integers(1, 6))
    metric = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    usepq = bool(rng.integers(0, 3) == 0) and n >= 40
    cfg = dict(n=n, d=d, M=M, layers=layers, metric=metric, kind=kind, pq=usepq)
    v = data(rng, n, d, kind)
    doc = [int(x) for x in rng.permutation(n * 2)[:n]]
    b = oracle.HnswBuilder(d, M, layers, int(rng.choice([10, 40, 100])), metric, int(rng.integers(1 << 20)))
    b.insert(v)
    lay, eps = b.layers(), b.entry_points()
    if len(lay) > 1:
        top = lay[-1]
        lay[-1] = {eps[0]: top[eps[0]], **{p: e for p, e in top.items() if p != eps[0]}}
    if usepq:
        sub, bits = pick_pq(rng, d)
        cfg.update(sub=sub, bits=bits)
        cb = H.train_pq_codebook(v[: min(n, 1000)], sub, bits, iters=2)
        codes = oracle.ProductQuantizer(d, sub, bits, cb, metric).quantize(v)
        hidx, hvec = F.write_hnsw_index(lay, doc, d // sub), F.write_vector_file(codes)
        g = BlockBasedHnsw(ctx, hidx, hvec, d, ProductQuantizer(d, sub, bits, cb, metric))
        o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_PQ, metric, sub, bits, cb))
    else:
        hidx, hvec = F.write_hnsw_index(lay, doc, d), F.write_vector_file(v)
        g = BlockBasedHnsw(ctx, hidx, hvec, d, NoQuantizer(d, metric))
        o = oracle.BlockBasedHnsw(hidx, hvec, d, oracle.Quant(oracle.QUANT_NONE, metric))
    nq = int(rng.choice([1, 9, 40]))
    q = (v[rng.integers(0, n, nq)] + rng.normal(0, 1, (nq, d))).astype(np.float32)
    rank = int(rng.integers(0, 4))   # upper layers on sorted positions: neither launch / the single or layer-1 launch / the top launch / both
    cfg.update(rank=rank)
    for k, ef in [(int(rng.choice([1, 5, 10, 80])), int(rng.choice([0, 1, 7, 64, 200, 256, 257, 400, 700])))]:
        cfg.update(k=k, ef=ef)
        o.stats()
        ores = o.ann_search(q, k, ef)
        evals, expanded = o.stats()
        with ctx.option("MDB_HNSW_RANK", rank):
            gres = g.ann_search(q, k, ef)
        err = rows_equal(gres, ores, nq)
        if err:
            return cfg, err
        st = ctx.stats()
        if (st["distance_evals"], st["expanded_nodes"]) != (evals, expanded):
            return cfg, "traversal counters %s vs %s" % ((st["distance_evals"], st["expanded_nodes"]), (evals, expanded))
    return cfg, None


def case_spann(ctx, rng):
    users = {}
    d = int(rng.choice([4, 16, 32]))
    usepq = bool(rng.integers(0, 3) == 0)
    nu = int(rng.choice([1, 2, 5]))
    cfg = dict(d=d, pq=usepq, users=nu)
    quant = oquant = None
    cb = None
    if usepq:
        sub, bits = pick_pq(rng, d)
        cfg.update(sub=sub, bits=bits)
    per_user = []
    for u in range(nu):
        n = int(rng.choice([30, 400, 1500]))
        v = data(rng, n, d, int(rng.integers(0, 3)))
        if usepq and cb is None:
            cb = H.train_pq_codebook(v[: min(n, 1000)], sub, bits, iters=2)
            opq = oracle.ProductQuantizer(d, sub, bits, cb)
            quant, oquant = ProductQuantizer(d, sub, bits, cb), oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, sub, bits, cb)
        nl = int(min(n, rng.choice([1, 4, 20])))
        doc = [int(x) + 1000 * u for x in range(n)]
        files, _, _ = H.build_spann_files(oracle, v, doc, nl, quantize=(opq.quantize if usepq else None), max_neighbors=6,
                                          max_layers=3, ef_construction=30, seed=int(rng.integers(1 << 20)))
        users[10 + 7 * u] = files
        per_user.append(v)
    cat = F.concat_multi_spann(users)
    g = MultiSpannIndex(ctx, cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], quant)
    o = oracle.MultiSpannIndex(cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"], oquant)
    b = int(rng.choice([1, 6, 30]))
    uids, qs = [], []
    for _ in range(b):
        u = int(rng.integers(0, nu + 1))  # nu = unknown user
        uids.append(10 + 7 * u if u < nu else 999)
        vv = per_user[min(u, nu - 1)]
        qs.append(vv[rng.integers(0, vv.shape[0])] + rng.normal(0, 1, d))
    q = np.asarray(qs, np.float32)
    k, ef = int(rng.choice([1, 5, 20])), int(rng.choice([1, 10, 100]))
    nexp = int(rng.choice([1, 3, 8]))
    ratio = float(rng.choice([0.0, 0.1, 0.5, 10.0]))
    cfg.update(k=k, ef=ef, nexp=nexp, ratio=ratio, b=b)
    p = SearchParams(k, ef).with_num_explored_centroids(nexp).with_centroid_distance_ratio(ratio)
    op = oracle.SearchParams(k, ef, num_explored_centroids=nexp, centroid_distance_ratio=ratio)
    err = rows_equal(g.search_for_user(uids, q, p), o.search_for_user(uids, q, op), b)
    return cfg, err


CASES = {"flat": case_flat, "ivf": case_ivf, "ivf_fused": case_ivf_fused, "hnsw": case_hnsw, "spann": case_spann}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="flat,ivf,hnsw,spann")
    ap.add_argument("--start", type=int, default=0, help="first iteration (a case's generator is seeded by its iteration number)")
    args = ap.parse_args()
    ctx = L.Context(0)
    names = args.only.split(",")
    t0 = time.time()
    counts = {n: 0 for n in names}
    it = args.start
    while time.time() - t0 < args.seconds:
        name = names[it % len(names)]
        seed = args.seed * 1_000_003 + it
        rng = np.random.default_rng(seed)
        try:
            cfg, err = CASES[name](ctx, rng)
        except L.MuopdbError as e:
            cfg, err = {"exception": str(e)}, "library error"
        if err:
            print("MISMATCH %s seed=%d it=%d: %s\n  config: %s" % (name, args.seed, it, err, cfg), flush=True)
            sys.exit(1)
        counts[name] += 1
        it += 1
    print("stress parity OK:", counts, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
