#!/usr/bin/env python
"""Stress of the batched flat path (sample bound -> MFMA filter -> exact refine) against the exact kernels on
the same index: any missed neighbour (a hole in the filter's error bound) shows up as an id / score mismatch.
Data kinds include large common offsets, tiny spreads, huge magnitudes, duplicates and near-ties.

    python scripts/stress_mfma.py --seconds 300 [--seed 0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from muopdb_amd import lib as L
from muopdb_amd.index import FlatIndex


def make(rng, n, d, kind):
    if kind == 0:
        x = rng.standard_normal((n, d))
    elif kind == 1:   # clusters far from the origin
        c = rng.uniform(-1, 1, (int(rng.integers(2, 200)), d)) * float(rng.choice([1, 100, 1e4]))
        x = c[rng.integers(0, len(c), n)] + rng.standard_normal((n, d)) * float(rng.choice([0.01, 1, 10]))
    elif kind == 2:   # common offset >> spread
        x = rng.standard_normal((n, d)) * float(rng.choice([1e-3, 1, 30])) + float(rng.choice([10, 1e3, 1e5]))
    elif kind == 3:   # integer grid with many ties
        x = rng.integers(0, int(rng.choice([2, 4, 256])), (n, d)).astype(np.float64)
    elif kind == 4:   # wide dynamic range per dimension
        x = rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 6, d))[None, :]
    else:             # unit vectors
        x = rng.standard_normal((n, d))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    ctx = L.Context(0)
    t0, it, used = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        rng = np.random.default_rng(args.seed * 7919 + it)
        n = int(rng.choice([66000, 100000, 200000, 300000]))
        d = int(rng.choice([4, 16, 30, 64, 128, 200, 768])) if n <= 100000 else int(rng.choice([4, 16, 30, 64, 128]))
        b = int(rng.choice([8, 17, 32, 33, 64, 100, 130]))
        k = int(rng.choice([1, 5, 10, 32, 100]))
        metric = int(rng.integers(0, 2))
        kind = int(rng.integers(0, 6))
        base = make(rng, n, d, kind)
        if rng.integers(0, 2):
            q = (base[rng.integers(0, n, b)] + rng.standard_normal((b, d)).astype(np.float32) * np.float32(rng.choice([0, 1e-3, 1]))).astype(np.float32)
        else:
            q = make(rng, b, d, kind)
        idx = FlatIndex(ctx, base, metric)
        for rep in range(2):   # second call: the cooldown state after an overflow must not change results either
            ids, dist, cnt = idx.search(q, k)
            with ctx.option("MDB_FLAT_NO_MFMA", 1):
                eids, edist, ecnt = idx.search(q, k)
            if not (np.array_equal(ids, eids) and np.array_equal(dist.view(np.uint32), edist.view(np.uint32)) and np.array_equal(cnt, ecnt)):
                bad = np.nonzero((ids != eids).any(1))[0]
                print("MISMATCH it=%d seed=%d cfg=%s rows=%s" % (it, args.seed, dict(n=n, d=d, b=b, k=k, metric=metric, kind=kind), bad[:5]), flush=True)
                print(ids[bad[0]] if len(bad) else None, eids[bad[0]] if len(bad) else None)
                sys.exit(1)
        idx.close()
        it += 1
    print("mfma stress OK: %d index/query sets in %.0f s" % (it, time.time() - t0))


if __name__ == "__main__":
    main()
