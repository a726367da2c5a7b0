#!/usr/bin/env python
"""Writes tests/golden/c1_flat.npz — BASELINE config C1 (flat 10k x 128, batch 1, top-10).

The base is the array py/create_test_hdf5.py of the reference generates (np.random.seed(42), 10 clusters x
1000 points, centre i*100, N(0, 5^2), shuffled, cast f32; tests/helpers.py::test_hdf5_like restates it without
h5py).  Only the first 256 rows are stored (to check that a regenerated base is the same array); the queries
(same generator, seed 43) and the expected top-10 (ids, f32 distances) over the FULL base come from the CPU
oracle, cross-checked here against a float64 brute force.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402

base = H.test_hdf5_like()
queries = H.test_hdf5_like(n_per=10, seed=43)[:16]
ids, dist = oracle.flat_topk(0, base, queries, 10)
d64 = np.sqrt(((queries[:, None, :].astype(np.float64) - base[None]) ** 2).sum(-1))
assert np.array_equal(np.sort(ids, 1), np.sort(np.argsort(d64, 1)[:, :10], 1)), "oracle disagrees with the f64 brute force"
out = os.path.join(ROOT, "tests", "golden", "c1_flat.npz")
np.savez_compressed(out, base_head=base[:256], queries=queries, ids=ids.astype(np.uint32), dist=dist.astype(np.float32))
print("wrote", out, os.path.getsize(out), "bytes")
