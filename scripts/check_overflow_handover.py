"""Run under `rocprofv3 --kernel-trace --stats`: a coarse search whose candidate list overflows must hand its overflow count to the
host from the refine kernel itself (last block, one 64-bit atomic per block), so that the NEXT calls are served by the exact kernels
(cooldown): expect flat_refine_group_kernel x1 and flat_scan_kernel x2 in the stats, and identical probes from all three calls."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from muopdb_amd import lib as L, formats as F
from muopdb_amd.index import BlockBasedIvf
ctx = L.Context(0)
rng = np.random.default_rng(1)
n, d = 65600, 24
cent = (rng.standard_normal((n, d)) * 30).astype(np.float32)
rows = rng.choice(n, 9000, replace=False)
cent[rows] = cent[rows[0]]
g = BlockBasedIvf(ctx, F.write_ivf_index(cent, list(range(1, n + 1)), [np.array([i], dtype=np.uint64) for i in range(n)]), F.write_vector_file(cent))
q = (cent[rng.integers(0, n, 40)]).astype(np.float32)
q[3] = cent[rows[0]]
ctx.set_option("MDB_REFINE_WAVE_MIN_B", 8); ctx.set_option("MDB_REFINE_GROUP_MIN_B", 8)
a = g.find_nearest_centroids(q, 24)   # query 3 overflows its list: the word must reach the host ...
b = g.find_nearest_centroids(q, 24)   # ... so that this call is served by the exact kernels (cooldown)
c = g.find_nearest_centroids(q, 24)
assert np.array_equal(a, b) and np.array_equal(a, c)
print("ok")
