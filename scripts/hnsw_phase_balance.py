#!/usr/bin/env python3
"""What a batch of 64 HNSW queries loses to its slowest query PER PHASE: the three launches of the table path (top layers + table |
layer 1 | layer 0) each end on their slowest query, so a batch costs the sum of three maxima.  A batch of 64 copies of ONE query
costs that query's own chain (every phase's maximum is that query) — the maximum of these times over the batch's 64 queries is
what a per-query pipeline across the launches would cost the mixed batch, their mean what perfectly even queries would.
usage (GPU box): python scripts/hnsw_phase_balance.py [n] — prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from muopdb_amd import lib as L, synth as S  # noqa: E402
from muopdb_amd.index import BlockBasedHnsw  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
torch.cuda.set_device(0)
ctx = L.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
env = bench.Env(args, ctx, 0, 1)
B, k, ef, d = 64, 10, 200, 128
x, queries, _ = env.sift(n, d, 4 * B, 1000)
index_bytes, vec_bytes = S.hnsw_files(x, max_neighbors=32, max_layers=8, kcand=64, seed=1)
h = BlockBasedHnsw(ctx, index_bytes, vec_bytes, d)
ids = torch.zeros((B, k, 2), dtype=torch.int64, device="cuda")
sc = torch.zeros((B, k), dtype=torch.float32, device="cuda")
cn = torch.zeros(B, dtype=torch.int32, device="cuda")


def timed(q, reps=20):
    for _ in range(3):
        h.ann_search_device(q.data_ptr(), B, k, ef, ids.data_ptr(), sc.data_ptr(), cn.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        h.ann_search_device(q.data_ptr(), B, k, ef, ids.data_ptr(), sc.data_ptr(), cn.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for bi in range(2):
    qb = queries[bi * B:(bi + 1) * B].contiguous()
    mixed = timed(qb)
    own = np.array([timed(qb[i:i + 1].repeat(B, 1).contiguous(), reps=8) for i in range(B)])
    out.append(dict(batch=bi, mixed_ms=round(mixed, 4), own_max_ms=round(float(own.max()), 4), own_mean_ms=round(float(own.mean()), 4),
                    own_min_ms=round(float(own.min()), 4)))
print(json.dumps(out))
