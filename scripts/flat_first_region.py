"""Where does the FIRST timed region of the flat batch-1 workloads lose its time (bench r05a: 0.80 ms/step first, 0.10 median)?
Per-step host + device times of the first steps after the warm-up, with the profiling brackets of Env.timed."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from muopdb_amd import lib as L
from muopdb_amd.index import FlatIndex

n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128, 10
ctx = L.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cpu"); g.manual_seed(1)
x = (torch.rand((n, d), generator=g) * 200).round().cuda()
q = (torch.rand((400, d), generator=g) * 200).round().cuda()
idx = FlatIndex(ctx, x.cpu().numpy(), 0)
ids = torch.zeros((1, k), dtype=torch.int64, device="cuda"); sc = torch.zeros((1, k), dtype=torch.float32, device="cuda"); cn = torch.zeros(1, dtype=torch.int32, device="cuda")


def step(i):
    idx.search_device(q[i:i + 1].data_ptr(), 1, k, ids.data_ptr(), sc.data_ptr(), cn.data_ptr())


for i in range(5):
    step(i)
ctx.sync(); ctx.set_profiling(False); ctx.get_profile(); torch.cuda.synchronize()
for rep in range(3):
    ts = []
    t00 = time.perf_counter()
    for j in range(50):
        if j % 4 == 0:
            ctx.set_profiling(True)
        elif j % 4 == 1:
            ctx.set_profiling(False)
        t0 = time.perf_counter()
        step(5 + j)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t00
    print("region %d: %.3f ms/step; host per call us: max %.0f at step %d, median %.0f, first five %s" % (
        rep, 1000 * tot / 50, 1e6 * max(ts), int(np.argmax(ts)), 1e6 * float(np.median(ts)), [int(1e6 * t) for t in ts[:5]]))
    print("   profile:", ctx.get_profile())
