#!/usr/bin/env python
"""Device-resident multi-user SPANN calls (the path bench.py and a serving host use: the ratio filter in the closure kernel's tail, the
merge launch that also remaps / re-ranks / passes the found flags on and saves the counters) against the SAME handle's host-buffer
calls (separate filter, merge and remap launches) on random collections, batches with unknown users, ragged batches and every
num_explored_centroids / ratio setting; counters must agree too.

    python scripts/stress_spann_device.py --seconds 120 [--seed 0]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from muopdb_amd import formats as F
from muopdb_amd import lib as L
from muopdb_amd.index import MultiSpannIndex, SearchParams
from tests import helpers as H
import oracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t0, it, calls = time.time(), 0, 0
    while time.time() - t0 < args.seconds:
        rng = np.random.default_rng(args.seed * 7907 + it)
        d = int(rng.choice([16, 32, 48, 128]))
        U = int(rng.integers(1, 7))
        users = {}
        for u in range(U):
            n = int(rng.integers(80, 700))
            v = (rng.standard_normal((n, d)) * float(rng.choice([1, 10])) + u).astype(np.float32)
            if rng.integers(0, 3) == 0:
                v = np.round(v)   # ties
            files, _, _ = H.build_spann_files(oracle, v, [1000 * u + i for i in range(n)], int(rng.integers(3, 40)), seed=u,
                                              max_neighbors=int(rng.choice([4, 8, 16])), max_layers=int(rng.integers(1, 4)), ef_construction=30)
            users[(u << 70) | (u + 1)] = (files, v)
        cat = F.concat_multi_spann({uid: f for uid, (f, _) in users.items()})
        ms = MultiSpannIndex(ctx, cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
        uids = list(users)
        for _ in range(6):
            b = int(rng.choice([1, 3, 4, 8, 17, 64, 130, 300]))
            k = int(rng.choice([1, 5, 10, 32]))
            p = SearchParams(k, int(rng.choice([50, 200]))).with_num_explored_centroids(int(rng.choice([1, 4, 16, 64]))) \
                .with_centroid_distance_ratio(float(rng.choice([0.0, 0.1, 0.5, 3.0])))
            pick = [uids[int(rng.integers(0, U))] if rng.integers(0, 8) else 12345 for _ in range(b)]
            q = np.stack([(users[u][1][int(rng.integers(0, len(users[u][1])))] if u in users else np.zeros(d, np.float32))
                          + rng.normal(0, 0.5, d) for u in pick]).astype(np.float32)
            want = ms.search_for_user(pick, q, p)
            st_h = ctx.stats()
            qd = torch.from_numpy(q).cuda()
            ids = torch.zeros((b, k, 2), dtype=torch.int64, device="cuda")
            sc = torch.zeros((b, k), dtype=torch.float32, device="cuda")
            cn = torch.zeros(b, dtype=torch.int32, device="cuda")
            fo = torch.zeros(b, dtype=torch.uint8, device="cuda")
            pc = p.to_c()
            for rep in range(2):   # twice: the second call starts on counters the first one's last launch cleared
                ctx.check(ctx.lib.mdb_multi_spann_search(ms.h, L.u128_array(pick), C.c_void_p(qd.data_ptr()), C.c_size_t(b), C.byref(pc),
                                                         C.c_int(L.MEM_DEVICE), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                         C.c_void_p(cn.data_ptr()), C.c_void_p(fo.data_ptr())))
                ctx.sync()
                st_d = ctx.stats()
                idn, scn, cnn, fon = ids.cpu().numpy().view(np.uint64), sc.cpu().numpy(), cn.cpu().numpy(), fo.cpu().numpy()
                for i in range(b):
                    ok = int(fon[i]) == int(want.found[i]) and (not want.found[i] or int(cnn[i]) == int(want.counts[i]))
                    if ok and want.found[i]:
                        c = int(cnn[i])
                        got = [(int(idn[i, j, 0]) | (int(idn[i, j, 1]) << 64), scn[i, j].tobytes()) for j in range(c)]
                        ok = got == [(int(doc), np.float32(s).tobytes()) for doc, s in want.id_with_scores(i)]
                    if not ok:
                        print("MISMATCH it=%d seed=%d b=%d k=%d query %d rep %d" % (it, args.seed, b, k, i, rep), flush=True)
                        sys.exit(1)
                if (st_d["distance_evals"], st_d["expanded_nodes"], st_d["scored_vectors"]) != (st_h["distance_evals"], st_h["expanded_nodes"], st_h["scored_vectors"]):
                    print("COUNTER MISMATCH it=%d seed=%d rep %d: device %s host %s" % (it, args.seed, rep, st_d, st_h), flush=True)
                    sys.exit(1)
                calls += 1
        ms.close()
        it += 1
    print("device-path SPANN stress OK: %d collections, %d device calls in %.0f s" % (it, calls, time.time() - t0))


if __name__ == "__main__":
    main()
