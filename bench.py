#!/usr/bin/env python
"""bench.py — measures BASELINE.json's metric on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = BASELINE.json configs[1]: SIFT-1M-like 1M x 128 f32 (synthetic, BASELINE.md
C2), HNSW ef=200, top-10, batch=64, through the C ABI (libmuopdb_hip.so) with queries and
outputs resident in HBM.  A "step" is one batch of 64 queries through BlockBasedHnsw::ann_search.
N>1: HNSW does not shard (replicas only) — every rank holds the graph and runs
its own batches, so per-GPU work is fixed ("weak") and value = all ranks' queries / max time.

Other workloads (--workload flat | ivfpq | spann) time the other §8 rows the same way; only the
default one is the headline.

One JSON line on rank 0 with `roofline` (dominant kernel: algorithmic bytes / HIP-event kernel
time vs 8 TB/s HBM) and `cpu_baseline` (the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
METRIC = "QPS @ recall@10, SIFT-1M d=128 top-10, batch=1/64; 1/2/4/8 GPUs"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="hnsw", choices=["hnsw", "flat", "ivfpq", "spann"])
    p.add_argument("--n", type=int, default=None, help="base vectors (default: the config's size)")
    p.add_argument("--dim", type=int, default=None)
    p.add_argument("--batch", type=int, default=None)
    p.add_argument("--ef", type=int, default=200)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--nprobe", type=int, default=16)
    p.add_argument("--max-neighbors", type=int, default=32)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


class Timer:
    def __init__(self, world):
        self.world = world

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds


def recall_at_k(found_lo, gt_idx, k):
    hits = 0
    for row, g in zip(found_lo, gt_idx):
        hits += len(set(row[:k].tolist()) & set(g[:k].tolist()))
    return hits / (len(gt_idx) * k)


# ------------------------------------------------------------------------------------------ workloads
def sift_base_and_queries(n, d, nq, rank):
    """BASELINE.md C2/C3 synthetic SIFT-1M: base rows (seed 1) and queries drawn from the same
    cluster centres (seed 1000 + rank)."""
    from muopdb_amd import build as B
    ncl = max(1, min(4096, n // 244))
    x = B.sift_like(n, d, n_clusters=ncl, seed=1)
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    centers = (torch.rand((ncl, d), generator=g) * 218.0).cuda()  # first draw of sift_like(seed=1)
    gq = torch.Generator(device="cpu"); gq.manual_seed(1000 + rank)
    qa = torch.randint(0, ncl, (nq,), generator=gq).cuda()
    q = torch.clamp(torch.round(centers[qa] + (torch.randn((nq, d), generator=gq) * 20.0).cuda()), 0, 218).contiguous()
    return x, q


def run_hnsw(args, ctx, rank, world, timer):
    from muopdb_amd import build as B
    from muopdb_amd.index import BlockBasedHnsw
    n = args.n or 1_000_000
    d = args.dim or 128
    batch = args.batch or 64
    k, ef = args.k, args.ef
    steps, warm = args.steps, args.warmup
    t0 = time.time()
    nq = (steps + warm) * batch
    x, queries = sift_base_and_queries(n, d, nq, rank)
    log("data %.1fs" % (time.time() - t0))
    t0 = time.time()
    index_bytes, vec_bytes = B.hnsw_files(x, max_neighbors=args.max_neighbors, max_layers=8, kcand=2 * args.max_neighbors, seed=1)
    log("graph build %.1fs (%d MiB index)" % (time.time() - t0, len(index_bytes) >> 20))
    t0 = time.time()
    hnsw = BlockBasedHnsw(ctx, index_bytes, vec_bytes, d)
    log("load %.1fs" % (time.time() - t0))
    ids = torch.zeros((batch, k, 2), dtype=torch.int64, device="cuda")
    sc = torch.zeros((batch, k), dtype=torch.float32, device="cuda")
    cn = torch.zeros(batch, dtype=torch.int32, device="cuda")

    def step(i, keep=None):
        q = queries[i * batch:(i + 1) * batch]
        hnsw.ann_search_device(q.data_ptr(), batch, k, ef, ids.data_ptr(), sc.data_ptr(), cn.data_ptr())
        if keep is not None:
            keep.append(ids[:, :, 0].clone())

    for i in range(warm):
        step(i)
    ctx.sync()
    ctx.set_profiling(True)
    ctx.get_profile()
    timer.barrier()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        step(i)
    timer.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.get_profile()
    ctx.set_profiling(False)
    elapsed = timer.max_over_ranks(elapsed)
    # untimed re-run of the timed batches: results for recall + exact traversal counters per launch
    found, evals, expanded, abytes = [], 0, 0, 0
    for i in range(warm, warm + steps):
        step(i, found)
        st = ctx.stats()
        evals += st["distance_evals"]; expanded += st["expanded_nodes"]; abytes += st["algorithmic_bytes"]
    found = torch.cat(found).cpu().numpy()
    tq = queries[warm * batch:(warm + steps) * batch]
    gt, _ = B.exact_knn(x, k, queries=tq, f64=True)
    rec = recall_at_k(found, gt.cpu().numpy(), k)
    out = dict(
        value=world * steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=rec,
        config={"workload": "SIFT-1M-like synthetic %dx%d f32 (4096 Gaussian clusters, sigma 20, clipped [0,218]); HNSW "
                            "max_neighbors=%d ef=%d top-%d batch=%d per GPU; replicas" % (n, d, args.max_neighbors, ef, k, batch),
                "n": n, "dim": d, "batch": batch, "ef": ef, "k": k, "index": "hnsw", "parallelism": "replica x%d" % world},
        roofline=dict(bound="hbm", kernel="hnsw_search_kernel",
                      achieved=(abytes / steps) / (kernel_ms / launches * 1e-3) / 1e9 if launches else None,
                      peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
                      bytes_per_launch=abytes / steps, kernel_ms=kernel_ms / max(launches, 1),
                      evals_per_query=evals / (steps * batch), expanded_per_query=expanded / (steps * batch)),
    )
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS if out["roofline"]["achieved"] else None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        o = oracle.BlockBasedHnsw(index_bytes, vec_bytes, d)
        qh = tq.cpu().numpy()
        t0 = time.perf_counter(); o.ann_search(qh[:64], k, ef, threads=1); dt = time.perf_counter() - t0
        ns = int(min(len(qh), max(64, args.cpu_seconds / (dt / 64))))
        t0 = time.perf_counter(); r = o.ann_search(qh[:ns], k, ef, threads=1); dt1 = time.perf_counter() - t0
        ok = all(r.doc_ids(i) == [int(v) for v in found[i][:int(r.counts[i])]] for i in range(min(ns, 256)))
        nt = oracle.num_threads()
        na = min(len(qh), max(ns, 8 * nt))
        t0 = time.perf_counter(); o.ann_search(qh[:na], k, ef, threads=nt); dta = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=ns / dt1, unit="queries/s", cores=1, kind="port",
                                   sample="%d of the timed queries, one thread (the reference runs one query per task, "
                                          "no intra-query parallelism); index fully memory-resident" % ns,
                                   all_cores_value=na / dta, all_cores=nt, ids_match_gpu=bool(ok))
    return out


def run_flat(args, ctx, rank, world, timer):
    """BASELINE config C1 by default (10k x 128, batch 1): py/create_test_hdf5.py-shaped data."""
    from muopdb_amd import build as B
    from muopdb_amd.index import FlatIndex
    n = args.n or 10_000
    d = args.dim or 128
    batch = args.batch or 1
    k = args.k
    steps, warm = args.steps, args.warmup
    g = torch.Generator(device="cpu"); g.manual_seed(42)
    lab = torch.arange(n) % 10
    x = (lab[:, None].float() * 100.0 + torch.randn((n, d), generator=g) * 5.0)
    x = x[torch.randperm(n, generator=g)].cuda().contiguous()
    nq = (steps + warm) * batch
    ql = torch.randint(0, 10, (nq,), generator=g)
    queries = (ql[:, None].float() * 100.0 + torch.randn((nq, d), generator=g) * 5.0).cuda().contiguous()
    # rows are sharded across ranks (SURVEY.md §8e flat: row-range shards); here every rank scans its shard
    lo, hi = rank * n // world, (rank + 1) * n // world
    idx = FlatIndex(ctx, None, device_ptr=x[lo:hi].data_ptr(), n=hi - lo, d=d)
    ids = torch.zeros((batch, k), dtype=torch.int32, device="cuda")
    ds = torch.zeros((batch, k), dtype=torch.float32, device="cuda")

    def step(i):
        idx.search_device(queries[i * batch:(i + 1) * batch].data_ptr(), batch, k, ids.data_ptr(), ds.data_ptr())

    for i in range(warm):
        step(i)
    ctx.sync(); ctx.set_profiling(True); ctx.get_profile()
    timer.barrier()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        step(i)
    timer.barrier()
    elapsed = timer.max_over_ranks(time.perf_counter() - t0)
    kernel_ms, launches = ctx.get_profile(); ctx.set_profiling(False)
    abytes = (hi - lo) * d * 4 + batch * d * 4 + batch * k * 8
    ach = abytes / (kernel_ms / launches * 1e-3) / 1e9
    out = dict(value=steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=1.0,
               config={"workload": "flat brute-force L2 %dx%d f32, batch=%d, top-%d (row-sharded x%d)" % (n, d, batch, k, world),
                       "n": n, "dim": d, "batch": batch, "k": k, "index": "flat"},
               roofline=dict(bound="hbm", kernel="flat_scan_kernel", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=ach / HBM_PEAK_GBS, traffic=None, bytes_per_launch=abytes, kernel_ms=kernel_ms / launches))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        xb, qh = x.cpu().numpy(), queries[warm * batch:].cpu().numpy()
        t0 = time.perf_counter(); oracle.flat_topk(0, xb, qh[:4], k); dt = time.perf_counter() - t0
        ns = int(min(len(qh), max(4, args.cpu_seconds / (dt / 4))))
        t0 = time.perf_counter(); oracle.flat_topk(0, xb, qh[:ns], k); dt1 = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=ns / dt1, unit="queries/s", cores=1, kind="port", sample="%d queries, one thread" % ns)
    return out


def run_ivfpq(args, ctx, rank, world, timer):
    """BASELINE config C3: SIFT-1M-like, IVF nlist=4096 + PQ m=16 (subdim 8) nbits=8, batch 256."""
    from muopdb_amd import build as B, formats as F
    from muopdb_amd.index import BlockBasedIvf, ProductQuantizer
    n = args.n or 1_000_000
    d = args.dim or 128
    batch = args.batch or 256
    k, P = args.k, args.nprobe
    steps, warm = args.steps, args.warmup
    nlist = max(1, min(4096, n // 244))
    nq = (steps + warm) * batch
    x, queries = sift_base_and_queries(n, d, nq, rank)
    t0 = time.time()
    cent = B.kmeans(x, nlist, iters=6, seed=3, sample=min(n, 400_000))
    assign = B.assign_nearest(x, cent)
    cb = B.train_pq_codebook(x, 8, 8, iters=6, seed=4, sample=100_000)
    pq = ProductQuantizer(d, 8, 8, cb)
    codes = pq.quantize(ctx, x.cpu().numpy())
    pls = B.posting_lists_from_assignment(assign, nlist)
    index_bytes = F.write_ivf_index(cent.cpu().numpy(), np.arange(n, dtype=np.uint64), pls, quantized_dimension=d // 8)
    vec_bytes = F.write_vector_file(codes)
    log("ivf-pq build %.1fs" % (time.time() - t0))
    ivf = BlockBasedIvf(ctx, index_bytes, vec_bytes, pq, shard_rank=rank, shard_world=world)
    import ctypes as C
    from muopdb_amd import lib as L
    ids = torch.zeros((batch, k, 2), dtype=torch.int64, device="cuda")
    sc = torch.zeros((batch, k), dtype=torch.float32, device="cuda")
    cn = torch.zeros(batch, dtype=torch.int32, device="cuda")

    def step(i, keep=None):
        q = queries[i * batch:(i + 1) * batch]
        ctx.check(ctx.lib.mdb_ivf_search(ivf.h, C.c_void_p(q.data_ptr()), C.c_size_t(batch), None, C.c_size_t(P), C.c_size_t(k),
                                         C.c_int(L.MEM_DEVICE), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                         C.c_void_p(cn.data_ptr())))
        if keep is not None:
            keep.append(ids[:, :, 0].clone())

    for i in range(warm):
        step(i)
    ctx.sync(); ctx.set_profiling(True); ctx.get_profile()
    timer.barrier()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        step(i)
    timer.barrier()
    elapsed = timer.max_over_ranks(time.perf_counter() - t0)
    kernel_ms, launches = ctx.get_profile(); ctx.set_profiling(False)
    found, scored, abytes = [], 0, 0
    for i in range(warm, warm + steps):
        step(i, found)
        st = ctx.stats(); scored += st["scored_vectors"]; abytes += st["algorithmic_bytes"]
    found = torch.cat(found).cpu().numpy()
    tq = queries[warm * batch:(warm + steps) * batch]
    gt, _ = B.exact_knn(x, k, queries=tq, f64=True)
    rec = recall_at_k(found, gt.cpu().numpy(), k)
    ach = (abytes / steps) / (kernel_ms / launches * 1e-3) / 1e9
    out = dict(value=steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=rec,
               config={"workload": "SIFT-1M-like synthetic %dx%d, IVF nlist=%d + PQ m=16 nbits=8 (symmetric distance), nprobe=%d, "
                                   "batch=%d, top-%d, lists sharded x%d" % (n, d, nlist, P, batch, k, world),
                       "n": n, "dim": d, "batch": batch, "k": k, "nprobe": P, "index": "ivf-pq"},
               roofline=dict(bound="hbm", kernel="ivf_scan_pq_kernel", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=ach / HBM_PEAK_GBS, traffic=None, bytes_per_launch=abytes / steps,
                             kernel_ms=kernel_ms / launches, scored_per_query=scored / (steps * batch)))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        o = oracle.BlockBasedIvf(index_bytes, vec_bytes, oracle.Quant(oracle.QUANT_PQ, oracle.METRIC_L2, 8, 8, cb))
        qh = tq.cpu().numpy()
        t0 = time.perf_counter(); o.search(qh[:32], k, num_probes=P); dt = time.perf_counter() - t0
        ns = int(min(len(qh), max(32, args.cpu_seconds / (dt / 32))))
        t0 = time.perf_counter(); r = o.search(qh[:ns], k, num_probes=P); dt1 = time.perf_counter() - t0
        ok = all(r.doc_ids(i) == [int(v) for v in found[i][:int(r.counts[i])]] for i in range(min(ns, 256)))
        out["cpu_baseline"] = dict(value=ns / dt1, unit="queries/s", cores=1, kind="port", sample="%d queries, one thread" % ns,
                                   ids_match_gpu=bool(ok))
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from muopdb_amd import lib as L
    ctx = L.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    timer = Timer(world)
    res = {"hnsw": run_hnsw, "flat": run_flat, "ivfpq": run_ivfpq}[args.workload](args, ctx, rank, world, timer)
    line = {"metric": METRIC, "value": res.pop("value"), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    line.update(res)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
