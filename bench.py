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

# The `concurrent` extra of the HNSW workload keeps 4 batches in flight on 4 HIP streams; with the runtime's default
# of 4 hardware queues two of them share a queue with torch's own stream and only 2 batches overlap (117 k
# queries/s); 8 queues give every stream its own (235 k queries/s).  No effect on the single-stream headline.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
METRIC = "QPS @ recall@10, SIFT-1M d=128 top-10, batch=1/64; 1/2/4/8 GPUs"


def measured_traffic(kind, cfg):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r*_traffic.json; rocprofv3 --pmc cannot run inside this process), only when the
    workload matches the profiled one; else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        with open(path) as f:
            e = json.load(f).get(kind)
        if e and all(cfg.get(k) == v for k, v in e["match"].items()):
            return (e["fetch_kib"] * e["fetch_correction"] + e["write_kib"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def dump(args, rank, **files):
    """--dump-dir: the workload's files for examples/replay_search.cpp (torch-free PMC passes)."""
    if not args.dump_dir or rank != 0:
        return
    os.makedirs(args.dump_dir, exist_ok=True)
    for name, data in files.items():
        with open(os.path.join(args.dump_dir, name), "wb") as f:
            f.write(data if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data).tobytes())
    log("dumped %s to %s" % (", ".join(files), args.dump_dir))


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
    p.add_argument("--nlist", type=int, default=None, help="ivfpq: number of posting lists (default min(4096, n/244))")
    p.add_argument("--users", type=int, default=128, help="spann workload: number of users (1024 = full C4)")
    p.add_argument("--max-neighbors", type=int, default=32)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--streams", type=int, default=4, help="hnsw: extra measurement with this many batches in flight (0/1 = skip)")
    p.add_argument("--dump-dir", default=None, help="write index files + queries for examples/replay_search.cpp")
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
    dump(args, rank, index=index_bytes, vectors=vec_bytes, **{"queries.f32": queries.cpu().numpy()})
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
        roofline=dict(bound="hbm", kernel="hnsw_beam_kernel" if ef <= 256 else "hnsw_search_kernel",
                      achieved=(abytes / steps) / (kernel_ms / launches * 1e-3) / 1e9 if launches else None,
                      peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
                      bytes_per_launch=abytes / steps, kernel_ms=kernel_ms / max(launches, 1),
                      evals_per_query=evals / (steps * batch), expanded_per_query=expanded / (steps * batch)),
    )
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS if out["roofline"]["achieved"] else None
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("hnsw", out["config"])
    if args.streams > 1:
        # Extra, NOT the headline: the same K batches issued round-robin on several HIP streams (one context +
        # attached index handle each, all over the same resident graph), i.e. several batches of 64 in flight.  One batch occupies 64 of the 256 CUs for
        # its whole latency-bound traversal, so a serving process overlaps batches to fill the chip.
        from muopdb_amd import lib as L
        lanes = []
        for _ in range(args.streams):
            st_ = torch.cuda.Stream()
            c_ = L.Context(torch.cuda.current_device()); c_.set_stream(st_.cuda_stream)
            # one resident index, one handle per stream over it (mdb_hnsw_attach)
            lanes.append((st_, c_, hnsw.attach(c_), torch.zeros_like(ids), torch.zeros_like(sc), torch.zeros_like(cn)))
        torch.cuda.synchronize()

        def cstep(i):
            _, _, h_, i_, s_, c_n = lanes[i % len(lanes)]
            q = queries[i * batch:(i + 1) * batch]
            h_.ann_search_device(q.data_ptr(), batch, k, ef, i_.data_ptr(), s_.data_ptr(), c_n.data_ptr())

        for i in range(warm):
            cstep(i)
        timer.barrier()
        t0 = time.perf_counter()
        for i in range(warm, warm + steps):
            cstep(i)
        timer.barrier()
        el = timer.max_over_ranks(time.perf_counter() - t0)
        same = True
        for j, i in enumerate(range(warm + steps - len(lanes), warm + steps)):  # last batch of every lane vs the serial run
            same &= bool(torch.equal(lanes[i % len(lanes)][3][:, :, 0].cpu(), torch.from_numpy(found[(i - warm) * batch:(i - warm + 1) * batch])))
        out["concurrent"] = dict(streams=args.streams, value=world * steps * batch / el, ms_per_step=1000 * el / steps,
                                 ids_equal_serial=same, note="same batches, several in flight on attached handles over one resident index "
                                      "(GPU_MAX_HW_QUEUES=%s); not the headline value" % os.environ.get("GPU_MAX_HW_QUEUES"))
        for lane_ in lanes:
            lane_[2].close(); lane_[1].close()
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
    nq = (steps + warm) * batch
    if n >= 100_000:  # "flat SIFT-1M" of the north star: the C2/C3 synthetic SIFT-like base
        x, queries = sift_base_and_queries(n, d, nq, rank)
        x = x.contiguous()
    else:             # C1: py/create_test_hdf5.py-shaped data
        g = torch.Generator(device="cpu"); g.manual_seed(42)
        lab = torch.arange(n) % 10
        x = (lab[:, None].float() * 100.0 + torch.randn((n, d), generator=g) * 5.0)
        x = x[torch.randperm(n, generator=g)].cuda().contiguous()
        ql = torch.randint(0, 10, (nq,), generator=g)
        queries = (ql[:, None].float() * 100.0 + torch.randn((nq, d), generator=g) * 5.0).cuda().contiguous()
    # rows are sharded across ranks (SURVEY.md §8e flat: row-range shards); here every rank scans its shard
    lo, hi = rank * n // world, (rank + 1) * n // world
    idx = FlatIndex(ctx, None, device_ptr=x[lo:hi].data_ptr(), n=hi - lo, d=d)
    if args.dump_dir:
        from muopdb_amd import formats as F
        dump(args, rank, vectors=F.write_vector_file(x.cpu().numpy()), **{"queries.f32": queries.cpu().numpy()})
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
    # recall@k of the last timed batch against the f64 brute force (untimed re-run of that batch)
    from muopdb_amd import build as B
    last = warm + steps - 1
    step(last)
    gt, _ = B.exact_knn(x[lo:hi], k, queries=queries[last * batch:(last + 1) * batch], f64=True)
    rec = recall_at_k(ids.cpu().numpy().astype(np.int64), gt.cpu().numpy(), k)
    batched = batch >= 8 and (hi - lo) >= 65536  # mdb_flat_mfma.hip: sample bound + MFMA filter + exact refine
    out = dict(value=steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=rec,
               config={"workload": "flat brute-force L2 %dx%d f32 (%s), batch=%d, top-%d (row-sharded x%d)"
                                   % (n, d, "SIFT-1M-like synthetic" if n >= 100_000 else "create_test_hdf5-like", batch, k, world),
                       "n": n, "dim": d, "batch": batch, "k": k, "index": "flat"},
               roofline=dict(bound="hbm", kernel="flat_mfma_filter_kernel" if batched else "flat_scan_kernel", achieved=ach,
                             peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None, bytes_per_launch=abytes,
                             kernel_ms=kernel_ms / launches))
    if batched:  # the filter is also on the f32-MFMA ridge: 2*B*N*d flop per launch against 157.3 TFLOP/s
        groups = (batch + 63) // 64
        out["roofline"]["bytes_per_launch"] = abytes * groups  # one pass over the base per 64 queries
        out["roofline"]["achieved"] = ach * groups
        out["roofline"]["frac"] = ach * groups / HBM_PEAK_GBS
        out["roofline"]["mfma_tflops"] = 2.0 * batch * (hi - lo) * d / (kernel_ms / launches * 1e-3) / 1e12
        out["roofline"]["mfma_frac_of_f32_peak"] = out["roofline"]["mfma_tflops"] / 157.3
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("flat_b64" if batched else "flat", out["config"])
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
    nlist = args.nlist or max(1, min(4096, n // 244))
    nq = (steps + warm) * batch
    x, queries = sift_base_and_queries(n, d, nq, 0)  # lists are sharded: every rank sees the SAME batch
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
    dump(args, rank, index=index_bytes, vectors=vec_bytes, **{"queries.f32": queries.cpu().numpy(), "codebook.f32": cb})
    import ctypes as C
    from muopdb_amd import lib as L
    from muopdb_amd import distributed as D
    ids = torch.zeros((batch, k, 2), dtype=torch.int64, device="cuda")
    sc = torch.zeros((batch, k), dtype=torch.float32, device="cuda")
    cn = torch.zeros(batch, dtype=torch.int32, device="cuda")

    def step(i, keep=None):
        q = queries[i * batch:(i + 1) * batch]
        probes = None
        if world > 1:  # the coarse quantizer is sharded too: 1/world of the centroids per rank + one all-gather of (distance, id) rows
            probes = D.sharded_probes(ctx, ivf, q.data_ptr(), batch, P, q.device)
        ctx.check(ctx.lib.mdb_ivf_search(ivf.h, C.c_void_p(q.data_ptr()), C.c_size_t(batch),
                                         C.c_void_p(probes.data_ptr()) if probes is not None else None, C.c_size_t(P), C.c_size_t(k),
                                         C.c_int(L.MEM_DEVICE), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                         C.c_void_p(cn.data_ptr())))
        res = ids
        if world > 1:  # one RCCL all-gather of the per-shard top-k + device merge
            gd, gs, gc = D.all_gather_topk(ids, sc, cn)
            res, _, _ = D.merge_shards_device(ctx, gd, gs, gc)
        if keep is not None:
            keep.append(res[:, :, 0].clone())

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
    nrec = min(len(tq), 12800, max(256, int(1.3e10 // n)))  # f64 ground truth for a bounded number of the timed queries
    gt, _ = B.exact_knn(x, k, queries=tq[:nrec], f64=True)
    rec = recall_at_k(found[:nrec], gt.cpu().numpy(), k)
    ach = (abytes / steps) / (kernel_ms / launches * 1e-3) / 1e9
    out = dict(value=steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=rec, scaling="strong",
               config={"workload": "SIFT-1M-like synthetic %dx%d, IVF nlist=%d + PQ m=16 nbits=8 (symmetric distance), nprobe=%d, "
                                   "batch=%d, top-%d, lists sharded x%d" % (n, d, nlist, P, batch, k, world),
                       "n": n, "dim": d, "batch": batch, "k": k, "nprobe": P, "index": "ivf-pq"},
               roofline=dict(bound="hbm", kernel="ivf_scan_pq2_kernel", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=ach / HBM_PEAK_GBS, traffic=None, bytes_per_launch=abytes / steps,
                             kernel_ms=kernel_ms / launches, scored_per_query=scored / (steps * batch)))
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("ivfpq", out["config"])
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

def run_spann(args, ctx, rank, world, timer):
    """BASELINE.md C4 shape: multi-user SPANN over unit-norm f32 rows, one (user, query) pair per user
    per batch, posting lists sharded l % world, one all-gather + merge per batch.  Defaults are a
    1/8 slice (128 users x 9766 x 768 = 3.8 GB); --users 1024 is the full 10M x 768 (30.7 GB)."""
    from muopdb_amd import build as B
    from muopdb_amd import formats as F
    from muopdb_amd import distributed as D
    from muopdb_amd.index import MultiSpannIndex, SearchParams
    import ctypes as C
    from muopdb_amd import lib as L
    U = args.users
    per = (args.n // U) if args.n else 9766
    d = args.dim or 768
    batch = args.batch or U
    k, P = args.k, args.nprobe
    steps, warm = args.steps, args.warmup
    nlist = max(1, per // 64)
    t0 = time.time()
    users, base = {}, []
    for u in range(U):
        x = B.unit_gaussian(per, d, seed=3_000_000 + u)
        cent = B.kmeans(x, nlist, iters=4, seed=u)
        pls = B.posting_lists_from_assignment(B.assign_nearest(x, cent), cent.shape[0])
        hi, hv = B.hnsw_files(cent, max_neighbors=16, max_layers=4, kcand=32, seed=u)
        docs = np.arange(u * per, (u + 1) * per, dtype=np.uint64)
        users[u + 1] = dict(hnsw_index=hi, hnsw_vectors=hv, ivf_index=F.write_ivf_index(cent.cpu().numpy(), docs, pls),
                            ivf_vectors=F.write_vector_file(x.cpu().numpy()))
        base.append(x)
    cat = F.concat_multi_spann(users)
    del users
    log("multi-user SPANN build: %d users x %d x %d, %.1fs, ivf_vectors %.2f GB" % (U, per, d, time.time() - t0,
                                                                                    len(cat["ivf_vectors"]) / 1e9))
    t0 = time.time()
    ms = MultiSpannIndex(ctx, cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"],
                         None, rank, world)
    log("load %.1fs" % (time.time() - t0))
    nq = (steps + warm) * batch
    gq = torch.Generator(device="cpu"); gq.manual_seed(77)          # same pairs on every rank (lists are sharded)
    quser = (torch.arange(nq) % U)
    qrow = torch.randint(0, per, (nq,), generator=gq)
    noise = torch.randn((nq, d), generator=gq) * (0.3 / d ** 0.5)
    queries = torch.stack([base[int(u)][int(r)] for u, r in zip(quser.tolist(), qrow.tolist())]) + noise.cuda()
    queries = (queries / queries.norm(dim=1, keepdim=True)).contiguous()
    if args.dump_dir:
        dump(args, rank, hnsw_index=cat["hnsw_index"], hnsw_vectors=cat["hnsw_vectors"], ivf_index=cat["ivf_index"],
             vectors=cat["ivf_vectors"], user_table=cat["user_table"],
             **{"queries.f32": queries.cpu().numpy(), "users.u64": (quser + 1).numpy().astype(np.uint64)})
    params = SearchParams(k, args.ef).with_num_explored_centroids(P).with_centroid_distance_ratio(0.1).to_c()
    ids = torch.zeros((batch, k, 2), dtype=torch.int64, device="cuda")
    sc = torch.zeros((batch, k), dtype=torch.float32, device="cuda")
    cn = torch.zeros(batch, dtype=torch.int32, device="cuda")
    fo = torch.zeros(batch, dtype=torch.uint8, device="cuda")
    uid_arrays = [L.u128_array([int(u) + 1 for u in quser[i * batch:(i + 1) * batch].tolist()]) for i in range(steps + warm)]

    def step(i, keep=None):
        q = queries[i * batch:(i + 1) * batch]
        ctx.check(ctx.lib.mdb_multi_spann_search(ms.h, uid_arrays[i], C.c_void_p(q.data_ptr()), C.c_size_t(batch), C.byref(params),
                                                 C.c_int(L.MEM_DEVICE), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                 C.c_void_p(cn.data_ptr()), C.c_void_p(fo.data_ptr())))
        res = ids
        if world > 1:
            gd, gs, gc = D.all_gather_topk(ids, sc, cn)
            res, _, _ = D.merge_shards_device(ctx, gd, gs, gc)
        if keep is not None:
            keep.append(res[:, :, 0].clone())

    for i in range(warm):
        step(i)
    ctx.sync(); ctx.set_profiling(1); ctx.get_profile()  # 1: posting-list scan only
    timer.barrier()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        step(i)
    timer.barrier()
    elapsed = timer.max_over_ranks(time.perf_counter() - t0)
    kernel_ms, launches = ctx.get_profile(); ctx.set_profiling(False)
    found, scored, abytes = [], 0, 0
    ctx.set_profiling(2); ctx.get_profile()  # untimed re-run: centroid-graph traversal kernel time
    for i in range(warm, warm + steps):
        step(i, found)
        st = ctx.stats(); scored += st["scored_vectors"]; abytes += st["algorithmic_bytes"]
    hnsw_ms, hnsw_launches = ctx.get_profile(); ctx.set_profiling(False)
    found = torch.cat(found).cpu().numpy()
    # exact per-user ground truth (f64) for recall
    hits = 0
    for j in range(steps * batch):
        qi = warm * batch + j
        u = int(quser[qi])
        xb = base[u].double()
        dd = ((xb - queries[qi].double()[None, :]) ** 2).sum(1)
        gt = (torch.topk(dd, k, largest=False).indices + u * per).cpu().numpy()
        hits += len(set(found[j][:k].tolist()) & set(gt.tolist()))
    rec = hits / (steps * batch * k)
    ach = (abytes / steps) / (kernel_ms / launches * 1e-3) / 1e9
    out = dict(value=steps * batch / elapsed, ms_per_step=1000 * elapsed / steps, recall_at_10=rec, scaling="strong",
               config={"workload": "multi-user SPANN, %d users x %d x %d f32 unit-norm (BASELINE.md C4 shape), batch=%d (user,query) "
                                   "pairs, ef=%d, num_explored_centroids=%d, ratio=0.1, top-%d, posting lists sharded x%d"
                                   % (U, per, d, batch, args.ef, P, k, world),
                       "users": U, "n": U * per, "dim": d, "batch": batch, "k": k, "nprobe": P, "index": "multi-spann"},
               roofline=dict(bound="hbm", kernel="ivf_scan_f32_kernel", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=ach / HBM_PEAK_GBS, traffic=None, bytes_per_launch=abytes / steps,
                             kernel_ms=kernel_ms / launches, scored_per_query=scored / (steps * batch),
                             centroid_hnsw_kernel_ms=hnsw_ms / max(hnsw_launches, 1)))
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("spann", out["config"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        o = oracle.MultiSpannIndex(cat["user_table"], d, cat["hnsw_index"], cat["hnsw_vectors"], cat["ivf_index"], cat["ivf_vectors"])
        op = oracle.SearchParams(k, args.ef, num_explored_centroids=P, centroid_distance_ratio=0.1)
        qh = queries[warm * batch:(warm + steps) * batch].cpu().numpy()
        uh = [int(u) + 1 for u in quser[warm * batch:(warm + steps) * batch].tolist()]
        t0 = time.perf_counter(); o.search_for_user(uh[:16], qh[:16], op); dt = time.perf_counter() - t0
        ns = int(min(len(qh), max(16, args.cpu_seconds / (dt / 16))))
        t0 = time.perf_counter(); r = o.search_for_user(uh[:ns], qh[:ns], op); dt1 = time.perf_counter() - t0
        ok = all(r.doc_ids(i) == [int(v) for v in found[i][:int(r.counts[i])]] for i in range(min(ns, 256)))
        out["cpu_baseline"] = dict(value=ns / dt1, unit="queries/s", cores=1, kind="port", sample="%d (user,query) pairs, one thread" % ns,
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
    res = {"hnsw": run_hnsw, "flat": run_flat, "ivfpq": run_ivfpq, "spann": run_spann}[args.workload](args, ctx, rank, world, timer)
    line = {"metric": METRIC, "value": res.pop("value"), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True, "scaling": res.pop("scaling", "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    line.update(res)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
