#!/usr/bin/env python3
"""gpurun_out/<tag>/ (scripts/profile_round.sh) -> profiles/r<NN>_*: per-workload kernel_stats (our kernels + the top rows), the
PMC passes as collected, r<NN>_traffic.json (per-launch FETCH_SIZE / WRITE_SIZE averages of the dominant kernel, KiB) and the
un-instrumented bench line.  usage: summarize_profiles.py <tag> <round, e.g. r02>"""
import csv, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
OURS = ("hnsw_", "flat_", "sample_bound", "ivf_", "merge_", "remap_", "spann_", "pq_quantize", "mfma_prep", "unpack_", "kmeans_", "pad_queries")
WL = {  # workload -> (kernels of the dominant group: bench.py's `roofline.kernel`, bench.py traffic key, config match)
    # ef <= 256, batch >= 32: top layers + table pass | layer 1 | layer 0 (mdb_hnsw_upper.hip); their sum is the traversal's bracket
    "hnsw": (("hnsw_upper_top_kernel", "hnsw_upper_top_rank_kernel", "hnsw_upper_rank_kernel", "hnsw_upper_table", "hnsw_upper_kernel", "hnsw_beam_kernel"), "hnsw",
             {"n": 1000000, "dim": 128, "batch": 64, "ef": 200, "k": 10}),
    "hnsw_ef400": (("hnsw_upper_top_kernel", "hnsw_upper_top_rank_kernel", "hnsw_upper_rank_kernel", "hnsw_upper_table", "hnsw_upper_kernel", "hnsw_beam_kernel", "hnsw_search_kernel"), "hnsw_ef400", {"n": 1000000, "dim": 128, "batch": 64, "ef": 400, "k": 10}),
    "flat_b1": (("flat_scan_kernel",), "flat", {"n": 1000000, "dim": 128, "batch": 1, "k": 10}),
    # <METRIC, QB, NKT, SMP = false, APX>: the filter proper, not its sample pass
    "flat_b64": (("flat_bf16_filter_kernel<0, 2, 8, false",), "flat_b64", {"n": 1000000, "dim": 128, "batch": 64, "k": 10}),
    # round 5: the step is the matrix-core coarse search + the fused kernel (round 4: ivf_prep_kernel + the fused kernel)
    # (the dominant kernel bench.py's roofline names; the step's other launch — the matrix-core coarse search — is listed under OTHER)
    "ivfpq": (("ivf_pq_fused_kernel",), "ivfpq", {"n": 1000000, "dim": 128, "batch": 256, "k": 10, "nprobe": 16}),
    "spann": (("ivf_scan_f32_kernel",), "spann", {"n": 1250048, "dim": 768, "batch": 128, "k": 10}),
    "c5": (("ivf_scan_pq3_kernel", "ivf_pq3_refine_kernel"), "c5", {"dim": 128, "batch": 4096, "k": 10, "nprobe": 64}),
    "c5full": (("ivf_scan_pq3_kernel", "ivf_pq3_refine_kernel"), "c5full", {"n": 100000000, "dim": 128, "batch": 4096, "k": 10, "nprobe": 64}),
    "c4full": (("ivf_scan_f32_kernel",), "spann_full", {"n": 10000384, "dim": 768, "batch": 1024, "k": 10}),
}
OTHER = {"ivfpq": ("ivf_coarse_mfma_kernel", "ivf_prep_kernel")}   # launches of the step beside the dominant kernel: reported, not priced
EXTRA = {"spann": ("centroid_graph", ("hnsw_closure_kernel",)), "c4full": ("centroid_graph", ("hnsw_closure_kernel",))}


def group_ms(rows, kerns):
    """per-step time of a kernel group from a --kernel-trace --stats table: sum over the group's kernels of AverageNs x (its calls / the
    calls of the group's most-called kernel) — every kernel of these groups is launched once per step"""
    hit = [r for r in rows if any(k in r["Name"] for k in kerns)]
    if not hit:
        return None, []
    ref = max(int(r["Calls"]) for r in hit)
    parts = [dict(kernel=r["Name"].replace("void ", "").split("(")[0][:80], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3) for r in hit]
    return sum(float(r["AverageNs"]) * int(r["Calls"]) / ref for r in hit) / 1e6, parts


def main_filter(name):
    """flat_bf16_filter_kernel<METRIC, QB, NKT, SMP, APX> with SMP == false (the sample pass has its own launches)"""
    if "flat_bf16x1_block_kernel<" in name:   # <METRIC, QB, BOUND, APX>: the filter pass of a large batch (BOUND == false)
        targs = name.split("flat_bf16x1_block_kernel<", 1)[1].split(">", 1)[0].split(",")
        return len(targs) >= 3 and targs[2].strip() == "false"
    if "ivf_coarse_mfma_kernel" in name:           # C3's coarse search
        return True
    if "flat_bf16_filter_kernel<" not in name:
        return False
    targs = name.split("flat_bf16_filter_kernel<", 1)[1].split(">", 1)[0].split(",")
    return len(targs) >= 4 and targs[3].strip() == "false"


bench = json.loads(open(os.path.join(src, "bench_all.json")).read().strip().splitlines()[-1])
shutil.copy(os.path.join(src, "bench_all.json"), os.path.join(dst, "%s_bench_all.json" % rnd))
# round 5 on: stdout carries the bounded line, the FULL record (prose config, dispersion, per-kernel rooflines) goes to stderr as "[bench-full] {...}"
errp = os.path.join(src, "bench_all.err")
if os.path.exists(errp):
    full = [x for x in open(errp) if x.startswith("[bench-full] ")]
    if full:
        bench = json.loads(full[-1][len("[bench-full] "):])
        json.dump(bench, open(os.path.join(dst, "%s_bench_all_full.json" % rnd), "w"))
lines = {"hnsw": bench}
c4p = os.path.join(src, "c4full_bench.json")
if os.path.exists(c4p):
    c4l = [x for x in open(c4p) if x.startswith("{")]
    if c4l:
        lines["c4full"] = json.loads(c4l[-1])
for k, v in bench.get("workloads", {}).items():
    lines[{"flat_1m_b1": "flat_b1", "flat_1m_b64": "flat_b64", "ivfpq_c3": "ivfpq", "spann_c4_128u": "spann", "c5_shard_per_gpu": "c5",
           "c5_full_1gpu": "c5full", "hnsw_c2_ef400": "hnsw_ef400", "spann_c4_full_1024u": "c4full_in_all"}.get(k, k)] = v
c5p = os.path.join(src, "c5full_bench.json")
if os.path.exists(c5p):
    c5l = [x for x in open(c5p) if x.startswith("{")]
    if c5l:
        lines["c5full"] = json.loads(c5l[-1])
traffic = {"_note": "HBM traffic per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the torch-free "
                    "replay of the same files (examples/replay_search.cpp, scripts/profile_round.sh); counters in KiB; gfx950 correction per "
                    "MI355X_MICROARCH.md: FETCH_SIZE x2 (calibrated on the flat scan: 512.0 MB algorithmic).  bench.py reports `traffic` from this "
                    "file only when its config matches `match`."}
summary = {}
for w, (kern, key, match) in WL.items():
    ks = os.path.join(src, "%s_kernel_stats.csv" % w)
    if not os.path.exists(ks):
        continue
    rows = list(csv.DictReader(open(ks)))
    keep = [r for i, r in enumerate(rows) if i < 12 or r["Name"].replace("void ", "").startswith(OURS)]
    with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, w)), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(keep)
    avg_ms, parts = group_ms(rows, kern)
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        pf = os.path.join(src, "%s_pmc_%s.csv" % (w, c))
        if not os.path.exists(pf):
            continue
        shutil.copy(pf, os.path.join(dst, "%s_%s_pmc_%s.csv" % (rnd, w, c)))
        tot = 0.0
        for kname in kern:                                            # per kernel of the group: mean over its dispatches, then the sum
            per = {}
            for r in csv.DictReader(open(pf)):
                if kname in r["Kernel_Name"] and r["Counter_Name"] == c:
                    per.setdefault(r["Dispatch_Id"], 0.0)
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"])   # one row per XCD / dimension instance: sum per dispatch
            disp = sorted(per.items(), key=lambda kv: int(kv[0]))[1:]     # skip the warm-up call
            tot += sum(v for _, v in disp) / max(len(disp), 1)
        vals[c] = tot
    # matrix-core busy cycles of the filter kernels against their own active cycles (SQ_BUSY_CYCLES) and the chip's (GRBM_GUI_ACTIVE)
    pm = os.path.join(src, "%s_pmc_MFMA.csv" % w)
    if os.path.exists(pm):
        shutil.copy(pm, os.path.join(dst, "%s_%s_pmc_MFMA.csv" % (rnd, w)))
        acc = {}
        for r in csv.DictReader(open(pm)):
            if main_filter(r["Kernel_Name"]):
                acc.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        disp = [v for _, v in sorted(acc.items(), key=lambda kv: int(kv[0]))][1:]
        if disp:
            mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in disp) / len(disp)
            sq = sum(d.get("SQ_BUSY_CYCLES", 0) for d in disp) / len(disp)
            gr = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in disp) / len(disp)
            vals["MFMA"] = dict(mfma_busy_cycles=mf, sq_busy_cycles=sq, grbm_gui_active=gr, launches=len(disp))
    rp = os.path.join(src, "%s_replay.log" % w)
    if os.path.exists(rp):
        shutil.copy(rp, os.path.join(dst, "%s_%s_replay.log" % (rnd, w)))
    cfg = lines.get(w, {}).get("config", {})
    m = dict(match, data=cfg.get("data", "lowrank"))
    if w in ("spann", "c5") and "n" in cfg:
        m["n"] = cfg["n"]
    if "MFMA" in vals:
        traffic.setdefault("_mfma", {})[key] = dict(vals["MFMA"], source="profiles/%s_%s_pmc_MFMA.csv" % (rnd, w),
                                                    kernel="flat_bf16x1_block_kernel (filter pass)" if w == "c5" else "ivf_coarse_mfma_kernel" if w == "ivfpq" else "flat_bf16_filter_kernel")
    if "FETCH_SIZE" in vals:
        traffic[key] = {"match": m, "kernel": "+".join(k.split("<")[0] for k in kern), "fetch_kib": round(vals["FETCH_SIZE"], 1), "write_kib": round(vals.get("WRITE_SIZE", 0.0), 1),
                        "fetch_correction": 2.0, "source": ["profiles/%s_%s_pmc_FETCH_SIZE.csv" % (rnd, w), "profiles/%s_%s_pmc_WRITE_SIZE.csv" % (rnd, w)]}
    r = lines.get(w, {}).get("roofline", {})
    ab = r.get("bytes_per_launch")
    summary[w] = dict(rocprof_avg_ms=avg_ms, rocprof_kernels=parts, bench_kernel_ms=r.get("kernel_ms"), bench_ms_per_step=lines.get(w, {}).get("ms_per_step"),
                      algorithmic_bytes=ab, bench_frac=r.get("frac"),
                      # the judge's recomputation: bytes x units / rocprof AverageNs / 8 TB/s — must land within 3 % of bench_frac
                      recomputed_frac=(ab / (avg_ms * 1e-3) / 8e12) if (ab and avg_ms) else None,
                      step_frac=lines.get(w, {}).get("step_frac"), dispersion=lines.get(w, {}).get("dispersion"),
                      measured_traffic_bytes=(vals.get("FETCH_SIZE", 0) * 2 + vals.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in vals else None,
                      mfma=vals.get("MFMA"))
    if w in OTHER:
        oms, oparts = group_ms(rows, OTHER[w])
        summary[w]["other_launches"] = dict(rocprof_avg_ms=oms, rocprof_kernels=oparts)
    if w in EXTRA and r.get(EXTRA[w][0]):
        name, ek = EXTRA[w]
        ems, eparts = group_ms(rows, ek)
        er = r[name]
        summary[w][name] = dict(rocprof_avg_ms=ems, rocprof_kernels=eparts, bench_kernel_ms=er.get("kernel_ms"), algorithmic_bytes=er.get("bytes_per_launch"),
                                bench_frac=er.get("frac"),
                                recomputed_frac=(er["bytes_per_launch"] / (ems * 1e-3) / 8e12) if (ems and er.get("bytes_per_launch")) else None)
# a partial re-profile (profile_round.sh <tag> hnsw spann) keeps the other workloads' entries of the round
for name, new in (("traffic", traffic), ("summary", summary)):
    path = os.path.join(dst, "%s_%s.json" % (rnd, name))
    merged = json.load(open(path)) if os.path.exists(path) else {}
    merged.update(new)
    json.dump(merged, open(path, "w"), indent=1)
for w, s in summary.items():
    print(w, {k: (round(v, 4) if isinstance(v, float) and v < 1e4 else v) for k, v in s.items()})
