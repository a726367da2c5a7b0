cd $GRAFT_REPO_ROOT
py() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "value=%.0f ms=%.4f scan=%.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"]), d.get("recall_at_10"), d["cpu_baseline"].get("ids_match_gpu"), r.get("scored_per_query"))
PY
}
python bench.py --workload ivfpq --n 12500000 --nlist 8192 --nprobe 64 --batch 1024 --steps 10 --warmup 2 > gpurun_out/c5n64_f.json 2>/dev/null; py gpurun_out/c5n64_f.json
MDB_PQ_NO_FILTER=1 python bench.py --workload ivfpq --n 12500000 --nlist 8192 --nprobe 64 --batch 1024 --steps 10 --warmup 2 > gpurun_out/c5n64_nf.json 2>/dev/null; py gpurun_out/c5n64_nf.json
