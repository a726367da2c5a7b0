cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
py() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "value=%.0f ms=%.4f scan=%.4f hnsw=%.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r.get("centroid_hnsw_kernel_ms",0)))
PY
}
python bench.py --workload spann --steps 20 --warmup 3 > gpurun_out/bench_spann_128u.json 2>/dev/null; py gpurun_out/bench_spann_128u.json
python bench.py --workload spann --users 1024 --batch 1024 --steps 20 --warmup 3 > gpurun_out/bench_spann_c4_full.json 2>/dev/null; py gpurun_out/bench_spann_c4_full.json
