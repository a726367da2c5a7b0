#!/bin/bash
# Round-end refresh (run on the GPU box): GPU test suite, every bench line, the rocprofv3 summaries.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
bash scripts/run_all_benches.sh > gpurun_out/run_all.log 2>&1
python bench.py --workload spann --users 1024 --batch 1024 --steps 20 --warmup 3 > gpurun_out/bench_spann_c4_full.json 2> gpurun_out/bench_spann_c4_full.err
python bench.py --workload ivfpq --n 12500000 --nlist 8192 --nprobe 8 --batch 4096 --steps 10 --warmup 2 > gpurun_out/bench_ivfpq_c5shard.json 2> gpurun_out/bench_ivfpq_c5shard.err
python bench.py --batch 256 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_hnsw_b256.json 2> gpurun_out/bench_hnsw_b256.err
python bench.py --workload ivfpq --n 12500000 --nlist 65536 --nprobe 64 --batch 4096 --steps 6 --warmup 2 > gpurun_out/bench_ivfpq_c5gpu.json 2> gpurun_out/bench_ivfpq_c5gpu.err
bash scripts/profile_bench.sh hnsw hnsw 128 10 200 64 > gpurun_out/prof_hnsw.log 2>&1
bash scripts/profile_bench.sh ivfpq ivfpq 128 10 16 256 --workload ivfpq > gpurun_out/prof_ivfpq.log 2>&1
bash scripts/profile_bench.sh spann mspann 768 10 16 128 --workload spann --steps 20 --warmup 3 > gpurun_out/prof_spann.log 2>&1
for f in gpurun_out/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("/")[-1], "value=%.0f ms=%.4f kernel_ms=%.4f frac=%.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"]), r.get("centroid_hnsw_kernel_ms",""))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
