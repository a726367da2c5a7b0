#!/bin/bash
# A/B helper (run on the GPU box): the scan-bound workloads, one line each
cd ${GRAFT_REPO_ROOT:-/root/repo}
py() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "value=%.0f ms=%.4f kernel_ms=%.4f" % (d["value"], d["ms_per_step"], r["kernel_ms"]), "ids_match", d["cpu_baseline"].get("ids_match_gpu"))
PY
}
python bench.py --workload ivfpq > gpurun_out/q_ivfpq_c3.json 2>/dev/null; py gpurun_out/q_ivfpq_c3.json
python bench.py --workload ivfpq --n 12500000 --nlist 8192 --nprobe 8 --batch 4096 --steps 10 --warmup 2 > gpurun_out/q_c5.json 2>/dev/null; py gpurun_out/q_c5.json
python bench.py --workload flat --n 1000000 > gpurun_out/q_flat_b1.json 2>/dev/null; py gpurun_out/q_flat_b1.json
python bench.py --workload flat --n 1000000 --batch 64 > gpurun_out/q_flat_b64.json 2>/dev/null; py gpurun_out/q_flat_b64.json
python bench.py --workload flat > gpurun_out/q_flat_c1.json 2>/dev/null; py gpurun_out/q_flat_c1.json
python bench.py --workload spann --steps 20 --warmup 3 > gpurun_out/q_spann.json 2>/dev/null; py gpurun_out/q_spann.json
