#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
timeout 100 python scripts/stress_parity.py --start 3958 --seconds 20 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 700 python scripts/stress_parity.py --start 3000 --seconds 500 2>&1 | tail -3
timeout 400 python scripts/stress_parity.py --only hnsw --seed 5 --seconds 240 2>&1 | tail -3
python bench.py --workload hnsw --streams 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hnsw', round(j['value']), round(j['ms_per_step'],4), j.get('dispersion',{}).get('region_ms_per_step'))"
