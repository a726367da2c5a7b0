#!/bin/bash
# one-block-per-query refine with 512-key blocks behind the whole-base bound (MDB_REFINE_GROUP_BIG=0, default) vs the 2048-key blocks:
# parity tests that reach it, then same-box replays under rocprofv3 and the C5 per-GPU bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "large_coarse or block_filter or mfma or flat_batched or c5 or coarse" 2>&1 | tail -3
timeout 600 python scripts/stress_mfma.py --seconds 120 --coarse 2>&1 | tail -1
bash scripts/c5_breakdown.sh ab_rg big:MDB_REFINE_GROUP_BIG=1 small:MDB_REFINE_GROUP_BIG=0 bigb:MDB_REFINE_GROUP_BIG=1 2>&1 | grep "^=="
for v in 0 1 0; do
  MDB_REFINE_GROUP_BIG=$v timeout 900 python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>/tmp/c5.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json')); r=j['roofline']
print('big=$v step %.4f ms' % (j['ms_per_step']))
" || tail -5 /tmp/c5.err
done
