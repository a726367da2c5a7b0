#!/bin/bash
# block-shared filter with one query block per wave (MDB_BF_BLOCK_QB=1, four blocks per CU): parity tests that reach it, the
# matrix-core stress, then the C5 per-GPU bench line under QB = 1 / 4 / 1
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "large_coarse or block_filter or mfma or flat_batched or c5 or coarse" 2>&1 | tail -4
timeout 600 python scripts/stress_mfma.py --seconds 150 2>&1 | tail -2; timeout 600 python scripts/stress_mfma.py --seconds 150 --coarse 2>&1 | tail -2
for qb in 1 4 1; do
  MDB_BF_BLOCK_QB=$qb timeout 900 python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>/tmp/c5.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json')); r=j['roofline']
print('qb=$qb step %.4f ms scan kernels %.4f ms frac %.3f recall %s' % (j['ms_per_step'], r['kernel_ms'], r['frac'], j.get('recall_at_10')))
" || tail -5 /tmp/c5.err
done
