#!/bin/bash
# posting-list scans that skip the tombstone / allow words when nothing was invalidated and no filter is given (default) vs
# MDB_SCAN_MASKS_ALWAYS=1: tests that invalidate / filter, then C5 per-GPU, C4 128 users and the full C4, alternated on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "tombstone or invalidat or filter or planner or segment or spann or ivf" 2>&1 | tail -2
for wl in "--workload c5 --steps 6 --warmup 2" "--workload spann --users 128 --no-sweep --steps 20 --warmup 5" "--workload spann --users 1024 --no-sweep --steps 6 --warmup 2"; do
  for v in 0 1 0 1; do
    MDB_SCAN_MASKS_ALWAYS=$v timeout 900 python bench.py $wl --no-cpu-baseline > /dev/null 2>/tmp/n.err
    python -c "
import json
j=json.load(open('gpurun_out/bench_full.json')); r=j['roofline']
print('$wl | masks_always=$v step %.4f ms scan %.4f ms' % (j['ms_per_step'], r['kernel_ms']))
" || tail -3 /tmp/n.err
  done
done
