#!/bin/bash
# block-shared filter pass, query blocks per wave (MDB_BF_BLOCK_QB) against the size of the base: flat n x 128 at batch 1024
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for n in 100000 250000 1000000; do for qb in 1 4 2; do
  MDB_BF_BLOCK_QB=$qb timeout 600 python bench.py --workload flat --n $n --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>/tmp/f.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json'))
print('n=$n qb=$qb step %.4f ms' % (j['ms_per_step']))
" || tail -3 /tmp/f.err
done; done
