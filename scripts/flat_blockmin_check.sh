#!/bin/bash
# from which batch the block-shared filter (one base pass per 128 QB queries, fragments handed round a block through LDS) beats the
# per-wave streaming one: flat 1 M x 128, MDB_BF_BLOCK_MIN_B x MDB_BF_BLOCK_QB
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for b in 128 256 512; do for v in "512 0" "128 1" "128 2" "128 4"; do
  set -- $v
  MDB_BF_BLOCK_MIN_B=$1 MDB_BF_BLOCK_QB=$2 timeout 600 python bench.py --workload flat --n 1000000 --batch $b --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>/tmp/f.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json'))
print('batch=$b block_min_b=$1 qb=$2 step %.4f ms recall %s' % (j['ms_per_step'], j.get('recall_at_10')))
" || tail -3 /tmp/f.err
done; done
