#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flat or coarse or large" 2>&1 | tail -3
run() { # label, env, args
  env $2 python bench.py $3 2> gpurun_out/err_$1.log | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(j['value']), round(j['ms_per_step'],4), j['roofline'].get('kernel_ms'), round(j['roofline']['frac'],4), j.get('dispersion',{}).get('region_ms_per_step',{}).get('median'))
"
}
run flat64 X=1 "--workload flat --n 1000000 --batch 64"
run flat32 X=1 "--workload flat --n 1000000 --batch 32"
run flat128 X=1 "--workload flat --n 1000000 --batch 128"
run flat256 X=1 "--workload flat --n 1000000 --batch 256"
run flat512 X=1 "--workload flat --n 1000000 --batch 512"
run flat512w MDB_BF_BLOCK_MIN_B=100000000 "--workload flat --n 1000000 --batch 512"
run flat256blk MDB_BF_BLOCK_MIN_B=256 "--workload flat --n 1000000 --batch 256"
