#!/bin/bash
# A/B of whole-library build variants on the GPU box: muopdb_amd/variants/lib_<v>.so (built by hand from csrc/build/*.o with one
# or more translation units recompiled under other flags) are copied over libmuopdb_hip.so in turn and the five non-HNSW
# workloads are timed.  Used for the scheduler-flag sweep of DESIGN 6d (result: max-ilp / no post-RA scheduling only pays for
# mdb_hnsw.hip; max-memory-clause and no-post-RA are neutral on the streaming kernels).
# usage (through gpurun): VARIANTS="a b" bash scripts/flags_ab.sh
for v in ${VARIANTS:-hnswonly all}; do
  cp muopdb_amd/variants/lib_$v.so muopdb_amd/libmuopdb_hip.so
  for args in "--workload flat --n 1000000 --batch 1" "--workload flat --n 1000000 --batch 64" "--workload ivfpq --no-sweep --streams 0" "--workload spann --users 128 --no-sweep" "--workload c5"; do
    timeout 400 python bench.py $args --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] $args'.ljust(60), round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d.get('recall_at_10'))"
  done
done
