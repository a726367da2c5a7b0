#!/bin/bash
# Every bench.py workload once (run on the GPU box); JSON lines land in gpurun_out/bench_<name>.json
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
python bench.py                                                  > $OUT/bench_hnsw_c2.json      2> $OUT/bench_hnsw_c2.err
python bench.py --workload flat                                  > $OUT/bench_flat_c1.json      2> $OUT/bench_flat_c1.err
python bench.py --workload flat --n 1000000                      > $OUT/bench_flat_1m_b1.json   2> $OUT/bench_flat_1m_b1.err
python bench.py --workload flat --n 1000000 --batch 64           > $OUT/bench_flat_1m_b64.json  2> $OUT/bench_flat_1m_b64.err
python bench.py --workload ivfpq                                 > $OUT/bench_ivfpq_c3.json     2> $OUT/bench_ivfpq_c3.err
python bench.py --workload spann --steps 20 --warmup 3           > $OUT/bench_spann_128u.json   2> $OUT/bench_spann_128u.err
for f in $OUT/bench_*.json; do echo "== $f"; tail -c 400 $f; echo; done
