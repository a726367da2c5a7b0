#!/bin/bash
# cycle sums of the sorted-position upper traversal (ab/ built with -DMDB_PIPE_DBG on mdb_hnsw_upper.hip only), per batch of 64:
# words: [3] | t_lookup t_have t_accept t_pop - steps nnew surv accepted sort_cycles total_cycles -     (MDB_HNSW_DBG=1: layer-1 launch, =2: top launch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp muopdb_amd/libmuopdb_hip.so /tmp/lib_a.so; cp ab/libmuopdb_hip.so muopdb_amd/libmuopdb_hip.so
for D in 1 2; do
  echo "MDB_HNSW_DBG=$D"
  MDB_HNSW_DBG=$D python bench.py --workload hnsw --streams 0 --no-cpu-baseline --no-sweep --no-insert-graph --steps 4 --warmup 1 2>&1 | grep "hnsw dbg" | tail -2
done
cp /tmp/lib_a.so muopdb_amd/libmuopdb_hip.so
