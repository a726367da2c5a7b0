#!/bin/bash
# cycle sums of the upper-layer traversal (ab/ built with -DMDB_PIPE_DBG on mdb_hnsw_upper.hip only): per batch of 64,
# words: [3] | t_sel t_wait t_accept t_push t_pop steps nnew surv na na1 na3 compactions
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp muopdb_amd/libmuopdb_hip.so /tmp/lib_a.so; cp ab/libmuopdb_hip.so muopdb_amd/libmuopdb_hip.so
MDB_HNSW_DBG=1 python bench.py --workload hnsw --streams 0 --no-cpu-baseline --steps 4 --warmup 1 2>&1 | grep "hnsw dbg" | tail -3
cp /tmp/lib_a.so muopdb_amd/libmuopdb_hip.so
