#!/bin/bash
# per-kernel durations of the HNSW headline (C2, batch 64, ef 200) with the upper layers on sorted positions (MDB_HNSW_RANK=1)
# and on the register beam (=0): rocprofv3 --kernel-trace --stats over the torch-free replay of the dumped files
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/hnsw_rank; mkdir -p $OUT
DUMP=/tmp/mdb_dump_rank
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/bench.py --workload hnsw --no-cpu-baseline --no-sweep --no-insert-graph --streams 0 --steps 5 --dump-dir $DUMP > $OUT/bench.log 2>&1
for R in ${RANKS:-3 2 0}; do
  rm -rf /tmp/prof_rank_$R
  MDB_HNSW_RANK=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rank_$R -o replay -- $REPO/muopdb_amd/replay_search hnsw $DUMP/hnsw 128 10 ${EF:-200} ${B:-64} 40 > $OUT/replay_$R.log 2>&1
  cp /tmp/prof_rank_$R/*kernel_stats.csv $OUT/kernel_stats_rank$R.csv
  echo "== MDB_HNSW_RANK=$R"; tail -2 $OUT/replay_$R.log; cut -d, -f1-4 $OUT/kernel_stats_rank$R.csv | head -8
done
