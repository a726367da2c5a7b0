#!/bin/bash
# flat 1M x 128 batch 1: flat_scan_kernel's average duration vs MDB_FLAT_BLOCKS (rocprofv3 over the torch-free replay), plus the
# asynchronous host path of the HNSW workload (replay hnsw vs hnsw-async).
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
DUMP=/tmp/mdb_dump_fs
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/bench.py --workload flat --n 1000000 --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-dir $DUMP > $OUT/bench_flat_b1.json 2> $OUT/bench_flat_b1.err
for B in 0 256 1024 2048 4096; do
  rm -rf /tmp/prof_fs_$B
  E=""; [ $B != 0 ] && E="MDB_FLAT_BLOCKS=$B"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fs_$B -o r -- $REPO/muopdb_amd/replay_search flat $DUMP/flat_b1 128 10 0 1 40 > $OUT/prof_$B.log 2>&1
  echo "blocks $B: $(grep -E 'flat_scan_kernel|merge_keys' /tmp/prof_fs_$B/*kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-30 | tr '\n' ' ') $(grep -E 'flat_scan_kernel|merge_keys' /tmp/prof_fs_$B/*kernel_stats.csv | awk -F'",' '{print $2}' | cut -d, -f3 | tr '\n' ' ') $(grep ms/step $OUT/prof_$B.log | sed 's/.*queries, //;s/ (host.*//')"
done
timeout 300 python $REPO/bench.py --workload hnsw --steps 6 --warmup 2 --no-cpu-baseline --streams 0 --dump-dir $DUMP > $OUT/bench_hnsw.json 2> $OUT/bench_hnsw.err
$REPO/muopdb_amd/replay_search hnsw $DUMP/hnsw 128 10 200 64 40 | tee $OUT/replay_hnsw_sync.log
for L in 2 4 8; do GPU_MAX_HW_QUEUES=8 $REPO/muopdb_amd/replay_search hnsw-async $DUMP/hnsw 128 10 200 64 80 $L | tee $OUT/replay_hnsw_async_$L.log; done
