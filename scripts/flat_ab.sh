#!/bin/bash
# A/B of two builds of the library on the flat 1M x 128 batch-64 workload: per-kernel stats of the torch-free replay.
# usage: scripts/flat_ab.sh <out tag> <other lib.so>
TAG=$1; OTHER=$2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
DUMP=/tmp/mdb_dump_flat
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/bench.py --workload flat --n 1000000 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --dump-dir $DUMP > $OUT/bench_flat.json 2> $OUT/bench_flat.err
ls $DUMP
for V in cur other; do
  P=""; [ $V = other ] && P=$OTHER
  rm -rf /tmp/prof_flat_$V
  LD_PRELOAD=$P timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flat_$V -o r -- $REPO/muopdb_amd/replay_search flat $DUMP/flat_b64 128 10 0 64 20 > $OUT/prof_$V.log 2>&1
  cp /tmp/prof_flat_$V/*kernel_stats.csv $OUT/kernel_stats_$V.csv 2>/dev/null
  echo == $V; tail -2 $OUT/prof_$V.log; head -6 $OUT/kernel_stats_$V.csv | cut -c1-60,200-
done
