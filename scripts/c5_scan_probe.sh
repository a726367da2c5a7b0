#!/bin/bash
# ivf_scan_pq2_kernel's average duration on the C5 shard at several nprobe values (fixed cost vs per-vector cost)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
DUMP=/tmp/mdb_dump_c5
cd /tmp && export TMPDIR=/tmp
if [ ! -d $DUMP/c5 ]; then
  timeout 900 python $REPO/bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP > $OUT/bench_c5.json 2> $OUT/bench_c5.err
fi
for P in "$@"; do
  rm -rf /tmp/prof_c5_p$P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5_p$P -o r -- $REPO/muopdb_amd/replay_search ivfpq $DUMP/c5 128 10 $P 4096 6 > $OUT/prof_p$P.log 2>&1
  cp /tmp/prof_c5_p$P/*kernel_stats.csv $OUT/kernel_stats_p$P.csv 2>/dev/null
  echo "nprobe $P: $(grep ivf_scan $OUT/kernel_stats_p$P.csv | awk -F'",' '{print $2}' | cut -d, -f1-3)  $(grep 'ms/step' $OUT/prof_p$P.log)"
done
