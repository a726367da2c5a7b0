#!/bin/bash
# Per-kernel breakdown of the C5 per-GPU step (run on the GPU box): build the shard once through bench.py --dump-dir, then
# rocprofv3 --kernel-trace --stats over the torch-free replay under each setting in VARIANTS ("name:ENV=V,ENV=V").
# usage: scripts/c5_breakdown.sh <out tag> [variant ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
DUMP=/tmp/mdb_dump_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ ! -d $DUMP/c5 ]; then
  timeout 900 python $REPO/bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP > $OUT/bench_c5.json 2> $OUT/bench_c5.err
fi
for V in default "$@"; do
  NAME=${V%%:*}; ENVS=""
  [ "$V" != "default" ] && ENVS=$(echo ${V#*:} | tr ',' ' ')
  env $ENVS $REPO/muopdb_amd/replay_search ivfpq $DUMP/c5 128 10 64 4096 6 > $OUT/replay_$NAME.log 2>&1
  rm -rf /tmp/prof_c5_$NAME
  env $ENVS timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5_$NAME -o r -- $REPO/muopdb_amd/replay_search ivfpq $DUMP/c5 128 10 64 4096 6 > $OUT/prof_$NAME.log 2>&1
  cp /tmp/prof_c5_$NAME/*kernel_stats.csv $OUT/kernel_stats_$NAME.csv 2>/dev/null
  echo "== $NAME: $(grep -i 'ms/step\|ms per' $OUT/replay_$NAME.log | head -2)"
  head -12 $OUT/kernel_stats_$NAME.csv | cut -d, -f1-5
done
