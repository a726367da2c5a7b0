#!/bin/bash
# C1 (10 k x 128, batch 1): rocprofv3 kernel durations of the flat step per variant (MDB_FLAT_NO_SMALL = 0 two launches, 4 one launch with block tickets, 2 unordered groups, 1 general kernel)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in ${VARIANTS:-0 4 1}; do
  rm -rf /tmp/prof_c1_$V
  MDB_FLAT_NO_SMALL=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1_$V -o b -- python $REPO/bench.py --workload flat --n 10000 --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_$V.log 2>&1
  cp /tmp/prof_c1_$V/*kernel_stats.csv $OUT/kernel_stats_$V.csv
  echo "== MDB_FLAT_NO_SMALL=$V"; grep -o '"ms_per_step":[0-9.e-]*' $OUT/bench_$V.log | head -1
  python3 - $OUT/kernel_stats_$V.csv <<'PY'
import csv,sys
for row in list(csv.DictReader(open(sys.argv[1])))[:6]:
    if int(row["Calls"]) > 100: print("  %-70s %6s %8.2f us" % (row["Name"][:70], row["Calls"], float(row["AverageNs"])/1e3))
PY
done
