#!/bin/bash
# PMC passes over the C5 replay for ivf_scan_pq2_kernel: where do its cycles go (VALU / LDS / waiting)?  One rocprofv3 run per counter
# group (never combined with sys / runtime tracing).  usage: scripts/c5_pmc.sh <tag> "<counters group 1>" "<group 2>" ...
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
DUMP=/tmp/mdb_dump_c5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
if [ ! -d $DUMP/c5 ]; then
  timeout 900 python $REPO/bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP > $OUT/bench_c5.json 2> $OUT/bench_c5.err
fi
i=0
for G in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc_c5_$i
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/pmc_c5_$i -o r -- $REPO/muopdb_amd/replay_search ivfpq $DUMP/c5 128 10 64 4096 3 > $OUT/pmc_$i.log 2>&1
  echo "rc=$? group: $G" | tee -a $OUT/pmc_$i.log
  for f in /tmp/pmc_c5_$i/*counter_collection.csv; do [ -f "$f" ] && (head -1 $f; grep -E "ivf_scan_pq|ivf_pq3|flat_refine" $f) > $OUT/pmc_$i.csv; done
  python3 - $OUT/pmc_$i.csv <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k=r["Kernel_Name"][:28]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc: print(k, len(n[k]), {c: round(v/len(n[k])) for c,v in acc[k].items()})
except Exception as e: print("ERR", e)
PY
done
