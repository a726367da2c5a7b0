#!/bin/bash
# same-box alternation, full C4: f32 posting lists padded to 1 | 2 | 4 units of 16 slots (MDB_IVF_LIST_PAD_UNITS; 4 = the 64-slot tiles of rounds 1-5)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_units; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DUMP=/tmp/mdb_dump_ab
timeout 900 python $REPO/bench.py --workload spann --users 1024 --no-sweep --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP/c4full --dump-big > $OUT/c4full_bench.json 2> $OUT/c4full_bench.err
for rep in 1 2 3; do for V in 1 4 2; do
  rm -rf /tmp/prof_ab
  MDB_IVF_LIST_PAD_UNITS=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o r -- $REPO/muopdb_amd/replay_search mspann $DUMP/c4full/spann 768 10 16 1024 10 200 > $OUT/c4_${V}_${rep}.log 2>&1
  python - <<PY
import csv,glob
for f in glob.glob('/tmp/prof_ab/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ivf_scan_f32' in r['Name'] or 'hnsw_closure' in r['Name']:
            print("pad_units=$V rep=$rep %-40s calls %s avg %.1f us" % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
done; done
rm -rf $DUMP/c4full
