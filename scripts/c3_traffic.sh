#!/bin/bash
# C3 fused step: where do the L2 misses (FETCH_SIZE) come from?  FETCH passes over the replay under a few settings (run through gpurun)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/mdb_dump_c3
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --workload ivfpq --no-sweep --streams 0 --no-cpu-baseline --steps 20 --warmup 5 --dump-dir $D > /dev/null 2>&1
for v in "X=1" "MDB_PQ_SDC_MAX_MB=0" "MDB_IVF_COARSE_MFMA=0" "MDB_PQF_TILES=1"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc3
    env $v timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc3 -o r -- $REPO/muopdb_amd/replay_search ivfpq $D/ivfpq 128 10 16 256 20 > /tmp/pc3.log 2>&1
    python3 - "$v" $C <<'P'
import csv, glob, sys
by = {}
for f in glob.glob("/tmp/pc3/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        if "ivf_" in n:
            by.setdefault(n, []).append(float(r["Counter_Value"]))
print(sys.argv[1], sys.argv[2], {k: round(sum(v[2:]) / max(1, len(v[2:])), 1) for k, v in by.items()})
P
  done
done
