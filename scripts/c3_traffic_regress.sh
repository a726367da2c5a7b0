REPO=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/mdb_dump_c3
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --workload ivfpq --no-sweep --streams 0 --no-cpu-baseline --steps 20 --warmup 5 --dump-dir $D > /dev/null 2>&1
for P in 1 4 16 64; do
    rm -rf /tmp/pc3
    MDB_PQ_SDC_MAX_MB=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pc3 -o r -- $REPO/muopdb_amd/replay_search ivfpq $D/ivfpq 128 10 $P 256 20 > /tmp/pc3.log 2>&1
    grep -i "scored\|bytes" /tmp/pc3.log | head -3
    python3 - "$P" <<'PY'
import csv, glob, sys
by = {}
for f in glob.glob("/tmp/pc3/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        if "ivf_" in n:
            by.setdefault(n, []).append(float(r["Counter_Value"]))
print("P", sys.argv[1], {k: round(sum(v[2:]) / max(1, len(v[2:])), 1) for k, v in by.items()})
PY
done
