cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o b -- python $REPO/bench.py --workload spann --users 128 --no-sweep --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
python3 - <<'P'
import csv, glob
for f in glob.glob("/tmp/ps/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("hnsw_closure", "ivf_scan_f32", "spann_filter", "merge_sorted", "remap_kernel")):
            print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
P
