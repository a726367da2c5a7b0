# same-box comparison of the C3 step's coarse-search variants (run through gpurun)
cd /root/repo
for v in "MDB_CM_SPLIT=0 MDB_CM_GLOBAL_BOUND=1" "MDB_CM_SPLIT=1 MDB_CM_GLOBAL_BOUND=1" "MDB_CM_SPLIT=1 MDB_CM_GLOBAL_BOUND=0" "MDB_IVF_COARSE_MFMA=0"; do
for r in 1 2; do env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ivfpq --no-sweep --streams 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['roofline']['kernel_ms'], j['recall_at_10'])"; done
done
cd /tmp && export TMPDIR=/tmp; MDB_CM_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ivfpq --no-sweep --streams 0 >/dev/null 2>&1; grep -E "ivf_|Name" /tmp/ps/*kernel_stats.csv | cut -c1-200 | head
