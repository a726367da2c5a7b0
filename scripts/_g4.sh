cd $GRAFT_REPO_ROOT
MDB_PQF_DBG=1 timeout 300 python bench.py --workload ivfpq --no-cpu-baseline --no-sweep --streams 0 --steps 5 --warmup 2 2>&1 | grep "\[pqf\]" | tail -4
