cd /root/repo
MDB_HNSW_DBG=1 python bench.py --workload hnsw --steps 5 --warmup 2 --no-cpu-baseline --streams 0 2>&1 | grep "hnsw dbg" | tail -2
