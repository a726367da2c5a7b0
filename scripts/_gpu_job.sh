cd $GRAFT_REPO_ROOT; timeout 600 python scripts/hnsw_phase_balance.py 2>&1 | tail -3
