cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -k "ivf" > gpurun_out/t_ivf.log 2>&1; tail -3 gpurun_out/t_ivf.log
cat > /tmp/hb.sh <<'X'
python bench.py --workload ivfpq --no-cpu-baseline --streams 0 --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f min=%.4f'%(d['ms_per_step'],d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step']['min']))"
X
bash scripts/ab_run.sh 2 'bash /tmp/hb.sh'
MDB_PQF_DBG=1 timeout 300 python bench.py --workload ivfpq --no-cpu-baseline --no-sweep --streams 0 --steps 5 --warmup 2 2>&1 | grep "\[pqf\]" | tail -2
