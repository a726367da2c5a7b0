cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ivf" > gpurun_out/t_ivf.log 2>&1; tail -5 gpurun_out/t_ivf.log
MDB_PQF_DBG=1 timeout 300 python bench.py --workload ivfpq --no-cpu-baseline --no-sweep --streams 0 --steps 5 --warmup 2 2>&1 | grep "\[pqf\]" | tail -2
timeout 300 python bench.py --workload ivfpq --no-cpu-baseline --streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f recall=%s'%(d['ms_per_step'],d['roofline']['kernel_ms'],d.get('recall_at_10'))); print({k:(round(v['ms_per_step'],4) if isinstance(v,dict) and 'ms_per_step' in v else None) for k,v in d.items() if isinstance(v,dict)}); print(d.get('nprobe_sweep'))"
