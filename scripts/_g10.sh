cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inplace.py -x -q > gpurun_out/t_inplace.log 2>&1; tail -3 gpurun_out/t_inplace.log
python bench.py --workload flat --n 1000000 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f min=%.4f frac=%.3f'%(d['ms_per_step'],d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step']['min'], d['roofline']['frac']))"
