cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traversal.py tests/test_gpu_inplace.py tests/test_gpu_boundary.py -x -q -k "noq or spann or multi or segment or filter" 2>&1 | tail -1
cat > /tmp/hb.py <<'X'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms=%.4f kernel=%.4f min=%.4f frac=%.3f'%(d['ms_per_step'],r['kernel_ms'], d['dispersion']['region_ms_per_step']['min'], r['frac']))
X
for ns in 0 1 3 4 5 8 12 16; do echo -n "c4 full nsplit $ns: "; MDB_SCAN_F32_NSPLIT=$ns python bench.py --workload spann --users 1024 --batch 1024 --no-sweep --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python /tmp/hb.py; done
for ns in 0 8 12 16; do echo -n "128u nsplit $ns: "; MDB_SCAN_F32_NSPLIT=$ns python bench.py --workload spann --no-sweep --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python /tmp/hb.py; done
