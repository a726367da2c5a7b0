#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traversal.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in full_64; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o c5 -- python $R/bench.py --workload c5 --steps 8 --warmup 2 --n 16000000 --no-cpu-baseline > /tmp/log_$v 2>&1
  f=$(find /tmp/prof_$v -name '*kernel_trace.csv' | head -1)
  echo "== $v"
  python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
g=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if any(x in n for x in ('flat_bf16','sample_bound','flat_refine_group')):
        g[(n[:52], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in sorted(g.items()):
    v.sort(); print(k, len(v), 'min %.1f med %.1f max %.1f'%(v[0], v[len(v)//2], v[-1]))
PY
done
cd $R
timeout 300 python scripts/stress_mfma.py --seconds 100 2>&1 | tail -2
python bench.py --workload flat --n 1000000 --batch 1024 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flat1024', round(j['value']), round(j['ms_per_step'],4), j['roofline'].get('kernel_ms'))"
