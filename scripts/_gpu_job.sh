#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c5 or coarse" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in full_64 smp_64; do
  case $v in smp*) export MDB_BF_NO_FULL_BOUND=1;; *) unset MDB_BF_NO_FULL_BOUND;; esac
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
unset MDB_BF_NO_FULL_BOUND
cd $R
run() { # label, env, args
  env $2 python bench.py $3 2> gpurun_out/err_$1.log | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j.get('rank_of_8_step') or {}
print('$1', round(j['value']), round(j['ms_per_step'],4), j['roofline'].get('kernel_ms'), round(j['roofline']['frac'],4), j.get('dispersion',{}).get('region_ms_per_step',{}).get('median'), r.get('ms_per_step'), r.get('scan_kernel_ms'))
"
}
run c5_30M X=1 "--workload c5 --steps 8 --warmup 2 --n 30000000"
run flat256 X=1 "--workload flat --n 1000000 --batch 256"
run flat1024 X=1 "--workload flat --n 1000000 --batch 1024"
run flat1024_smp MDB_BF_NO_FULL_BOUND=1 "--workload flat --n 1000000 --batch 1024"
run flat1024_wave MDB_BF_BLOCK_MIN_B=100000000 "--workload flat --n 1000000 --batch 1024"
