#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py > gpurun_out/bench_all.json 2> gpurun_out/bench_all.err
echo "bench exit $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_all.json').read().strip().splitlines()[-1])
print(j['metric'], j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('step_frac'))
for k,v in j.get('workloads',{}).items():
    if isinstance(v,dict) and 'value' in v:
        print(k, round(v['value']), round(v['ms_per_step'],4), v.get('roofline',{}).get('frac'), v.get('step_frac'), v.get('seconds'))
    else: print(k, str(v)[:200])
PY
tail -5 gpurun_out/bench_all.err
