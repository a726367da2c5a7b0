cd /root/repo
( time python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err ) 2>&1 | tail -3
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_final_bench.json') if x.startswith('{')][-1])
print('hnsw', round(j['value']), round(j['ms_per_step'],4), j['roofline']['frac'], j.get('cpu_baseline',{}).get('value'))
for k,v in j['workloads'].items(): print(k, v.get('error') or (round(v['value']), round(v['ms_per_step'],4), v.get('recall_at_10')))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
