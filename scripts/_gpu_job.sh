cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for w in c5 flat; do
python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_$w.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_$w.json') if x.startswith('{')][-1])
print('$w', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'), j['config'].get('workload'))
PY
done
