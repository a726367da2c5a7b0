cd /root/repo
for cfg in "MDB_BF_QB=4" "MDB_BF_QB=8"; do
env $cfg python bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('$cfg', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
done
