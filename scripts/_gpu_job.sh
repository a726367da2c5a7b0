cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
python bench.py --workload spann --steps 40 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null > gpurun_out/r3_s.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_s.json') if x.startswith('{')][-1])
print('spann128', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('recall_at_10'))
PY
done
timeout 400 python scripts/stress_parity.py --seconds 240 --seed 123 2>&1 | tail -1
