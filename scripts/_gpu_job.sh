cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ivf or pq" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload ivfpq --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c3.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c3.json') if x.startswith('{')][-1])
print('c3', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('recall_at_10'), [ (x.get('nprobe'), round(x.get('ms_per_step'),4)) for x in j.get('nprobe_sweep',[])])
PY
done
timeout 400 python scripts/stress_parity.py --seconds 240 --seed 5 2>&1 | tail -2
