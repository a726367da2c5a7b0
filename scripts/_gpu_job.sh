cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c1 or flat or c5" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload flat --n 1000000 --batch 64 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_f.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_f.json') if x.startswith('{')][-1])
print('flat b64', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'])
PY
done
timeout 300 python scripts/stress_mfma.py --seconds 150 --seed 21 2>&1 | tail -1
