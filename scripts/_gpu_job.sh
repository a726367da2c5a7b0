cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or c3" 2>&1 | tail -2
python bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('c5', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
MDB_PQ_NO_QUANTIZE8=1 python bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('c5 old quantize', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
timeout 300 python scripts/stress_parity.py --seconds 200 --seed 8 2>&1 | tail -1
