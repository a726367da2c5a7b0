cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coarse or sharded" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5" 2>&1 | tail -3
python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('c5', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
