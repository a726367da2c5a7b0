cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_traversal.py -x -q -m gpu -k "hnsw" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "hnsw or c2" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload hnsw --steps 30 --warmup 5 --no-cpu-baseline --streams 0 2>/dev/null > gpurun_out/r3_h.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_h.json') if x.startswith('{')][-1])
print('spec', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('recall_at_10'))
PY
done
