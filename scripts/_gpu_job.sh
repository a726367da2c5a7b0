cd /root/repo
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2_hnsw" 2>&1 | tail -3
for b in 64 1 256; do
python bench.py --workload hnsw --batch $b --streams 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value']), d['ms_per_step'], d['recall_at_10'], d['roofline']['kernel_ms'])"
done
