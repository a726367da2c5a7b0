cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_traversal.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2_hnsw" 2>&1 | tail -3
for ef in 200 400; do
python bench.py --workload hnsw --ef $ef --streams 0 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ef $ef', round(d['value']), d['ms_per_step'], d['recall_at_10'], d['roofline']['kernel_ms'], d['roofline']['evals_per_query'], d['roofline']['expanded_per_query'])"
done
