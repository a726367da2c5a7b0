cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_traversal.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2_hnsw" 2>&1 | tail -3
for a in "" "--batch 1" "--ef 400"; do
python bench.py --workload hnsw $a --streams 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), d['ms_per_step'], d['recall_at_10'], d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step'])"
done
MDB_HNSW_NO_TABLE=1 python bench.py --workload hnsw --streams 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no_table', round(d['value']), d['ms_per_step'], d['recall_at_10'], d['roofline']['kernel_ms'])"
