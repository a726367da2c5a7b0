cd /root/repo
for a in "--workload hnsw --streams 0" "--workload hnsw --batch 1 --streams 0" "--workload ivfpq --no-sweep --streams 0" "--workload flat --n 1000000 --batch 64"; do
python bench.py $a --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['dispersion']['region_ms_per_step'])"
done
