cd /root/repo
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c2_hnsw" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in 0 1; do
rm -rf /tmp/ps_$v
MDB_HNSW_NO_SPLIT=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$v -o bench -- python $R/bench.py --workload hnsw --streams 0 --no-cpu-baseline --steps 30 --warmup 5 > $R/gpurun_out/ps_$v.log 2>&1
cp /tmp/ps_$v/*kernel_stats.csv $R/gpurun_out/ps_${v}_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/ps_${v}_stats.csv")):
    if "hnsw" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us=%.1f" % (float(r["AverageNs"])/1e3), "min=%.1f max=%.1f" % (float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
grep -o '"value": [0-9.]*, "unit": "queries/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' $R/gpurun_out/ps_$v.log
done
for b in 64 256; do
python $R/bench.py --workload hnsw --batch $b --streams 0 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', round(d['value']), d['ms_per_step'], d['recall_at_10'], d['dispersion']['region_ms_per_step'])"
done
