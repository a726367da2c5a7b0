cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ivf or pq" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or c3" 2>&1 | tail -3
python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('c5', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p1; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o bench -- python /root/repo/bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/p1.log 2>&1; f=$(ls /tmp/p1/*kernel_stats.csv | head -1); cp $f /root/repo/gpurun_out/r3_c5_kernel_stats.csv
