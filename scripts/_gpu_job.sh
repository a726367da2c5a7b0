cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p2; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o o -- python /root/repo/scripts/_ovf_check.py > /tmp/p2.log 2>&1; grep -c "^ok" /tmp/p2.log
python - <<'PY'
import csv, glob
f=glob.glob('/tmp/p2/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(x in r['Name'] for x in ('flat_refine_group','flat_bf16_filter','flat_scan_kernel','merge_keys')): print(r['Name'][:50], r['Calls'])
PY
cd /root/repo
for i in 1 2; do
python bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_c5.json
python - <<PY
import json
j=json.loads([x for x in open('gpurun_out/r3_c5.json') if x.startswith('{')][-1])
print('c5', round(j['value']), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('rank_of_8_step',{}).get('ms_per_step'))
PY
done
