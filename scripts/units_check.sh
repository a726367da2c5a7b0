# f32 posting lists in 32-slot units: parity tests that touch f32 lists, then C4 at 128 / 1024 users (step, scan kernel, resident bytes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "ivf or spann or segment or shard or fullsize or c4 or boundary or build or inplace" 2>&1 | tail -8
for u in 128 1024; do
  timeout 600 python bench.py --workload spann --users $u --steps 20 --warmup 5 --no-cpu-baseline --no-sweep >/dev/null 2>/tmp/b.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json')); r=j['roofline']
print('users $u step %.4f ms scan %.4f ms frac %.3f closure %.4f ms hbm %.2f GB file %.2f GB ratio %.3f recall %.4f' % (j['ms_per_step'], r['kernel_ms'], r['frac'], r['centroid_graph']['kernel_ms'], j['hbm_resident_bytes']/1e9, j['file_bytes']/1e9, j['hbm_over_file_bytes'], j['recall_at_10']))
" || tail -5 /tmp/b.err
done
