# list-sharded multi-user SPANN with the closure run once per pair (mdb_multi_spann_probes / _search_shard_probes):
# the parity test, two gloo ranks on ONE GPU with the closure shared / replicated (logic only, never a bench result),
# and the closure kernel's time against the batch (what `--share-closure auto` keys on)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_traversal.py -x -q -k "probe_rows or multi_spann" 2>&1 | tail -5
export MDB_BENCH_DEVICE=0 MDB_BENCH_BACKEND=gloo MDB_BENCH_WATCHDOG=200
PORT=29617
for a in "--share-closure on" "--share-closure off"; do
  PORT=$((PORT+7)); sleep 2
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --streams 0 --workload spann --users 64 --shard lists --no-sweep $a 2>/tmp/mr.err | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    j=json.loads(l); print('$a', j['value'], j['ms_per_step'], j.get('closure'), j.get('rows_equal_across_ranks'), j.get('exchange'), j.get('recall_at_10'))
except Exception as e:
    print('FAILED', '$a', l[:300]); print(open('/tmp/mr.err').read()[-1500:])
"
done
unset MDB_BENCH_DEVICE MDB_BENCH_BACKEND MDB_BENCH_WATCHDOG
for b in 16 64 128 256 512 1024; do
  timeout 300 python bench.py --workload spann --users 128 --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-sweep >/dev/null 2>/tmp/b.err
  python -c "
import json
j=json.load(open('gpurun_out/bench_full.json')); r=j['roofline']
print('batch $b step %.4f ms scan %.4f ms closure %.4f ms' % (j['ms_per_step'], r['kernel_ms'], r['centroid_graph']['kernel_ms']))
" || tail -5 /tmp/b.err
done
