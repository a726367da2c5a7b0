cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_traversal.py -x -q -k "partitionings" 2>&1 | tail -5
# one GPU, two gloo ranks: validates the multi-rank logic of every partitioning (never a bench result); a port per run, a watchdog per rank
export MDB_BENCH_DEVICE=0 MDB_BENCH_BACKEND=gloo MDB_BENCH_WATCHDOG=200
PORT=29517
for a in "--workload spann --users 64 --shard users --no-sweep" "--workload spann --users 64 --shard lists --no-sweep" "--workload ivfpq --shard batch --no-sweep --base-n 200000" "--workload c5full --base-n 3000000 --shard batch" "--workload c5full --base-n 3000000 --shard lists"; do
  PORT=$((PORT+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --streams 0 $a 2>/tmp/mr.err | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    j=json.loads(l); print('$a'.split('--')[1:3], j['value'], j['ms_per_step'], j.get('shard'), j.get('rows_equal_unsharded'), j.get('rows_equal_across_ranks'), j.get('exchange'), j.get('recall_at_10'))
except Exception as e:
    print('FAILED', '$a', l[:300]); print(open('/tmp/mr.err').read()[-1500:])
"
done
