cd /root/repo
export MDB_BENCH_BACKEND=gloo MDB_BENCH_DEVICE=0
(time timeout 900 python bench.py --gpus 2 --workload c5full --base-n 4000000 --nlist 4096 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/g2_c5.json 2> gpurun_out/g2_c5.err) 2>&1 | tail -3
tail -4 gpurun_out/g2_c5.err; tail -c 1500 gpurun_out/g2_c5.json; echo
unset MDB_BENCH_BACKEND MDB_BENCH_DEVICE
python bench.py --workload c5full --base-n 4000000 --nlist 4096 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1gpu', d['value'], d['ms_per_step'])"
