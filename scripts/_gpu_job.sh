cd /root/repo
(time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5_full" 2>&1 | tail -15) 2>&1
(time timeout 1200 python bench.py --workload c5full --steps 6 --warmup 2 > gpurun_out/c5full.json 2> gpurun_out/c5full.err) 2>&1 | tail -3
tail -5 gpurun_out/c5full.err; tail -c 2500 gpurun_out/c5full.json
