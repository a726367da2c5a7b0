cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge_coarse_keys_rows" 2>&1 | tail -15
