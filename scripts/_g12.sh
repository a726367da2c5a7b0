cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inplace.py -x -q -k "flat" > gpurun_out/t_flat.log 2>&1; tail -3 gpurun_out/t_flat.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "flat or c1" > gpurun_out/t_flat2.log 2>&1; tail -3 gpurun_out/t_flat2.log
timeout 600 python scripts/stress_mfma.py --help 2>&1 | head -5
