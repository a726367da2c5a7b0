cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inplace.py tests/test_gpu_traversal.py tests/test_gpu_boundary.py -x -q > gpurun_out/t_last.log 2>&1; tail -2 gpurun_out/t_last.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c4 or spann" > gpurun_out/t_c4.log 2>&1; tail -2 gpurun_out/t_c4.log
