cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
bash scripts/profile_round.sh r04f > gpurun_out/profile_round.log 2>&1; tail -15 gpurun_out/profile_round.log
