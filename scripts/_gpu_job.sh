# the round's last verification job (run on the GPU box: gpurun -- 'bash scripts/_gpu_job.sh')
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python scripts/stress_parity.py --seconds 240 2>&1 | tail -2
timeout 300 python scripts/stress_mfma.py --seconds 60 2>&1 | tail -1
