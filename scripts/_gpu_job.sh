#!/bin/bash
# scratch job of the moment (gpurun runs it from the repo root)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
bash scripts/profile_round.sh r04d c5 c5full
