cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 800 python scripts/stress_parity.py --seconds 600 --seed 20260930 2>&1 | tail -3
timeout 400 python scripts/stress_mfma.py --seconds 240 --seed 11 2>&1 | tail -2
timeout 400 python scripts/stress_mfma.py --coarse --seconds 240 --seed 12 2>&1 | tail -2
bash scripts/profile_round.sh r3d > gpurun_out/r3d_round.log 2>&1
tail -2 gpurun_out/r3d_round.log
