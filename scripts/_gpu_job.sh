cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python scripts/stress_parity.py --seconds 360 --seed 20261001 2>&1 | tail -2
bash scripts/profile_round.sh r3f > gpurun_out/r3f_round.log 2>&1
tail -2 gpurun_out/r3f_round.log
