cd /root/repo
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python scripts/stress_parity.py --seconds 300 --seed 20261004 2>&1 | tail -2
