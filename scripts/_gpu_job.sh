cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python scripts/stress_parity.py --seconds 180 --seed 4242 2>&1 | tail -1
