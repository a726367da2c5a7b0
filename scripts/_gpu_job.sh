cd /root/repo
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
