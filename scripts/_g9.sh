cd $GRAFT_REPO_ROOT
cat > /tmp/hb.sh <<'X'
python bench.py --workload flat --n 1000000 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f min=%.4f'%(d['ms_per_step'],d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step']['min']))"
X
for i in 1 2; do for nb in 1024 1536 2048 3072 4096; do echo -n "blocks $nb: "; MDB_FLAT_BLOCKS=$nb bash /tmp/hb.sh; done; done
