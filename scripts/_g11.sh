cd $GRAFT_REPO_ROOT
cat > /tmp/hb.sh <<'X'
python bench.py --workload flat --n 1000000 --batch 64 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f min=%.4f'%(d['ms_per_step'],d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step']['min']))"
X
for i in 1 2; do for r in 0 1; do echo -n "rows $r: "; MDB_FLAT_ROWS=$r bash /tmp/hb.sh; done; done
echo -n "rows 1 slices 4: "; MDB_REFINE_SLICES=4 bash /tmp/hb.sh
echo -n "rows 1 slices 2: "; MDB_REFINE_SLICES=2 bash /tmp/hb.sh
echo -n "rows 1 slices 16: "; MDB_REFINE_SLICES=16 bash /tmp/hb.sh
MDB_MF_DBG=1 python bench.py --workload flat --n 1000000 --batch 64 --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep "\[mf\]" | tail -1
