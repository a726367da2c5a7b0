cd $GRAFT_REPO_ROOT
cat > /tmp/hb.sh <<'X'
python bench.py --workload ivfpq --no-cpu-baseline --streams 0 --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms=%.4f kernel=%.4f min=%.4f'%(d['ms_per_step'],d['roofline']['kernel_ms'], d['dispersion']['region_ms_per_step']['min']))"
X
for i in 1 2; do echo -n "default: "; bash /tmp/hb.sh; echo -n "quant_in_prep: "; MDB_PQF_QUANT_IN_PREP=1 bash /tmp/hb.sh; done
