cd /root/repo
run() { for v in 0 1; do MDB_HNSW_NO_DUAL=$v python bench.py --workload hnsw --steps 30 --warmup 5 --no-cpu-baseline --streams 0 2>/dev/null | python -c "
import json,sys
j=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('$1 nodual=$v', j['ms_per_step'], j['roofline']['kernel_ms'])"; done; }
run normal
for b in 1 16 256; do for v in 0 1; do MDB_HNSW_NO_DUAL=$v python bench.py --workload hnsw --batch $b --steps 30 --warmup 5 --no-cpu-baseline --streams 0 2>/dev/null | python -c "
import json,sys
j=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('batch $b nodual=$v', j['ms_per_step'], j['roofline']['kernel_ms'])"; done; done
rm -f muopdb_amd/csrc/build/mdb_hnsw.o; MDB_EXTRA_FLAGS=-DMDB_PIPE_DBG bash muopdb_amd/csrc/build.sh > /tmp/b.log 2>&1; tail -1 /tmp/b.log
run dbg
