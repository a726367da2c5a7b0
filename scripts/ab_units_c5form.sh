#!/bin/bash
# same-box alternations: (a) f32 list scan, whole tiles on the constant-stride loader (default) vs the runtime-stride one
# (MDB_SCAN_F32_UNIT_ONLY=1), full C4; (b) the block-shared filter pass of the C5 coarse search under MDB_BF_BLOCK_FORM=0..3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_units; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DUMP=/tmp/mdb_dump_ab
timeout 900 python $REPO/bench.py --workload spann --users 1024 --no-sweep --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP/c4full --dump-big > $OUT/c4full_bench.json 2> $OUT/c4full_bench.err
for rep in 1 2; do for V in 0 1; do
  rm -rf /tmp/prof_ab
  MDB_SCAN_F32_UNIT_ONLY=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o r -- $REPO/muopdb_amd/replay_search mspann $DUMP/c4full/spann 768 10 16 1024 10 200 > $OUT/c4_$V_$rep.log 2>&1
  echo "unit_only=$V rep=$rep: $(grep ivf_scan_f32 /tmp/prof_ab/*kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ')"
done; done
rm -rf $DUMP/c4full
bash $REPO/scripts/c5_breakdown.sh ab_c5form form1:MDB_BF_BLOCK_FORM=1 form2:MDB_BF_BLOCK_FORM=2 form3:MDB_BF_BLOCK_FORM=3 again0:MDB_BF_BLOCK_FORM=0 2>&1 | grep -E "^==|flat_bf16x1_block|ivf_scan_pq3"
