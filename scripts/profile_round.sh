#!/bin/bash
# Round profile set (run on the GPU box through gpurun): for every bench workload
#   pass 1  rocprofv3 --kernel-trace --stats over bench.py itself (per-kernel average duration; C5: over the torch-free replay —
#           tracing the 100M-row shard build under rocprofv3 takes longer than the box allows)
#   pass 2/3 rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each in its own run, over the torch-free replay of the dumped files
#           (examples/replay_search.cpp): counter mode crashes inside torch's own kernels on this image.  --pmc is never combined
#           with sys / runtime trace flags.
# Output: gpurun_out/<tag>/<workload>_{kernel_stats.csv,pmc_FETCH_SIZE.csv,pmc_WRITE_SIZE.csv,replay.log} + bench_all.json;
# the ones to be judged are copied to profiles/ by hand.
#   pass 4  (flat_b64, c5: the matrix-core filters) rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over the same replay:
#           MFMA-busy against the kernel's own active cycles (north_star: "MFMA-busy against chip peak")
#   c4full  the full-size C4 (1024 users, 30.7 GB): bench.py --users 1024 --dump-big writes the files, the replay is profiled like the others
# usage: scripts/profile_round.sh <tag> [workloads...]   (default: hnsw hnsw_ef400 flat_b1 flat_b64 ivfpq spann c5 c4full c5full)
TAG=$1; shift
WL="$@"; [ -z "$WL" ] && WL="hnsw hnsw_ef400 flat_b1 flat_b64 ivfpq spann c5 c4full c5full"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
DUMP=/tmp/mdb_dump_round
PAT="ivf_coarse|hnsw_|flat_scan|flat_mfma|flat_bf16|flat_refine|sample_bound|ivf_scan|ivf_pq3|ivf_prep|ivf_pq_fused|merge_keys|merge_points"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every workload's files + the un-instrumented bench line
(time timeout 900 python $REPO/bench.py --steps 20 --warmup 5 --dump-dir $DUMP) > $OUT/bench_all.json 2> $OUT/bench_all.err
for W in $WL; do
  case $W in
    # (round 6: the headline graph is built by insertion ON THE GPU — thousands of searches through the same kernels at other shapes — so the
    # per-kernel averages come from the torch-free replay of the dumped files, like C5's, not from a traced bench run)
    hnsw)     SUB=hnsw;     REPLAY="hnsw $DUMP/hnsw 128 10 200 64 20";        BARGS="";;
    hnsw_ef400) SUB=hnsw;   REPLAY="hnsw $DUMP/hnsw 128 10 400 64 20";        BARGS="";;
    flat_b1)  SUB=flat_b1;  REPLAY="flat $DUMP/flat_b1 128 10 0 1 20";        BARGS="--workload flat --n 1000000 --batch 1";;
    flat_b64) SUB=flat_b64; REPLAY="flat $DUMP/flat_b64 128 10 0 64 20";      BARGS="--workload flat --n 1000000 --batch 64";;
    ivfpq)    SUB=ivfpq;    REPLAY="ivfpq $DUMP/ivfpq 128 10 16 256 20";      BARGS="--workload ivfpq --no-sweep --streams 0";;
    spann)    SUB=spann;    REPLAY="mspann $DUMP/spann 768 10 16 128 20 200"; BARGS="--workload spann --users 128 --no-sweep";;
    c5)       SUB=c5;       REPLAY="ivfpq $DUMP/c5 128 10 64 4096 6 0 8";     BARGS="";;   # the whole index's files, rank 0 of 8's share loaded
    c5full)   SUB=c5full;   REPLAY="ivfpq $DUMP/c5full/c5full 128 10 64 4096 6"; BARGS="";
              (time timeout 1200 python $REPO/bench.py --workload c5full --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP/c5full) > $OUT/c5full_bench.json 2> $OUT/c5full_bench.err;;
    c4full)   SUB=spann;    REPLAY="mspann $DUMP/c4full/spann 768 10 16 1024 6 200"; BARGS="";
              df -h /tmp | tail -1 > $OUT/c4full_df.log
              (time timeout 900 python $REPO/bench.py --workload spann --users 1024 --no-sweep --steps 6 --warmup 2 --no-cpu-baseline --dump-dir $DUMP/c4full --dump-big) > $OUT/c4full_bench.json 2> $OUT/c4full_bench.err;;
  esac
  if [ -n "$BARGS" ]; then
    rm -rf /tmp/prof_stats_$W
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$W -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline $BARGS > $OUT/${W}_bench_stats.log 2>&1
    cp /tmp/prof_stats_$W/*kernel_stats.csv $OUT/${W}_kernel_stats.csv 2>/dev/null
  fi
  $REPO/muopdb_amd/replay_search $REPLAY > $OUT/${W}_replay.log 2>&1
  if [ -z "$BARGS" ]; then
    rm -rf /tmp/prof_stats_$W
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$W -o replay -- $REPO/muopdb_amd/replay_search $REPLAY > $OUT/${W}_replay_stats.log 2>&1
    cp /tmp/prof_stats_$W/*kernel_stats.csv $OUT/${W}_kernel_stats.csv 2>/dev/null
  fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_${C}_$W
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_${C}_$W -o replay -- $REPO/muopdb_amd/replay_search $REPLAY > $OUT/${W}_replay_$C.log 2>&1
    echo "rc=$?" >> $OUT/${W}_replay_$C.log
    for f in /tmp/prof_${C}_$W/*counter_collection.csv; do [ -f "$f" ] && (head -1 $f; grep -E "$PAT" $f | head -400) > $OUT/${W}_pmc_$C.csv; done
  done
  case $W in flat_b64|c5|ivfpq)
    rm -rf /tmp/prof_MFMA_$W
    timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_MFMA_$W -o replay -- $REPO/muopdb_amd/replay_search $REPLAY > $OUT/${W}_replay_MFMA.log 2>&1
    echo "rc=$?" >> $OUT/${W}_replay_MFMA.log
    for f in /tmp/prof_MFMA_$W/*counter_collection.csv; do [ -f "$f" ] && (head -1 $f; grep -E "flat_bf16|flat_mfma|ivf_coarse_mfma" $f | head -600) > $OUT/${W}_pmc_MFMA.csv; done;;
  esac
  [ $W = c4full ] && rm -rf $DUMP/c4full
  [ $W = c5full ] && rm -rf $DUMP/c5full
  echo "== $W: $(tail -1 $OUT/${W}_replay.log)"
done
rm -rf $DUMP
du -sh $OUT; ls $OUT | head -60
