#!/bin/bash
# rocprofv3 passes for one bench.py workload (run on the GPU box through gpurun); summaries land in
# gpurun_out/prof_<tag>/ and the ones to be judged are copied to profiles/ by hand.
# usage: scripts/profile_bench.sh <tag> <replay kind> <dim> <k> <knob> <batch> [bench args...]
#   pass 1: kernel trace + stats over bench.py itself (per-kernel average duration)
#   pass 2/3: FETCH_SIZE / WRITE_SIZE, each in its own run, over the torch-free replay of the same
#             files (examples/replay_search.cpp) — rocprofv3 counter mode crashes inside torch's own
#             kernels on this image.  No sys/runtime trace flags are combined with --pmc.
TAG=$1; KIND=$2; DIM=$3; K=$4; KNOB=$5; BATCH=$6; shift 6
EXTRA=""; [ "$KIND" = "mspann" ] && EXTRA="200"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
DUMP=/tmp/mdb_dump_$TAG
PAT="hnsw_|flat_scan|flat_mfma|flat_refine|ivf_scan"
rm -rf $OUT $DUMP; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$TAG -o bench -- python $REPO/bench.py --no-cpu-baseline --dump-dir $DUMP "$@" > $OUT/bench_stats.log 2>&1
cp /tmp/prof_stats_$TAG/*kernel_stats.csv $OUT/ 2>/dev/null
for f in /tmp/prof_stats_$TAG/*kernel_trace.csv; do [ -f "$f" ] && (head -1 $f; grep -E "$PAT" $f | head -300) > $OUT/kernel_trace_dominant.csv; done
# uninstrumented replay first (checksum + host-buffer ms/step), then the counter passes
$REPO/muopdb_amd/replay_search $KIND $DUMP $DIM $K $KNOB $BATCH 10 $EXTRA > $OUT/replay.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_${C}_$TAG -o replay -- $REPO/muopdb_amd/replay_search $KIND $DUMP $DIM $K $KNOB $BATCH 10 $EXTRA > $OUT/replay_$C.log 2>&1
  echo "rc=$?" >> $OUT/replay_$C.log
  for f in /tmp/prof_${C}_$TAG/*counter_collection.csv; do [ -f "$f" ] && (head -1 $f; grep -E "$PAT" $f | head -200) > $OUT/pmc_$C.csv; done
done
rm -rf $DUMP
du -sh $OUT; ls $OUT
