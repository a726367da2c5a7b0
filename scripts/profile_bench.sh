#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun); summaries land in gpurun_out/prof_<tag>/
# usage: scripts/profile_bench.sh <tag> [bench args...]
TAG=${1:-hnsw}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# pass 1: per-kernel time (kernel trace + stats)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
cp /tmp/prof_stats/*kernel_stats.csv $OUT/ 2>/dev/null; ls -la /tmp/prof_stats | head
# per-dispatch rows of the dominant kernels only (the full trace is large)
for f in /tmp/prof_stats/*kernel_trace.csv; do [ -f "$f" ] && (head -1 $f; grep -E "hnsw_search|flat_scan|ivf_scan" $f | head -400) > $OUT/kernel_trace_dominant.csv; done
# pass 2/3: HBM traffic counters, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o bench -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $OUT/bench_$C.log 2>&1
  echo "rc=$?" >> $OUT/bench_$C.log
  for f in /tmp/prof_$C/*counter_collection.csv; do [ -f "$f" ] && (head -1 $f; grep -E "hnsw_search|flat_scan|ivf_scan" $f | head -200) > $OUT/pmc_$C.csv; done
done
du -sh $OUT; ls $OUT
