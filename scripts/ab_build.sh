#!/bin/bash
# A/B builds for same-box comparisons: scripts/ab_build.sh "<extra hipcc flags>" file.hip [file.hip ...]
# -> ab/libmuopdb_hip.so = the current tree with the named files recompiled under the extra flags (the rest: the objects of the
# last build.sh).  On the GPU box: scripts/ab_run.sh '<command>' runs the command with the tree's library (A) and ab/'s (B) in turn.
set -e
cd "$(dirname "$0")/../muopdb_amd/csrc"
XF="$1"; shift
mkdir -p ../../ab/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-variable $XF"
objs=""
for o in build/*.o; do
  s=$(basename ${o%.o}).hip; use=$o
  for f in "$@"; do
    if [ "$f" = "$s" ]; then
      extra=""
      { [ "$s" = mdb_hnsw.hip ] || [ "$s" = mdb_hnsw_upper.hip ]; } && extra="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -enable-post-misched=0"
      hipcc $FLAGS $extra -c $s -o ../../ab/build/$(basename $o) &
      use=../../ab/build/$(basename $o)
    fi
  done
  objs="$objs $use"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmuopdb_hip.so $objs
echo "built ab/libmuopdb_hip.so"
