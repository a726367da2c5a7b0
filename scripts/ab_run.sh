#!/bin/bash
# scripts/ab_run.sh <rounds> '<command printing one line>' : A = the tree's library, B = ab/libmuopdb_hip.so, alternating on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
R=$1; shift
cp muopdb_amd/libmuopdb_hip.so /tmp/lib_a.so; cp ab/libmuopdb_hip.so /tmp/lib_b.so
for i in $(seq 1 $R); do
  for v in a b; do cp /tmp/lib_$v.so muopdb_amd/libmuopdb_hip.so; echo -n "$v: "; bash -c "$*" 2>/dev/null | tail -1; done
done
cp /tmp/lib_a.so muopdb_amd/libmuopdb_hip.so
