#!/bin/bash
# same-box A/B of environment settings on one bench workload: scripts/ab_env.sh "<bench args>" "VAR=a" "VAR=b" [repeats]
# prints ms_per_step (first region) and the median region of every run, alternating the settings
ARGS="$1"; A="$2"; B="$3"; R=${4:-3}
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    env $v python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$v" <<'P'
import json, sys
j = json.load(open("gpurun_out/bench_full.json"))
d = j.get("dispersion", {}).get("region_ms_per_step", {})
print(sys.argv[1], "first %.5f median %.5f min %.5f kernel_ms %s recall %s" % (j["ms_per_step"], d.get("median", 0), d.get("min", 0), j["roofline"].get("kernel_ms"), j.get("recall_at_10")))
P
  done
done
