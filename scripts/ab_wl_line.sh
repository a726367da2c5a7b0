#!/bin/bash
# one line for scripts/ab_run.sh: bash scripts/ab_wl_line.sh "<bench.py args>"  -> first / median region, kernel time, recall of that workload
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py $1 --no-cpu-baseline > /dev/null 2>&1
python - <<'P'
import json
j = json.load(open("gpurun_out/bench_full.json"))
d = j["dispersion"]["region_ms_per_step"]
print("first %.5f median %.5f min %.5f kernel_ms %.5f recall %s" % (j["ms_per_step"], d["median"], d["min"], j["roofline"]["kernel_ms"], j["recall_at_10"]))
P
