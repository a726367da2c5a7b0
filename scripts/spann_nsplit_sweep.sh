cd ${GRAFT_REPO_ROOT:-/root/repo}
for ns in 0 3 4 5 6 8 10 12 16 24; do
  for blk in 256 128; do
  MDB_SCAN_F32_NSPLIT=$ns MDB_SCAN_F32_BLK=$blk python bench.py --workload spann --users 128 --no-sweep --no-cpu-baseline > /dev/null 2>&1
  python - $ns $blk <<'P'
import json, sys
j = json.load(open("gpurun_out/bench_full.json"))
d = j["dispersion"]["region_ms_per_step"]
print("nsplit", sys.argv[1], "blk", sys.argv[2], "median %.5f kernel_ms %.5f" % (d["median"], j["roofline"]["kernel_ms"]))
P
  done
done
