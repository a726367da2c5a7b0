#!/usr/bin/env python3
"""Static instruction count of one kernel instance per SOURCE REGION, from a `-gline-tables-only -S` listing: every instruction
is attributed to the last `.loc` line that lies inside the kernel body (instructions of inlined helpers inherit the region of
their call site — approximate where the scheduler interleaves).  Used for DESIGN 6d: the step of hnsw_beam_kernel is one wave's
instruction chain, so a region's static size on the common path IS its share of the step.
usage: isa_region_counts.py <listing.s> <mangled kernel name> <source file> "name=first-last" ..."""
import re, sys

lst, kern, src = sys.argv[1], sys.argv[2], sys.argv[3]
regions = []
for spec in sys.argv[4:]:
    name, rng = spec.split("=")
    lo, hi = rng.split("-")
    regions.append((name, int(lo), int(hi)))
body_lo, body_hi = min(r[1] for r in regions), max(r[2] for r in regions)
counts = {r[0]: {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "branch": 0, "other": 0} for r in regions}
cur, inside = None, False
for line in open(lst):
    if line.startswith(kern + ":"):
        inside = True
        continue
    if not inside:
        continue
    s = line.strip()
    if s.startswith("s_endpgm"):
        break
    m = re.match(r"\.loc\s+0\s+(\d+)\s", s)
    if m:
        ln = int(m.group(1))
        if body_lo <= ln <= body_hi:
            cur = next((r[0] for r in regions if r[1] <= ln <= r[2]), cur)
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":") or cur is None:
        continue
    op = s.split()[0]
    kind = ("branch" if op.startswith(("s_cbranch", "s_branch")) else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "s_load", "s_buffer")) else
            "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
    counts[cur][kind] += 1
print("%-34s %6s %6s %6s %6s %6s %6s" % ("region", "VALU", "SALU", "LDS", "VMEM", "branch", "total"))
for name, _, _ in regions:
    c = counts[name]
    print("%-34s %6d %6d %6d %6d %6d %6d" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], c["branch"], sum(c.values())))
