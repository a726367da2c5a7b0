#!/bin/bash
# registers / scratch / LDS of the kernels whose mangled name matches $1 inside the built library (no GPU needed):
#   scripts/kernel_regs.sh ivf_coarse_mfma
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
cp "$(dirname "$0")/../muopdb_amd/libmuopdb_hip.so" $T/
(cd $T && $LLVM/llvm-objdump --offloading libmuopdb_hip.so > /dev/null 2>&1)
for f in $T/*gfx950; do
  $LLVM/llvm-readelf --notes $f 2>/dev/null | awk -v pat="$1" '
    /\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2}
    /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {s=$2; if (n ~ pat) printf "%-90s vgpr %3d agpr %3d spill %3d scratch %4d lds %6d\n", n, v, a, s, p, l}'
done
rm -rf $T
