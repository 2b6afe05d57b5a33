#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of one HIP source (device-only compile with the product's flags).
# usage: scripts/kernel_regs.sh k_dc.hip [name filter]      (KEEP_ASM=path keeps the assembly)
SRC=$1; FILTER=${2:-.}
CSRC=$(dirname "$0")/../mvs-texturing_amd/csrc
EXTRA="${DEFS:-}"; [ "$SRC" = "k_bvh.hip" ] && EXTRA="$EXTRA -fno-slp-vectorize"
OUT=${KEEP_ASM:-${TMPDIR:-/tmp}/regs_$$.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $EXTRA --cuda-device-only -S "$CSRC/$SRC" -o "$OUT" 2>/dev/null || exit 1
awk '
  /^ +\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2} /\.vgpr_spill_count:/ {sp=$2}
  /\.wavefront_size:/ {printf "%s vgpr %3s sgpr %3s lds %6s scratch %4s spill %s\n", name, v, s, l, p, sp}' "$OUT" | c++filt | sed -E 's/\(anonymous namespace\):://; s/^void //; s/\((float|unsigned|mvs::|HIP_vector|int|char|bool|double|long|short|uint)[^)]*(\)|$)//; s/\(.* vgpr/ vgpr/' | grep -E "$FILTER"
[ -z "$KEEP_ASM" ] && rm -f "$OUT"
