#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel instance in rt_kernels.hip (device-only compile to assembly).
# usage: bash tools/kernel_regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../ray-tracing_amd" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-slp-vectorize --cuda-device-only -DRT_TU_PHASED -mllvm -structurizecfg-skip-uniform-regions=1 "$@" -S csrc/rt_kernels.hip -o /tmp/rt_kernels.s 2>/dev/null || exit 1
awk '/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2} /\.vgpr_spill_count:/ {print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", g, "spill", $2}' /tmp/rt_kernels.s | c++filt | sed 's/(anonymous namespace):://'
