#!/bin/bash
# Register / spill / LDS metadata of the kernels in the built library: tools/kernel_meta.sh [name filter]
set -e
T=$(mktemp -d); LIB=${COSMOFIT_LIB:-$(dirname $0)/../cosmology-model-fit_amd/libcosmofit_hip.so}
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.elf
$B/llvm-readelf --notes $T/co.elf | awk -v f="${1:-.}" '
  /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2}
  /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2}
  /\.vgpr_spill_count:/ {if (name ~ f) printf "%-90s vgpr %3d sgpr %3d sgpr_spill %3d vgpr_spill %3d scratch %4d lds %6d\n", name, vg, sg, ss, $2, scr, lds}'
rm -rf $T
