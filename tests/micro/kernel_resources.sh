#!/bin/bash
# registers, spills, LDS and scratch of the kernels of one source file (device-only compile into a scratch directory)
#   usage: tests/micro/kernel_resources.sh struspattern_amd/csrc/l1_kernel.hip [name filter] [extra flags]
set -e
SRC=$(realpath "$1"); FILTER=${2:-.}; shift; shift || true
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cd "$W"
hipcc --offload-arch=gfx950 -std=c++17 -O3 -x hip "$SRC" --cuda-device-only -S -o k.s "$@" 2> err.log || { cat err.log; exit 1; }
[ -n "$KEEP_ASM" ] && cp k.s "$KEEP_ASM"
grep -E "^\s+\.(name|vgpr_count|sgpr_spill_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" k.s | paste - - - - - - | sed 's/  */ /g; s/\t/ /g' | grep -E "$FILTER"
