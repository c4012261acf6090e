#!/bin/bash
# Same-box A/B timing of build variants of the library (box-to-box variation is larger than most single
# optimizations).  A variant is a set of preprocessor flags; every variant is built OUT OF TREE into its own
# directory (the tracked sources and the product library are never touched) and loaded through SPA_LIB.
#   usage: tests/micro/ab.sh "<command>" name1:"<flags>" name2:"<flags>" ...
#   e.g.   tests/micro/ab.sh "python3 tests/micro/perf_l2.py 12288" base: occ4:"-DSPA_L2_WAVES_PER_EU=4"
# Variants run interleaved, two rounds each; the command's output lines are prefixed with the variant name.
set -euo pipefail
CMD=$1; shift
ROOT=$(pwd)
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
names=()
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -s -C struspattern_amd/csrc OUTDIR="$WORK/$name" EXTRA="$flags" > "$WORK/$name.build.log" 2>&1 || { tail -20 "$WORK/$name.build.log"; exit 1; }
  names+=("$name")
done
for round in 1 2; do
  for name in "${names[@]}"; do
    if ! SPA_LIB="$WORK/$name/libstruspattern_amd.so" timeout -k 10 300 $CMD > "$WORK/$name.run.log" 2>&1; then
      echo "$name: FAILED"; tail -20 "$WORK/$name.run.log"; exit 1
    fi
    sed "s/^/$name: /" "$WORK/$name.run.log" | tail -6
  done
done
