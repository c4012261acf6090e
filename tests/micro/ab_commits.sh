#!/bin/bash
# Same-box timing of the library as it was at earlier commits (box-to-box variation is larger than most single changes).
#   step 1, in the build container:  tests/micro/ab_commits.sh build <commit> [<commit> ...]
#       (sources of each commit -> /tmp, library -> tests/micro/_build/<commit>/, which travels to the GPU box but is git-ignored)
#   step 2, on the GPU box:          tests/micro/ab_commits.sh run "<command>" <commit>|head ...
#       (the command runs with SPA_LIB pointing at each build in turn, two rounds; the Python package must match the C-ABI
#        of those commits)
set -euo pipefail
mode=$1; shift
if [ "$mode" = build ]; then
  for c in "$@"; do
    w=$(mktemp -d)
    git archive "$c" struspattern_amd/csrc include | tar -x -C "$w"
    make -s -C "$w/struspattern_amd/csrc" OUTDIR="$(pwd)/tests/micro/_build/$c" > "$w/build.log" 2>&1 || { tail -20 "$w/build.log"; exit 1; }
    rm -rf "$w"
    ls -l "tests/micro/_build/$c/libstruspattern_amd.so"
  done
else
  cmd=$1; shift
  for round in 1 2; do
    for v in "$@"; do
      if [ "$v" = head ]; then unset SPA_LIB; else export SPA_LIB="tests/micro/_build/$v/libstruspattern_amd.so"; fi
      $cmd 2>&1 | tail -3 | sed "s/^/$v: /"
    done
  done
fi
