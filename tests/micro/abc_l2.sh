#!/bin/bash
# timing of several versions of the automaton kernel on the same GPU box: abc_l2.sh <kernel.hip>...
set -e
cp struspattern_amd/csrc/l2_kernel.hip /tmp/l2_kernel_orig.hip
for round in 1 2; do
  for f in "$@"; do
    cp $f struspattern_amd/csrc/l2_kernel.hip
    rm -f struspattern_amd/_build/obj/l2_kernel.hip.o
    make -s -C struspattern_amd/csrc > /dev/null 2>&1
    timeout -k 10 200 python tests/micro/perf_l2.py 12288 2>&1 | grep "op=" | sed "s|^|$(basename $f): |"
  done
done
cp /tmp/l2_kernel_orig.hip struspattern_amd/csrc/l2_kernel.hip
