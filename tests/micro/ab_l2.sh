#!/bin/bash
# A/B timing of two versions of the automaton kernel on the same GPU box (box-to-box variation is
# larger than most single optimizations).  Usage: ab_l2.sh <kernelA.hip> <kernelB.hip> [capiA.cpp capiB.cpp]
set -e
A=$1; B=$2
cp struspattern_amd/csrc/l2_kernel.hip /tmp/l2_kernel_orig.hip
run() {
  cp $1 struspattern_amd/csrc/l2_kernel.hip
  rm -f struspattern_amd/_build/obj/l2_kernel.hip.o
  make -s -C struspattern_amd/csrc > /dev/null 2>&1
  timeout -k 10 200 python tests/micro/perf_l2.py 12288 2>&1 | grep "op=" | sed "s/^/$2: /"
}
for round in 1 2; do
  run $A A
  run $B B
done
cp /tmp/l2_kernel_orig.hip struspattern_amd/csrc/l2_kernel.hip
