#!/bin/bash
# A/B of two versions of the automaton kernel on the pipeline bench, same GPU box.
# Usage: ab_bench.sh <kernelA.hip> <kernelB.hip>
set -e
A=$1; B=$2
cp struspattern_amd/csrc/l2_kernel.hip /tmp/l2_kernel_orig.hip
run() {
  cp $1 struspattern_amd/csrc/l2_kernel.hip
  rm -f struspattern_amd/_build/obj/l2_kernel.hip.o
  make -s -C struspattern_amd/csrc > /dev/null 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['config'].get('kernel_ms') or d.get('kernel_ms'))"
}
run $A A
run $B B
run $A A
run $B B
cp /tmp/l2_kernel_orig.hip struspattern_amd/csrc/l2_kernel.hip
