#!/bin/bash
# A/B timing of two versions of the lexer kernel on the same GPU box.  Usage: ab_l1.sh <kernelA.hip> <kernelB.hip>
set -e
A=$1; B=$2
cp struspattern_amd/csrc/l1_kernel.hip /tmp/l1_kernel_orig.hip
run() {
  cp $1 struspattern_amd/csrc/l1_kernel.hip
  rm -f struspattern_amd/_build/obj/l1_kernel.hip.o
  make -s -C struspattern_amd/csrc > /dev/null 2>&1
  timeout -k 10 200 python tests/micro/quick_l1.py 2>&1 | grep "npat" | tail -1 | sed "s/^/$2: /"
  timeout -k 10 200 python bench.py --workload lexer --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$2: lexer-256', round(d['value'],3), 'GB/s', round(d['kernel_ms']['spa_l1_lex_kernel'],1), 'ms')"
}
for round in 1 2; do
  run $A A
  run $B B
done
cp /tmp/l1_kernel_orig.hip struspattern_amd/csrc/l1_kernel.hip
