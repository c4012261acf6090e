#!/bin/bash
# the pipeline bench with and without shared first positions in the lexer tables (same box)
for m in off on off on; do
  export SPA_L1_SHARE=$m
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], d['kernel_ms'])"
done
