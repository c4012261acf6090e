#!/bin/bash
# Does the size of the per-wave arena (same work, larger address range) change the automaton's time?
set -e
for g in 0 4 5; do
  SPA_BENCH_EXTRA_GROW=$g SPA_L2_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 2>&1 | grep -E '"metric"|arena:' | tail -2 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('extra_grow=$g', d['value'], d['config'].get('kernel_ms') or d.get('kernel_ms'))
    else: print(l.strip())"
done
