#!/bin/bash
# occupancy / register budget sweep of the L2 kernel (run on the GPU box)
set -e
for w in 2 3 4; do
  rm -f struspattern_amd/_build/obj/l2_kernel.hip.o struspattern_amd/_build/obj/capi_l2*.o
  make -s -C struspattern_amd/csrc EXTRA="-DSPA_L2_WAVES_PER_EU=$w -DSPA_L2_WAVES_PER_CU=$((4*w))" > /dev/null 2>&1
  echo "== waves/SIMD $w"
  timeout -k 10 200 python tests/micro/perf_l2.py 12288 2>&1 | grep "op="
done
