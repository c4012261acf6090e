#!/bin/bash
# occupancy / register budget sweep of the arena (general) L2 kernel, variants built out of tree (GPU box)
exec tests/micro/ab.sh "python3 tests/micro/perf_l2.py 12288" \
  w2:"-DSPA_L2_WAVES_PER_EU=2 -DSPA_L2_WAVES_PER_CU=8" w3:"-DSPA_L2_WAVES_PER_EU=3 -DSPA_L2_WAVES_PER_CU=12" w4:"-DSPA_L2_WAVES_PER_EU=4 -DSPA_L2_WAVES_PER_CU=16"
