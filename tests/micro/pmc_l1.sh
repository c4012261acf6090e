#!/bin/bash
# instruction mix of the two L1 kernels (scan, post) on the 10k-regex set (separate counter passes; run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_l1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/a --output-format csv -- python3 tests/micro/quick_l1.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- python3 tests/micro/quick_l1.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU -d $OUT/c --output-format csv -- python3 tests/micro/quick_l1.py > $OUT/c.log 2>&1
grep "npat" $OUT/a.log | tail -1
python3 - <<'PY'
import csv, glob, collections
for run in "abc":
    for f in glob.glob("gpurun_out/pmc_l1/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "spa_l1_" in kn:
                key = ("scan" if "scan" in kn else "post", r["Counter_Name"])
                acc[key] += float(r["Counter_Value"]); n[key] += 1
        for k in sorted(acc): print(run, k[0], k[1], "per launch %.4g (%d launches)" % (acc[k] / n[k], n[k]))
PY
