#!/bin/bash
# instruction mix of the L2 kernel (separate counter passes; run on the GPU box from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/a --output-format csv -- python3 tests/micro/perf_l2.py 8192 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- python3 tests/micro/perf_l2.py 8192 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU -d $OUT/c --output-format csv -- python3 tests/micro/perf_l2.py 8192 > $OUT/c.log 2>&1
grep "op=" $OUT/a.log
python3 - <<'PY'
import csv, glob, collections
for run in "abc":
    for f in glob.glob("gpurun_out/pmc_l2/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "l2_match" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(run, k, "sum %.4g over %d dispatches" % (acc[k], n[k]))
PY
