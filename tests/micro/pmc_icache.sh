#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_ic
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $OUT/a --output-format csv -- python3 tests/micro/perf_l2.py 6144 > $OUT/a.log 2>&1 || { tail -20 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH_LEVEL -d $OUT/b --output-format csv -- python3 tests/micro/perf_l2.py 6144 > $OUT/b.log 2>&1 || { tail -20 $OUT/b.log; exit 1; }
tail -3 $OUT/a.log
python3 - <<'PY'
import csv, glob, collections
for run in "ab":
    for f in glob.glob("gpurun_out/pmc_ic/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "l2_match" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(run, k, "sum %.4g (%d launches)" % (acc[k], n[k]))
PY
