#!/bin/bash
# instruction cache behaviour of the flat-rule automaton kernel on the pipeline workload (one PMC pass)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_icf
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $OUT/a --output-format csv -- python3 tests/micro/prof_pipe.py 3072 n > $OUT/a.log 2>&1 || { tail -20 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS -d $OUT/b --output-format csv -- python3 tests/micro/prof_pipe.py 3072 n > $OUT/b.log 2>&1 || { tail -20 $OUT/b.log; exit 1; }
tail -2 $OUT/a.log
python3 - <<'PY'
import csv, glob, collections
for run in "ab":
    for f in glob.glob("gpurun_out/pmc_icf/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "l2_fast" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = sorted(v); v = [x for x in v if x > 0.5 * v[-1]] or v
            print(run, k, "per steady launch %.4g (%d launches)" % (sum(v) / len(v), len(v)))
PY
