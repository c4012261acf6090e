#!/bin/bash
# instruction mix and wait cycles of the automaton kernel on the pipeline workload (separate counter passes; GPU box, repo root)
#   usage: [SPA_LIB=...] tests/micro/pmc_pipe.sh <tag> [ndocs]   -> gpurun_out/pmc_pipe_<tag>/summary.txt (per event)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=$1; N=${2:-6144}
OUT=gpurun_out/pmc_pipe_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 --output-format csv -- python3 tests/micro/prof_pipe.py $N n > $OUT/$1.log 2>&1 || { tail -20 $OUT/$1.log; exit 1; }; }
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
run c "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run d "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
grep "pipeline L2" $OUT/a.log
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
ev = None
for l in open(out + "/a.log"):
    m = re.search(r"(\d+) events", l)
    if m: ev = int(m.group(1))
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("spa_l2_fast"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for c, v in sorted(acc.items()):
        v = sorted(v); v = [x for x in v if x > 0.5 * v[-1]] or v
        line = "%s per launch %.4g per event %.2f" % (c, sum(v)/len(v), sum(v)/len(v)/ev)
        print(line); fo.write(line + "\n")
PY
