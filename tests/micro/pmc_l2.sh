#!/bin/bash
# instruction mix and wait cycles of the L2 kernels (separate counter passes; run on the GPU box from the repo root)
#   usage: tests/micro/pmc_l2.sh [ndocs]   -> gpurun_out/pmc_l2/summary.json (per kernel, per operator mix, per event)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
N=${1:-8192}
OUT=gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 --output-format csv -- python3 tests/micro/perf_l2.py $N > $OUT/$1.log 2>&1 || { tail -20 $OUT/$1.log; exit 1; }; }
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run b "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run c "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run d "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
grep "op=" $OUT/a.log
python3 - <<'PY'
import csv, glob, collections, json, re
events = {}
for l in open("gpurun_out/pmc_l2/a.log"):
    m = re.match(r"op=(\S+) docs=\d+: .* (\d+) events", l)
    if m: events[m.group(1)] = int(m.group(2))
out = {}
for run in "abcd":
    for f in glob.glob("gpurun_out/pmc_l2/%s/**/*counter_collection.csv" % run, recursive=True):
        # the script runs the 5-op mix first, then sequence only: dispatches in order
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("spa_l2")]
        disp = sorted(set(int(r["Dispatch_Id"]) for r in rows))
        half = disp[len(disp)//2] if disp else 0
        for r in rows:
            op = "None" if int(r["Dispatch_Id"]) < half else "sequence"
            k = out.setdefault(op, {}).setdefault(r["Kernel_Name"].split("(")[0], {})
            k.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summ = {}
for op, ks in out.items():
    for kn, cs in ks.items():
        d = {c: sorted(v)[len(v)//2] for c, v in cs.items()}     # median launch
        d["per_event"] = {c: d[c] / events.get(op, 1) for c in d}
        summ.setdefault(op, {})[kn] = d
json.dump(summ, open("gpurun_out/pmc_l2/summary.json", "w"), indent=1)
for op in summ:
    for kn, d in summ[op].items():
        print(op, kn, {c: round(v, 1) for c, v in d["per_event"].items()})
PY
