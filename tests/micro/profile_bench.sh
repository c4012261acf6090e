#!/bin/bash
# rocprofv3 evidence for the headline bench shape (run on the GPU box from the repo root):
#   pass 1: --kernel-trace --stats (per-kernel durations)
#   pass 2/3: FETCH_SIZE / WRITE_SIZE in their own runs (HBM-side traffic)
#   pass 4-6: instruction mix, wave / wait cycles, LDS conflicts (SQ counters, own runs)
# -> gpurun_out/prof_bench/summary.json (copied to profiles/rNN_bench_pmc_summary.json) + kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_bench
CMD="bench.py --no-cpu-baseline --docbytes 65536 --steps 2 --warmup 1"
rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace $2 -d $OUT/$1 --output-format csv -- python3 $CMD > $OUT/$1.log 2>&1 || { tail -20 $OUT/$1.log; exit 1; }; tail -1 $OUT/$1.log | cut -c1-200; }
run stats "--stats"
run fetch "--pmc FETCH_SIZE"
run write "--pmc WRITE_SIZE"
run insts "--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
run waits "--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
run lds "--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - <<'PY'
import csv, glob, json, collections
out = {"command": "python3 bench.py --no-cpu-baseline --docbytes 65536 --steps 2 --warmup 1",
       "note": "FETCH_SIZE/WRITE_SIZE in KB as reported by rocprofv3; hbm_read_bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction of MI355X_MICROARCH.md, HBM section). "
               "Steady state = launches within 2x of the longest one of a kernel (bench.py starts with short sizing launches). SQ_* per launch; SQ_WAVE_CYCLES / WAIT / ACTIVE are quad-cycles. "
               "VGPR_Count / SGPR_Count are what the kernel trace reports per dispatch (granule-rounded allocation of the ARCH registers; the code-object metadata is in rNN_kernel_resources.txt).",
       "kernels": {}}
def group(name):
    return name.split("(")[0]
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_bench/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("spa_"):
            dur[group(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            k = out["kernels"].setdefault(group(r["Kernel_Name"]), {})
            for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                if c in r: k[c] = int(float(r[c]))
for k, v in dur.items():
    steady = [x for x in v if x > 0.5 * max(v)]
    out["kernels"][k].update({"launches": len(v), "steady_state_launches": len(steady), "kernel_ms_steady": sum(steady) / len(steady)})
for run_, keys in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                   ("insts", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"]),
                   ("waits", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES"]),
                   ("lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"])):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/prof_bench/%s/**/*counter_collection.csv" % run_, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("spa_") and r["Counter_Name"] in keys:
                acc[group(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            v = sorted(v)
            v = [x for x in v if x > 0.5 * v[-1]] or v          # steady-state launches
            out["kernels"].setdefault(k, {})[c + "_per_launch"] = sum(v) / len(v)
for k, d in out["kernels"].items():
    if "FETCH_SIZE_per_launch" in d: d["hbm_read_bytes_per_launch"] = 2 * 1024 * d["FETCH_SIZE_per_launch"]
    if "WRITE_SIZE_per_launch" in d: d["hbm_write_bytes_per_launch"] = 1024 * d["WRITE_SIZE_per_launch"]
json.dump(out, open("gpurun_out/prof_bench/summary.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {c: (round(v, 3) if isinstance(v, float) and v < 1e6 else v) for c, v in d.items() if "per_launch" not in c or c.startswith("hbm")})
PY
