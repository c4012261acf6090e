#!/bin/bash
# everything profiles/rNN_* is made of, in one call on the GPU box (repo root): bench lines of the four workloads, rocprofv3 stats + PMC
# passes of the headline bench (profile_bench.sh), plugin path rates
set -e
R=${1:-r03}
mkdir -p gpurun_out/evidence
python3 bench.py --docbytes 65536 > gpurun_out/evidence/${R}_bench_line.json 2> gpurun_out/evidence/bench.err || { tail -5 gpurun_out/evidence/bench.err; exit 1; }
python3 bench.py --workload lexer --no-cpu-baseline > gpurun_out/evidence/${R}_bench_line_lexer.json 2>> gpurun_out/evidence/bench.err
python3 bench.py --workload l2 > gpurun_out/evidence/${R}_bench_line_l2.json 2>> gpurun_out/evidence/bench.err
timeout -k 10 300 python3 bench.py --workload trees --no-cpu-baseline > gpurun_out/evidence/${R}_bench_line_trees.json 2>> gpurun_out/evidence/bench.err || echo "trees bench failed"
timeout -k 10 300 python3 tests/micro/plugin_threads.py 64 16384 1,4,8,16 > gpurun_out/evidence/${R}_plugin_threads.txt 2>&1 || true; cat gpurun_out/evidence/${R}_plugin_threads.txt
bash tests/micro/profile_bench.sh > gpurun_out/evidence/profile_bench.log 2>&1 || { tail -20 gpurun_out/evidence/profile_bench.log; exit 1; }
cp gpurun_out/prof_bench/summary.json gpurun_out/evidence/${R}_bench_pmc_summary.json
cp gpurun_out/prof_bench/kernel_stats.csv gpurun_out/evidence/${R}_bench_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trees --output-format csv -- python3 bench.py --workload trees --no-cpu-baseline --steps 2 > gpurun_out/evidence/trees_prof.log 2>&1 || true
cp $(ls gpurun_out/prof_trees/*/*kernel_stats.csv | head -1) gpurun_out/evidence/${R}_trees_kernel_stats.csv || true
tail -c 600 gpurun_out/evidence/${R}_bench_line.json; echo; cat gpurun_out/evidence/${R}_plugin_threads.txt; tail -8 gpurun_out/evidence/profile_bench.log
