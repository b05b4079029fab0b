#!/bin/bash
# The bench lines of every workload plus a kernel trace of the headline command, run on the GPU box from the
# repository root:  bash profiles/collect_bench.sh <outdir>      (counter passes: profiles/collect_pmc.sh)
# Stops at the first step that fails; stderr of every step is kept beside its output.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=${1:-gpurun_out/meas}
mkdir -p "$out"
timeout -k 10 400 python3 -u bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-large > "$out/bench_under_rocprof.json" 2> "$out/rocprof.err"
cp "$out/prof/bench_kernel_stats.csv" "$out/bench_kernel_stats.csv"
rm -rf "$out/prof"
echo "kernel stats done"
timeout -k 10 300 python3 -u bench.py --workload footprint_checks > "$out/extra_footprint.json" 2> "$out/extra_footprint.err"
timeout -k 10 300 python3 -u bench.py --workload footprint_checks_car > "$out/extra_footprint_car.json" 2> "$out/extra_footprint_car.err"
timeout -k 10 300 python3 -u bench.py --workload rs_table --table-pr 1 > "$out/extra_table_pr1.json" 2> "$out/extra_table_pr1.err"
timeout -k 10 300 python3 -u bench.py --workload rs_table --table-pr 10 > "$out/extra_table_pr10.json" 2> "$out/extra_table_pr10.err"
timeout -k 10 300 python3 -u bench.py --workload single_query --steps 1 > "$out/extra_single_query.json" 2> "$out/extra_single_query.err"
timeout -k 10 600 python3 -u bench.py --workload modes --steps 2 > "$out/extra_modes.json" 2> "$out/extra_modes.err"
echo "extras done"
HYAS_TRACE=1 timeout -k 10 600 python3 -u bench.py --workload queries_large --steps 1 --warmup 0 > "$out/extra_queries_large.json" 2> "$out/extra_queries_large.err"
echo "large done"
