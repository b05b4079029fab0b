#!/bin/bash
# The round's measurement set, run on the GPU box from the repository root:  bash profiles/collect_all.sh r01
#  1. bench.py (headline line)               -> gpurun_out/meas/bench_line.json
#  2. the same command under rocprofv3 stats -> gpurun_out/meas/kernel_stats.csv
#  3. HBM-side traffic counter passes        -> gpurun_out/pmc/step_kernel_traffic.json
# Copy the results into profiles/<round>_* afterwards (profiles/install_results.sh does that).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/meas
timeout -k 10 400 python3 -u bench.py > gpurun_out/meas/bench_line.json
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/meas/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-large > gpurun_out/meas/bench_under_rocprof.json 2> gpurun_out/meas/rocprof.log
cp gpurun_out/meas/prof/bench_kernel_stats.csv gpurun_out/meas/kernel_stats.csv
rm -rf gpurun_out/meas/prof
echo "kernel stats done"
bash profiles/collect_traffic.sh
bash profiles/collect_l2.sh
# 4. the other configurations (bench.py --workload ...)
timeout -k 10 300 python3 -u bench.py --workload footprint_checks > gpurun_out/meas/extra_footprint.json
timeout -k 10 300 python3 -u bench.py --workload rs_table --table-pr 1 > gpurun_out/meas/extra_table_pr1.json
timeout -k 10 300 python3 -u bench.py --workload rs_table --table-pr 10 > gpurun_out/meas/extra_table_pr10.json
echo "extras done"
timeout -k 10 600 python3 -u bench.py --workload queries_large --steps 1 --warmup 0 > gpurun_out/meas/extra_queries_large.json
# the same once more with the trace's timers on (they cost a few per cent): what the chunks spend where
HYAS_TRACE=1 timeout -k 10 600 python3 -u bench.py --workload queries_large --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/meas/extra_queries_large_traced.json 2> gpurun_out/meas/extra_queries_large.log
echo "large done"
timeout -k 10 300 python3 -u bench.py --workload single_query > gpurun_out/meas/extra_single_query.json
timeout -k 10 400 python3 -u bench.py --workload modes > gpurun_out/meas/extra_modes.json
timeout -k 10 300 python3 -u bench.py --workload footprint_checks_car > gpurun_out/meas/extra_footprint_car.json
echo "single, modes, car done"
