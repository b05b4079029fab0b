#!/bin/bash
# HBM-side traffic of the round kernel: two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share
# one), each over a short run of the benchmark workload (1024 maze queries, 300 iterations per query).
# Run on the GPU box from the repository root:  bash profiles/collect_traffic.sh
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc/fetch -o fetch --output-format csv -- python3 profiles/traffic_workload.py > gpurun_out/pmc/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc/write -o write --output-format csv -- python3 profiles/traffic_workload.py > gpurun_out/pmc/write.log 2>&1
python3 profiles/summarise_traffic.py gpurun_out/pmc/fetch/fetch_counter_collection.csv gpurun_out/pmc/write/write_counter_collection.csv > gpurun_out/pmc/step_kernel_traffic.json
cat gpurun_out/pmc/step_kernel_traffic.json
