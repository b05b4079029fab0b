#!/bin/bash
# development helper: GPU parity tests with a per-test limit
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread "$@" 2>&1 | tee gpurun_out/pytest_gpu.log
