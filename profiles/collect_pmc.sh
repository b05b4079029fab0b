#!/bin/bash
# rocprofv3 counter passes over one short workload (profiles/workload.py <name>), each pass a run of its own with at
# most 8 SQ, 4 TCC (FETCH_SIZE takes 3, WRITE_SIZE 2), 4 TCP and 2 GRBM counters, plus a kernel trace.
# Run on the GPU box from the repository root:  bash profiles/collect_pmc.sh <workload> <outdir>
# Stops at the first pass that fails.  Results: <outdir>/<pass>/<pass>_counter_collection.csv, <outdir>/trace_kernel_stats.csv
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
wl=$1
out=$2
mkdir -p "$out"
pass() {
    name=$1
    shift
    timeout -s KILL 180 rocprofv3 --pmc "$@" -d "$out/$name" -o "$name" --output-format csv -- python3 profiles/workload.py "$wl" > "$out/$name.log" 2>&1
}
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python3 profiles/workload.py "$wl" > "$out/trace.log" 2>&1
cp "$out/trace/trace_kernel_stats.csv" "$out/trace_kernel_stats.csv"
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS
pass fetch FETCH_SIZE TCC_HIT_sum
pass write WRITE_SIZE TCC_REQ_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
rm -rf "$out/trace"
echo "passes done: $wl"
