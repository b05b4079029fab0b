#!/bin/bash
# rocprofv3 counter passes over one form of profiles/table_micro (bit mask <forms>: 1 walk, 2 regroup, 4 regroup1), each
# pass a run of its own, plus a kernel trace.  Run on the GPU box from the repository root:
#   bash profiles/collect_table_micro.sh <forms> <outdir>
# Results: <outdir>/<pass>/<pass>_counter_collection.csv, <outdir>/trace_kernel_stats.csv
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
forms=$1
out=$2
mkdir -p "$out"
pass() {
    name=$1
    shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out/$name" -o "$name" --output-format csv -- ./profiles/table_micro 1 10 "$forms" > "$out/$name.log" 2>&1
}
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- ./profiles/table_micro 3 10 "$forms" > "$out/trace.log" 2>&1
cp "$out/trace/trace_kernel_stats.csv" "$out/trace_kernel_stats.csv"
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS
pass grbm GRBM_GUI_ACTIVE WRITE_SIZE
rm -rf "$out/trace"
echo "passes done"
