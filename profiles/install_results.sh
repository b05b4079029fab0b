#!/bin/bash
# Copies what profiles/collect_all.sh left under gpurun_out/ into profiles/<round>_*; the traffic figure goes into the bench line.
set -e
cd "$(dirname "$0")/.."
r=${1:-r01}
cp gpurun_out/meas/kernel_stats.csv profiles/${r}_bench_kernel_stats.csv
cp gpurun_out/pmc/step_kernel_traffic.json profiles/${r}_step_kernel_traffic.json
cp gpurun_out/pmc/step_kernel_l2.json profiles/${r}_step_kernel_l2.json
python3 - "$r" <<'PY'
import json, sys
r = sys.argv[1]
line = json.loads(open("gpurun_out/meas/bench_line.json").read().strip().splitlines()[-1])
t = json.load(open(f"profiles/{r}_step_kernel_traffic.json"))
if line.get("roofline"):
    line["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
    line["roofline"]["traffic_note"] = f"from profiles/{r}_step_kernel_traffic.json (same build, separate counter passes, full launches of the first 300 iterations)"
open(f"profiles/{r}_bench_line.json", "w").write(json.dumps(line) + "\n")
print(json.dumps(line)[:600])
PY
for x in footprint footprint_car table_pr1 table_pr10 queries_large single_query modes; do
    if [ -s gpurun_out/meas/extra_$x.json ]; then tail -n 1 gpurun_out/meas/extra_$x.json > profiles/${r}_extra_$x.json; fi
done
if [ -s gpurun_out/meas/extra_queries_large.log ]; then grep "^hyas chunk\|^hyas rounds (" gpurun_out/meas/extra_queries_large.log > profiles/${r}_extra_queries_large_trace.txt; fi
