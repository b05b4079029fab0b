#!/bin/bash
# L2-side counters of the round kernel (one rocprofv3 counter pass over the short benchmark workload):
# requests, hits and misses per launch.  Run on the GPU box from the repository root:  bash profiles/collect_l2.sh
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc/l2 -o l2 --output-format csv -- python3 profiles/traffic_workload.py > gpurun_out/pmc/l2.log 2>&1
python3 - <<'PY' > gpurun_out/pmc/step_kernel_l2.json
import csv, collections, json
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for r in csv.DictReader(open("gpurun_out/pmc/l2/l2_counter_collection.csv")):
    if "k_round" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for r in csv.DictReader(open("gpurun_out/pmc/l2/l2_kernel_trace.csv")):
    if "k_round" in r["Kernel_Name"]:
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
per = {k: tot[k] / n[k] for k in tot}
us = sum(dur) / len(dur) / 1e3
out = {"kernel": "k_round", "launches": n["TCC_REQ_sum"], "l2_requests_per_launch": per["TCC_REQ_sum"],
       "l2_hits_per_launch": per["TCC_HIT_sum"], "l2_misses_per_launch": per["TCC_MISS_sum"],
       "l2_hit_rate": per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1.0),
       "avg_launch_us_in_this_pass": us,
       "l2_GBps_at_64B_per_request": per["TCC_REQ_sum"] * 64 / us / 1e3, "l2_GBps_at_128B_per_request": per["TCC_REQ_sum"] * 128 / us / 1e3,
       "l2_peak_GBps": 34500.0, "workload": "1024 maze queries, max_iterations 300 (profiles/traffic_workload.py); launches serialised by the counter pass"}
print(json.dumps(out))
PY
cat gpurun_out/pmc/step_kernel_l2.json
