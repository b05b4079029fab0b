"""Per-launch HBM-side bytes of the round kernel from two rocprofv3 counter CSVs (FETCH_SIZE, WRITE_SIZE; both in KB).

On gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes for wide streaming reads; this kernel's reads are
narrow gathers for which the counter is not calibrated, so the raw figure and the doubled one are both reported
and `hbm_bytes_per_launch` uses the doubled read figure (an upper estimate)."""
import csv
import json
import sys


def per_launch(path, counter, kernel):
    tot = 0.0
    n = 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
    return (tot / n * 1024.0 if n else 0.0), n


fetch, n1 = per_launch(sys.argv[1], "FETCH_SIZE", "k_round")
write, n2 = per_launch(sys.argv[2], "WRITE_SIZE", "k_round")
print(json.dumps({"kernel": "k_round", "launches_fetch_pass": n1, "launches_write_pass": n2, "fetch_bytes_per_launch_raw": fetch,
                  "fetch_bytes_per_launch_x2": 2 * fetch, "write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": 2 * fetch + write,
                  "workload": "1024 maze queries, max_iterations 300 (profiles/traffic_workload.py)"}))
