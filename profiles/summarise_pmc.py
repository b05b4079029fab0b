"""Per-launch averages of every counter collect_pmc.sh gathered for the kernels whose name contains <kernel>:
  python3 profiles/summarise_pmc.py <outdir> <kernel substring> [units per launch] > summary.json
Byte figures: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming
reads at 64 bytes, so `hbm_bytes_per_launch` doubles it (an upper estimate for narrow gathers)."""
import collections
import csv
import glob
import json
import sys

out, kernel = sys.argv[1], sys.argv[2]
units = float(sys.argv[3]) if len(sys.argv) > 3 else None
tot = collections.defaultdict(float)
cnt = collections.defaultdict(int)
for path in glob.glob(f"{out}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
per = {k: tot[k] / cnt[k] for k in tot}
res = {"kernel": kernel, "launches_per_pass": max(cnt.values()) if cnt else 0, "counters_per_launch": per}
for r in csv.DictReader(open(f"{out}/trace_kernel_stats.csv")):
    if kernel in r["Name"]:
        res["kernel_trace"] = {"name": r["Name"][:120], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                               "max_ns": float(r["MaxNs"])}
        break
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    res["fetch_bytes_per_launch_raw"] = per["FETCH_SIZE"] * 1024
    res["write_bytes_per_launch"] = per["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = 2 * per["FETCH_SIZE"] * 1024 + per["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
    res["l2_hit_rate"] = per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1.0)
if all(k in per for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")):
    flops = 64.0 * (per["SQ_INSTS_VALU_ADD_F64"] + per["SQ_INSTS_VALU_MUL_F64"] + 2 * per["SQ_INSTS_VALU_FMA_F64"] + per["SQ_INSTS_VALU_TRANS_F64"])
    res["f64_flops_per_launch"] = flops
    if units:
        res["f64_flops_per_entry"] = flops / units
if "SQ_BUSY_CYCLES" in per and "SQ_ACTIVE_INST_VALU" in per:
    res["note_cycles"] = "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves"
if units:
    res["units_per_launch"] = units
print(json.dumps(res))
