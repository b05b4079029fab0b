"""Phase clocks of the search-round kernel on the benchmark batch (1024 maze queries, hot-path parameters):
hyas_set_stats(ctx, 2) makes hyas_plan_batch run the kernel variant that sums shader clocks per phase over the shot
workgroups.  Prints one JSON line with the averages per shot workgroup, in shader-clock cycles.

    python profiles/phase_clocks.py [--out FILE]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from path_planner_amd import Hyas, hot_path_params  # noqa: E402
from path_planner_amd.sharding import load_map, make_queries  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    h = Hyas(hot_path_params(), device=0)
    h.set_costmap(load_map("map_maze"), build_table=True)
    S, G = make_queries(h.collision, 1024, 20250228, 50, 50)
    h.plan_batch(S, G)  # warm-up with the plain kernel
    h.set_stats(2)
    t0 = time.perf_counter()
    res = h.plan_batch(S, G)
    wall = time.perf_counter() - t0
    st = list(h.stats().values())  # in the order of hyas_get_stats
    ms, launches = h.kernel_ms(0)
    wg = max(st[6], 1)
    per = lambda i: st[i] / wg  # noqa: E731
    out = {
        "workload": "map_maze 50x50, 1024 queries, hot-path parameters, kernel variant with phase clocks",
        "shot_workgroups": st[6], "launches": int(launches), "kernel_us_per_launch": 1e3 * ms / max(launches, 1), "batch_s": wall,
        "nodes_expanded": int(res["nodes_expanded"].sum()),
        "cycles_per_shot_workgroup": {
            "node_and_goal_arrived": per(15), "goal_frame_normalised": per(11), "candidates_evaluated": per(12), "winner_known": per(13),
            "prefix_states_stored": per(14), "curve_solve_phase": per(8), "sample_interpolation": per(9), "sample_collision_tests": per(10),
            "pose_tests_of_wave_0": per(17), "segment_tests_of_wave_2": per(18), "whole_workgroup": per(16)},
        "sample_passes_per_shot": st[19] / wg,
        "note": "clock64() differences summed over the shot workgroups (lane 0 of the wave named), divided by their number",
    }
    line = json.dumps(out)
    print(line)
    if a.out:
        Path(a.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
