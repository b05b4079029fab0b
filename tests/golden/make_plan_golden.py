#!/usr/bin/env python3
"""Generates tests/golden/plan_golden.npz: end-to-end search outputs of the CPU
oracle for fixed queries (status, counters, ordered expanded-node indices, path
poses, costs).  The reference itself cannot be built in this image (ROS, Eigen,
Boost.Heap, OMPL and glog are absent), so these vectors pin the restatement
against regressions and the GPU path against the restatement; they are not
outputs of the reference.

    make -C oracle && python tests/golden/make_plan_golden.py
"""
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
import orc  # noqa: E402

MAZE_PICKS = [0, 1, 4, 5, 14, 15, 17, 22, 25]  # SHOT after 30..1800 expansions, NO_PATH, MAX_ITER, direct shot


def maze_queries(o, n=40):
    """The first n benchmark-style queries on the maze (same sampling as the benchmark: default_rng(20250228))."""
    rng = np.random.default_rng(20250228)

    def draw(k):
        out = []
        while len(out) < k:
            p = np.stack([rng.uniform(1, 49, 4 * k), rng.uniform(1, 49, 4 * k), rng.uniform(0, 2 * math.pi, 4 * k)], 1).astype(np.float32)
            ok = o.collision(p)
            out.extend(p[ok == 1])
        return np.array(out[:k], np.float32)

    return draw(n), draw(n)


def main():
    out = {}
    names = []

    def record(name, o, mp, pr, s, g):
        res, path, expanded = o.plan(np.asarray(s, np.float32), np.asarray(g, np.float32))
        out[name + "/result"] = res
        out[name + "/path"] = path.astype(np.float32)
        out[name + "/expanded"] = expanded.astype(np.int32)
        out[name + "/start"] = np.asarray(s, np.float32)
        out[name + "/goal"] = np.asarray(g, np.float32)
        out[name + "/map"] = np.array(mp)
        out[name + "/pr"] = np.array([pr], np.int32)
        names.append(name)
        print(name, res)

    # the reference's fixed poses on the parking map, src/planner.cpp:193-207
    o = orc.Oracle(orc.hot_path_params(position_resolution=2))
    o.set_costmap(orc.load_map("map_parking"), build_table=True, nthreads=8)
    record("parking_fixed", o, "map_parking", 2, (2, 28, 0), (58, 2, 0))
    # the same query with the parameters the single-query configuration is quoted on (position_resolution 10)
    o = orc.Oracle(orc.hot_path_params())
    o.set_costmap(orc.load_map("map_parking"), build_table=True, nthreads=8)
    record("parking_fixed_pr10", o, "map_parking", 10, (2, 28, 0), (58, 2, 0))
    o = orc.Oracle(orc.hot_path_params(position_resolution=2))
    o.set_costmap(orc.load_map("map_maze"), build_table=True, nthreads=8)
    S, G = maze_queries(o)
    for i in MAZE_PICKS:
        record(f"maze_{i:02d}", o, "map_maze", 2, S[i], G[i])
    out["names"] = np.array(names)
    np.savez_compressed(ROOT / "tests" / "golden" / "plan_golden.npz", **out)


if __name__ == "__main__":
    main()
