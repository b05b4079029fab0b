"""One-off check beyond the test suite's sample: every 16th query of the 65 536-query map_large batch (seed 3; 4096
queries, full iteration limit) planned on the GPU in one call and by the CPU oracle on the box's cores; result records,
ordered expanded cells and paths must be bit-identical.  Prints one JSON line.

    python tests/large_batch_check.py [--every 16] [--out FILE]      (not collected by pytest: it takes minutes)
    python tests/large_batch_check.py --maze-modes [--out FILE]

--maze-modes: the 1024 queries of the benchmark's maze batch under the configurations beside the benchmark's (adaptive
steering angle and step size, adaptive step size, cubic Bezier curves, Dubins curves, the shipped parameter file with a
6 m obstacle range), every query against the oracle.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))  # (the script lives there: the CPU oracle is the tests' checker)

import orc  # noqa: E402
from path_planner_amd.sharding import make_queries  # noqa: E402
from test_gpu_configs import _compare_threaded, _pair  # noqa: E402


def maze_modes(a):
    from path_planner_amd import Hyas, Params
    from path_planner_amd import _lib as hl

    mp = orc.load_map("map_maze")
    H, W = mp.shape
    shipped = dict(hl.SHIPPED_YAML)
    shipped.update(obstacle_detection_range=6.0, max_iterations=10000)
    variants = {
        "adaptive steering angle and step size": dict(adaptive_steering_angle_and_step_size=1),
        "adaptive step size": dict(adaptive_step_size=1),
        "cubic Bezier curves": dict(curve_type=2),
        "Dubins curves, shots at the paper's schedule": dict(curve_type=0, analytical_expansion_every_point=0),
        "shipped parameter file with a 6 m obstacle range": shipped,
    }
    rows, S, G = {}, None, None
    for name, kw in variants.items():
        t1 = time.perf_counter()
        po = (orc.default_params if kw is shipped else orc.hot_path_params)(**kw)
        ph = Params()
        for field, _ in Params._fields_:
            setattr(ph, field, getattr(po, field))
        o = orc.Oracle(po)
        o.set_costmap(mp, build_table=True, nthreads=16)
        h = Hyas(ph)
        h.set_costmap(mp, build_table=True)
        if S is None:
            S, G = make_queries(h.collision, 1024, 20250228, W, H)
        res = h.plan_batch(S, G)
        bad = _compare_threaded(o, h, S, G, res)
        rows[name] = {"queries": 1024, "nodes_expanded": int(res["nodes_expanded"].sum()),
                      "status_counts": {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))},
                      "queries_that_differ_from_the_oracle": [int(i) for i in bad]}
        print(f"{name}: {len(bad)} differ, {time.perf_counter() - t1:.0f} s", file=sys.stderr, flush=True)
        h.close()
    line = json.dumps({"workload": "map_maze, the 1024 queries of seed 20250228, max_iterations 10000, one configuration per row", "rows": rows})
    print(line)
    if a.out:
        Path(a.out).write_text(line + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=16)
    ap.add_argument("--maze-modes", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    if not (orc.ORACLE_DIR / "liboracle.so").exists():
        orc.build_oracle()
    if a.maze_modes:
        return maze_modes(a)
    o, h, mp = _pair("map_large")
    H, W = mp.shape
    S, G = make_queries(h.collision, 65536, 3, W, H)
    pick = np.arange(0, 65536, a.every)
    S, G = S[pick], G[pick]
    t0 = time.perf_counter()
    res = h.plan_batch(S, G)
    t_gpu = time.perf_counter() - t0
    bad, t1 = [], time.perf_counter()
    # in slices, so that a line appears every half minute or so

    class Slice:  # the batch's outputs of queries [b, e), addressed from 0 as _compare_threaded expects
        def __init__(self, b):
            self.b = b

        def path(self, i):
            return h.path(self.b + i)

        def expanded(self, i):
            return h.expanded(self.b + i)

    for b in range(0, len(pick), 512):
        e = min(len(pick), b + 512)
        bad += [b + i for i in _compare_threaded(o, Slice(b), S[b:e], G[b:e], res[b:e])]
        print(f"queries {b}..{e}: {len(bad)} differ so far, {time.perf_counter() - t1:.0f} s", file=sys.stderr, flush=True)
    line = json.dumps({"workload": f"map_large, every {a.every}th of the 65536 queries of seed 3, max_iterations 10000", "queries": int(len(pick)),
                       "nodes_expanded": int(res["nodes_expanded"].sum()), "gpu_seconds": t_gpu, "oracle_seconds": time.perf_counter() - t1,
                       "status_counts": {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))},
                       "queries_that_differ_from_the_oracle": [int(pick[i]) for i in bad]})
    print(line)
    if a.out:
        Path(a.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
