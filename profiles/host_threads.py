"""Throughput of the query workloads against the host threads a rank may use (HYAS_THREADS), on one GPU: what one rank
can expect when a node's cores are divided among eight ranks.

    python profiles/host_threads.py [--out FILE]

Every point is a process of its own (the library reads HYAS_THREADS when its worker pools are made): the maze batch of
bench.py (1024 queries, 3 steps after 1 warm-up, median) and 16384 queries of the map_large batch (one call).  Prints
one JSON line."""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
from path_planner_amd import Hyas, hot_path_params
from path_planner_amd.sharding import load_map, make_queries
what = sys.argv[1]
if what == "maze":
    mp = load_map("map_maze"); nq, seed, steps = 1024, 20250228, 3
else:
    mp = load_map("map_large"); nq, seed, steps = 16384, 3, 1
H, W = mp.shape
ctx = Hyas(hot_path_params())
ctx.set_costmap(mp, build_table=True)
S, G = make_queries(ctx.collision, nq, seed, W, H)
ctx.plan_batch(S[:1024], G[:1024])
t = []
for _ in range(steps):
    t0 = time.perf_counter()
    res = ctx.plan_batch(S, G)
    t.append(time.perf_counter() - t0)
sec = float(np.median(t))
print(json.dumps({"queries": nq, "seconds": sec, "queries_per_s": nq / sec, "successors_per_s": float(res["successors"].sum()) / sec}))
""" % str(ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    a = ap.parse_args()
    rows = {"maze_1024_queries": {}, "map_large_16384_queries": {}}
    for threads in (2, 4, 8, 16):
        for key, what in (("maze_1024_queries", "maze"), ("map_large_16384_queries", "large")):
            env = dict(os.environ, HYAS_THREADS=str(threads))
            out = subprocess.run([sys.executable, "-c", CHILD, what], env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                raise SystemExit(out.stderr[-2000:])
            rows[key][str(threads)] = json.loads(out.stdout.strip().splitlines()[-1])
            print(key, threads, rows[key][str(threads)], file=sys.stderr, flush=True)
    line = json.dumps({"workload": "query batches against HYAS_THREADS, 1 x MI355X, one process per point", "cpus_of_the_box": len(os.sched_getaffinity(0)),
                       "rows": rows})
    print(line)
    if a.out:
        Path(a.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
