"""ms per call of hyas_plan_batch on k maze queries at a time (k = 1 .. 1024)."""
import json, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, ".")
from path_planner_amd import Hyas, hot_path_params
from path_planner_amd.sharding import load_map, make_queries

mp = load_map("map_maze")
ctx = Hyas(hot_path_params())
ctx.set_costmap(mp, build_table=True)
S, G = make_queries(ctx.collision, 1024, 20250228, 50, 50)
ctx.plan_batch(S[:64], G[:64])
out = {}
for k in (1, 8, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    span = 128 if k < 64 else (768 if k in (96, 192, 384, 768) else 1024)
    t0 = time.perf_counter()
    for i in range(0, span, k):
        ctx.plan_batch(S[i:i + k], G[i:i + k])
    out[k] = round((time.perf_counter() - t0) * 1e3 / (span / k), 2)
print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith("HYAS_")}, ms_per_call=out)))
