"""Short product-only runs for counter collection and kernel traces (see collect_pmc.sh).

  python3 profiles/workload.py maze        1024 maze queries, 300 iterations at most (k_round, k_flood)
  python3 profiles/workload.py footprint   5 launches of 10 M footprint checks, in-tree footprint (k_collision_footprint_window<5>)
  python3 profiles/workload.py footprint_car   the same with the 2 x 4.5 cell footprint (<9>)
  python3 profiles/workload.py table_pr1 | table_pr10   one build of the 180 x 60 Reeds-Shepp table (k_table_rs, k_table_install)
  python3 profiles/workload.py flood_large  2D floods of 2048 goals on map_large (k_flood)
"""
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from path_planner_amd import Hyas, default_params, hot_path_params  # noqa: E402
from path_planner_amd._lib import MODE_FOOTPRINT  # noqa: E402
from path_planner_amd.sharding import load_map, make_queries  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "maze"
if what == "maze":
    h = Hyas(hot_path_params(max_iterations=300))
    h.set_costmap(load_map("map_maze"), build_table=True)
    S, G = make_queries(h.collision, 1024, 20250228, 50, 50)
    res = h.plan_batch(S, G)
    print(int(res["nodes_expanded"].sum()), h.kernel_ms(0))
elif what in ("footprint", "footprint_car"):
    import torch

    W = H = 4096
    n = 10_000_000
    grid = np.where(np.random.default_rng(1).random((H, W)) < 0.10, 254, 0).astype(np.uint8)
    rng = np.random.default_rng(2)
    xyt = np.stack([rng.uniform(8, W - 8, n), rng.uniform(8, H - 8, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
    fp = (0.18, 0.22) if what == "footprint" else (2.0, 4.5)
    h = Hyas(default_params(fp_width=fp[0], fp_length=fp[1]))
    h.set_grid(grid)
    d_xyt = torch.from_numpy(xyt).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        h.collision_dev(d_xyt.data_ptr(), None, n, d_out.data_ptr(), MODE_FOOTPRINT)
    print(int(d_out.sum().item()), h.kernel_ms(2))
elif what in ("table_pr1", "table_pr10"):
    pr = 10 if what.endswith("10") else 1
    h = Hyas(hot_path_params(position_resolution=pr, cd_min_turning_radius=0.35))
    h.set_costmap(np.zeros((60, 180), np.uint8), build_table=True)
    nx, ny, ns = h.table_dims()
    print(nx * ny * ns, h.kernel_ms(3))
elif what == "flood_large":
    mp = load_map("map_large")
    H, W = mp.shape
    h = Hyas(hot_path_params(position_resolution=1, max_iterations=1))
    h.set_costmap(mp, build_table=True)
    import os

    S, G = make_queries(h.collision, int(os.environ.get("FLOOD_QUERIES", "2048")), 3, W, H)
    res = h.plan_batch(S, G)
    print(len(res))
else:
    raise SystemExit(f"unknown workload {what}")
