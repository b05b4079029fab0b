"""Worker of tests/test_dist_cpu.py: one of `world` gloo ranks on the CPU."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import orc  # noqa: E402
from path_planner_amd import sharding  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {}
    # 1. time/work aggregation used by bench.py
    t, counts = sharding.aggregate(dist, 1.0 + rank, [10 * (rank + 1), 7])
    res["agg_t"], res["agg_counts"] = t, counts
    # 2. heading-sharded table: every rank builds its slices with the CPU oracle, the all-gather assembles them
    o = orc.Oracle(orc.hot_path_params(position_resolution=1, cd_min_turning_radius=0.35))
    o.set_costmap(orc.load_map("map_small"), build_table=False)
    nx, ny, ns = o.table_dims()
    slices, per = sharding.heading_slices(ns, world)
    k0, k1 = slices[rank]
    local = torch.zeros((per, ny, nx), dtype=torch.float32)
    if k1 > k0:
        local[: k1 - k0] = torch.from_numpy(o.table_build(1, k0, k1))
    full = sharding.gather_slices(local, ns, dist)
    want = o.table_build(1, 0, ns)
    res["table_equal"] = bool(np.array_equal(full.numpy().view(np.uint32), want.view(np.uint32)))
    res["table_shape"] = list(full.shape)
    # 3. query shards: every rank plans its block with the oracle; rank 0 checks the blocks tile the batch
    b, e = sharding.query_shard(11, rank, world)
    res["shard"] = [b, e]
    (out_dir / f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
