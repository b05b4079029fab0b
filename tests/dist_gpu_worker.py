"""Worker of tests/test_gpu_configs.py::test_sharded_table_from_two_processes: one of `world` ranks that all use GPU 0,
joined over gloo (a one-GPU box has no second device for RCCL).  Every rank builds its heading slices of the
Reeds-Shepp table with the HIP kernel, the all-gather assembles them, every rank installs the table and answers the
same lookups."""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import orc  # noqa: E402
from path_planner_amd import Hyas, hot_path_params, sharding  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    ctx = Hyas(hot_path_params(position_resolution=2, cd_min_turning_radius=0.35), device=0)
    ctx.set_costmap(orc.load_map("map_parking"), build_table=False)
    times = sharding.build_table_sharded(ctx, 1, dist, device=torch.device("cuda", 0))
    q = np.load(out_dir / "lookups.npy")
    np.save(out_dir / f"rank{rank}.npy", ctx.table_lookup(q))
    (out_dir / f"rank{rank}.json").write_text(json.dumps({"times": times, "dims": ctx.table_dims()}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
