"""Short run of the benchmark workload for counter collection (see collect_traffic.sh)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import orc  # map fixture loader only
from path_planner_amd import Hyas, hot_path_params
from path_planner_amd.sharding import make_queries

h = Hyas(hot_path_params(max_iterations=300))
h.set_costmap(orc.load_map("map_maze"), build_table=True)
S, G = make_queries(h.collision, 1024, 20250228, 50, 50)
res = h.plan_batch(S, G)
print(int(res["nodes_expanded"].sum()), h.kernel_ms(0))
