"""MI355X-native Hybrid A* expansion path (drop-in for JialiangHan/path_planner's
HybridAStarPlanner hot path).  All compute runs in libhyas.so (hand-written HIP
for gfx950) behind the C ABI of include/hyas.h; this package is the thin host
mirror of the reference's plugin interface."""
from ._lib import (GOAL_DT, NODE_DT, RESULT_DT, SUCC_DT, Hyas, HyasError, Params, default_params,
                   hot_path_params, lib_path, load_library)
from .planner import Costmap2D, HybridAStarPlanner, PoseStamped

__all__ = ["Hyas", "HyasError", "Params", "default_params", "hot_path_params", "lib_path", "load_library",
           "HybridAStarPlanner", "Costmap2D", "PoseStamped", "NODE_DT", "GOAL_DT", "SUCC_DT", "RESULT_DT"]
