"""Host-side mirror of the reference's plugin interface
(include/hybrid_a_star_planner.h:26-58, src/hybrid_a_star_planner.cpp).

Same class and method names, argument meaning and error behaviour as
HybridAStar::HybridAStarPlanner : nav_core::BaseGlobalPlanner, with the
ROS message types reduced to what the planner reads from them.  The 1 s sleep
of Clear() (src/hybrid_a_star_planner.cpp:184-192) is not reproduced.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from ._lib import STATUS_BAD_POSE, STATUS_NO_PATH, Hyas, default_params


@dataclass
class PoseStamped:
    """geometry_msgs/PoseStamped reduced to position + planar orientation."""

    x: float = 0.0
    y: float = 0.0
    qz: float = 0.0
    qw: float = 1.0
    frame_id: str = "map"

    @staticmethod
    def from_yaw(x, y, yaw, frame_id="map"):
        # tf::createQuaternionFromYaw (src/utility.cpp:23)
        return PoseStamped(float(x), float(y), math.sin(yaw / 2.0), math.cos(yaw / 2.0), frame_id)

    def yaw(self):
        # tf::getYaw for a planar quaternion (src/utility.cpp:11)
        return math.atan2(2.0 * self.qw * self.qz, 1.0 - 2.0 * self.qz * self.qz)


@dataclass
class Costmap2D:
    """costmap_2d::Costmap2D as the planner uses it: char map + geometry."""

    charmap: np.ndarray
    resolution: float = 1.0
    origin_x: float = 0.0
    origin_y: float = 0.0
    frame_id: str = "map"

    def getSizeInCellsX(self):
        return int(self.charmap.shape[1])

    def getSizeInCellsY(self):
        return int(self.charmap.shape[0])

    def getResolution(self):
        return self.resolution

    def getOriginX(self):
        return self.origin_x

    def getOriginY(self):
        return self.origin_y

    def getCharMap(self):
        return self.charmap


@dataclass
class HybridAStarPlanner:
    """nav_core::BaseGlobalPlanner implementation on one MI355X."""

    params: object = None
    device: int = 0
    initialized_: bool = False
    frame_id_: str = "map"
    last_result: object = None
    published_plans: list = field(default_factory=list)

    def initialize(self, name, costmap, frame_id=None):
        """initialize(name, Costmap2D*, frame_id), src/hybrid_a_star_planner.cpp:54-87.
        A second call is ignored (:43-51)."""
        if self.initialized_:
            return
        self.name = name
        self.costmap = costmap
        self.frame_id_ = frame_id or costmap.frame_id
        # without explicit parameters: the reference's struct defaults, as its ParameterManager starts from
        self._ctx = Hyas(self.params if self.params is not None else default_params(), self.device)
        self._ctx.set_costmap(costmap.getCharMap(), costmap.getResolution(), costmap.getOriginX(),
                              costmap.getOriginY(), build_table=True)
        self.initialized_ = True

    def makePlan(self, start, goal, plan):
        """bool makePlan(start, goal, plan&), src/hybrid_a_star_planner.cpp:99-124: clears `plan`,
        returns False when uninitialised, when a pose check fails or when no path exists."""
        if not self.initialized_:
            return False
        plan.clear()
        res = self._ctx.plan_batch([[start.x, start.y, start.yaw()]], [[goal.x, goal.y, goal.yaw()]])[0]
        self.last_result = res
        if res["status"] in (STATUS_BAD_POSE, STATUS_NO_PATH) or res["path_len"] == 0:
            return False
        for x, y, t in self._ctx.path(0):
            p = PoseStamped.from_yaw(x, y, t, "map")  # Utility::TypeConversion, src/utility.cpp:21-35
            plan.append(p)
        self.publishPlan(plan)
        return True

    def makePlanBatch(self, starts_xyt, goals_xyt):
        """n independent makePlan calls in one launch sequence; returns (results, paths)."""
        if not self.initialized_:
            raise RuntimeError("The planner has not been initialized, please call initialize() to use the planner")
        res = self._ctx.plan_batch(starts_xyt, goals_xyt)
        return res, [self._ctx.path(i) for i in range(len(res))]

    def makePlanService(self, req_start, req_goal):
        """make_plan service body (:89-97): always answers, the plan may be empty."""
        plan = []
        self.makePlan(req_start, req_goal, plan)
        return plan

    def publishPlan(self, path):
        """nav_msgs::Path on ~/plan (:157-183); kept as a list for callers without ROS."""
        self.published_plans.append(list(path))
