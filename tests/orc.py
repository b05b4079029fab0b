"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- test helper only.

The oracle is the checker, never the product: nothing under path_planner_amd/
imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"


class Params(C.Structure):
    """Mirror of hyas_params (include/hyas.h)."""

    _fields_ = [
        ("position_resolution", C.c_int32),
        ("headings", C.c_int32),
        ("cd_min_turning_radius", C.c_float),
        ("curve_type", C.c_int32),
        ("vehicle_width", C.c_float),
        ("vehicle_length", C.c_float),
        ("obstacle_detection_range", C.c_float),
        ("map_boundary_obstacle", C.c_int32),
        ("number_of_successors", C.c_int32),
        ("adaptive_steering_angle_and_step_size", C.c_int32),
        ("adaptive_step_size", C.c_int32),
        ("analytical_expansion_every_point", C.c_int32),
        ("analytical_expansion", C.c_int32),
        ("penalty_turning", C.c_float),
        ("penalty_reverse", C.c_float),
        ("penalty_change_of_direction", C.c_float),
        ("reverse", C.c_int32),
        ("max_iterations", C.c_int32),
        ("steering_angle", C.c_float),
        ("goal_range", C.c_float),
        ("tie_breaker", C.c_float),
        ("min_turning_radius", C.c_float),
        ("fp_width", C.c_double),
        ("fp_length", C.c_double),
        ("fp_cell_size", C.c_float),
        ("short_cut", C.c_int32),
        ("max_steering_angle", C.c_float),
        ("consider_steering_angle_range_for_obstacle_density", C.c_int32),
        ("add_one_more_successor_only_in_free_angle_range", C.c_int32),
        ("add_one_more_successor", C.c_int32),
        ("steering_angle_towards_free_angle_range_for_obstacle_angle_range", C.c_int32),
        ("fixed_number_of_steering_angle_in_free_angle_range", C.c_int32),
        ("constant_density", C.c_float),
        ("piecewise_cubic_bezier_interpolation", C.c_int32),
        ("curve_step_size", C.c_float),
        ("reserved", C.c_int32),
    ]


NODE_DT = np.dtype([("x", "f4"), ("y", "f4"), ("t", "f4"), ("g", "f4"), ("prim", "i4")])
GOAL_DT = np.dtype([("x", "f4"), ("y", "f4"), ("t", "f4")])
SUCC_DT = np.dtype(
    [("x", "f4"), ("y", "f4"), ("t", "f4"), ("g", "f4"), ("h", "f4"), ("idx", "i4"), ("prim", "i1"), ("free", "u1"), ("pad", "u2")]
)
RESULT_DT = np.dtype(
    [
        ("status", "i4"),
        ("nodes_explored", "i4"),
        ("nodes_expanded", "i4"),
        ("successors", "i4"),
        ("path_len", "i4"),
        ("g_final", "f4"),
        ("shot_length", "f4"),
        ("path_length", "f4"),
    ]
)
assert SUCC_DT.itemsize == 28 and NODE_DT.itemsize == 20 and RESULT_DT.itemsize == 32


def default_params(**kw):
    """Struct defaults of the reference (include/parameter_manager.h:10-42, :124-175)."""
    p = Params(
        position_resolution=10, headings=72, cd_min_turning_radius=6.0, curve_type=0, vehicle_width=1.0,
        vehicle_length=2.0, obstacle_detection_range=5.0, map_boundary_obstacle=0, number_of_successors=3,
        adaptive_steering_angle_and_step_size=0, adaptive_step_size=1, analytical_expansion_every_point=1,
        analytical_expansion=1, penalty_turning=1.05, penalty_reverse=1.0, penalty_change_of_direction=2.0,
        reverse=0, max_iterations=10000, steering_angle=5.0, goal_range=1e-3, tie_breaker=0.01,
        min_turning_radius=6.0, fp_width=0.18, fp_length=0.22, fp_cell_size=1.0, short_cut=1, max_steering_angle=30.0,
        consider_steering_angle_range_for_obstacle_density=0, add_one_more_successor_only_in_free_angle_range=0,
        add_one_more_successor=1, steering_angle_towards_free_angle_range_for_obstacle_angle_range=1,
        fixed_number_of_steering_angle_in_free_angle_range=1, constant_density=0.5, piecewise_cubic_bezier_interpolation=0,
        curve_step_size=1.0, reserved=0,
    )
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def hot_path_params(**kw):
    """Fixed-step Reeds-Shepp configuration the benchmarks are quoted on."""
    base = dict(curve_type=1, number_of_successors=5, steering_angle=5.0, reverse=1,
                analytical_expansion=1, analytical_expansion_every_point=1, short_cut=0, adaptive_step_size=0)
    base.update(kw)
    return default_params(**base)


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "liboracle.so", "liboracle_libm.so"], check=True)


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Oracle:
    def __init__(self, params, libm=False):
        name = "liboracle_libm.so" if libm else "liboracle.so"
        path = ORACLE_DIR / name
        if not path.exists():
            build_oracle()
        L = C.CDLL(str(path))
        self.L = L
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Params)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_costmap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_table_checksum.restype = C.c_double
        L.orc_table_checksum.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_rs_distance.restype = C.c_double
        L.orc_rs_distance.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
        L.orc_utility.restype = C.c_double
        L.orc_utility.argtypes = [C.c_int, C.c_void_p]
        L.orc_plan_many.restype = C.c_double
        for f in ("orc_collision_batch", "orc_expand_batch", "orc_shot_batch", "orc_flood", "orc_table_build",
                  "orc_table_lookup", "orc_plan", "orc_get_path", "orc_get_expanded", "orc_footprint_table",
                  "orc_get_obstacles", "orc_in_range_count", "orc_in_range_list", "orc_num_obstacles", "orc_num_successors",
                  "orc_table_dims", "orc_build_table", "orc_rs_path", "orc_rs_sample", "orc_dubins",
                  "orc_dubins_sample", "orc_plan_many", "orc_twod_size", "orc_set_grid", "orc_plan_r", "orc_shortcut", "orc_expand_batch_from",
                  "orc_step_steer_pairs", "orc_adaptive_cell"):
            getattr(L, f).argtypes = None
        self.params = params
        self.h = C.c_void_p(L.orc_create(C.byref(params)))
        self.w = self.hgt = 0

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_costmap(self, charmap, res=1.0, ox=0.0, oy=0.0, build_table=True, nthreads=None):
        cm = np.ascontiguousarray(charmap, dtype=np.uint8)
        self.hgt, self.w = cm.shape
        nthreads = nthreads or min(8, os.cpu_count() or 1)
        self.L.orc_set_costmap(self.h, cm.ctypes.data_as(C.c_void_p), C.c_uint32(self.w), C.c_uint32(self.hgt),
                               C.c_double(res), C.c_double(ox), C.c_double(oy), int(build_table), int(nthreads))

    def set_grid(self, charmap):
        charmap = np.ascontiguousarray(charmap, np.uint8)
        self.hgt, self.w = charmap.shape
        self.L.orc_set_grid(self.h, charmap.ctypes.data_as(C.c_void_p), C.c_uint32(self.w), C.c_uint32(self.hgt))

    def obstacles(self):
        n = self.L.orc_num_obstacles(self.h)
        out = np.zeros((n, 8), np.float32)
        self.L.orc_get_obstacles(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        return out

    def in_range_count(self, key):
        return self.L.orc_in_range_count(self.h, C.c_uint32(key))

    def in_range_list(self, key):
        n = max(0, self.in_range_count(key))
        out = np.zeros((n, 8), np.float32)
        self.L.orc_in_range_list(self.h, C.c_uint32(key), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        return out

    def collision(self, xyt, pred_xy=None, mode=0):
        xyt = np.ascontiguousarray(xyt, np.float32)
        n = xyt.shape[0]
        out = np.zeros(n, np.uint8)
        pp = None
        if pred_xy is not None:
            pred_xy = np.ascontiguousarray(pred_xy, np.float32)
            pp = pred_xy.ctypes.data_as(C.c_void_p)
        self.L.orc_collision_batch(self.h, int(mode), xyt.ctypes.data_as(C.c_void_p), pp, C.c_size_t(n),
                                   out.ctypes.data_as(C.c_void_p))
        return out

    def footprint_table(self):
        n = self.L.orc_footprint_table(self.h, None, C.c_size_t(0))
        out = np.zeros((n, 129), np.int32)
        self.L.orc_footprint_table(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        return out

    def num_successors(self):
        return self.L.orc_num_successors(self.h)

    def expand(self, preds, goals, twod=None, starts=None):
        preds = np.ascontiguousarray(preds, NODE_DT)
        goals = np.ascontiguousarray(goals, GOAL_DT)
        n = preds.shape[0]
        K = self.num_successors()
        out = np.zeros((n, K), SUCC_DT)
        cnt = np.zeros(n, np.uint8)
        tp = sp = None
        if twod is not None:
            twod = np.ascontiguousarray(twod, np.float32)
            tp = twod.ctypes.data_as(C.c_void_p)
        if starts is not None:
            starts = np.ascontiguousarray(starts, GOAL_DT)
            sp = starts.ctypes.data_as(C.c_void_p)
        self.L.orc_expand_batch_from(self.h, preds.ctypes.data_as(C.c_void_p), goals.ctypes.data_as(C.c_void_p), sp, tp,
                                     C.c_size_t(n), out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
        return out, cnt

    def step_steer_pairs(self, pred, goal, start, cap=64):
        """{step size, steering angle} pairs of one node (reverse pair included), as CreateSuccessor forms them."""
        a, g, s = np.zeros(1, NODE_DT), np.zeros(1, GOAL_DT), np.zeros(1, GOAL_DT)
        a["x"], a["y"], a["t"] = pred
        g["x"], g["y"], g["t"] = goal
        s["x"], s["y"], s["t"] = start
        out = np.zeros((cap, 2), np.float32)
        n = self.L.orc_step_steer_pairs(self.h, a.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                                        s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), cap)
        return out[:min(n, cap)], n

    def adaptive_cell(self, key, cap=256, cut=False):
        """(min distance, normalised density, [(distance, start, range)...]) of one grid cell, or None; cut: the ranges
        as FindFreeAngleRangeAndObstacleAngleRange hands them on (cut to the circle less 0.0001 degrees)."""
        md, de = C.c_float(), C.c_float()
        r = np.zeros((cap, 3), np.float32)
        n = self.L.orc_adaptive_cell(self.h, C.c_uint32(key), C.byref(md), C.byref(de), r.ctypes.data_as(C.c_void_p), cap, int(cut))
        if n < 0:
            return None
        return md.value, de.value, r[:n].copy()

    def shot(self, nodes, goals):
        nodes = np.ascontiguousarray(nodes, NODE_DT)
        goals = np.ascontiguousarray(goals, GOAL_DT)
        n = nodes.shape[0]
        ok = np.zeros(n, np.uint8)
        ln = np.zeros(n, np.float32)
        self.L.orc_shot_batch(self.h, nodes.ctypes.data_as(C.c_void_p), goals.ctypes.data_as(C.c_void_p),
                              C.c_size_t(n), ok.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p))
        return ok, ln

    def flood(self, goal_xyt):
        g = np.zeros(1, GOAL_DT)
        g["x"], g["y"], g["t"] = goal_xyt
        out = np.zeros(self.L.orc_twod_size(self.h), np.float32)
        self.L.orc_flood(self.h, g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def table_dims(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orc_table_dims(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def table_build(self, curve, b, e):
        nx, ny, ns = self.table_dims()
        out = np.zeros((e - b, ny, nx), np.float32)
        self.L.orc_table_build(self.h, int(curve), int(b), int(e), out.ctypes.data_as(C.c_void_p))
        return out

    def build_table(self, curve, nthreads=8):
        self.L.orc_build_table(self.h, int(curve), int(nthreads))

    def table_lookup(self, xyt):
        xyt = np.ascontiguousarray(xyt, np.float32)
        out = np.zeros(xyt.shape[0], np.float32)
        self.L.orc_table_lookup(self.h, xyt.ctypes.data_as(C.c_void_p), C.c_size_t(xyt.shape[0]),
                                out.ctypes.data_as(C.c_void_p))
        return out

    def table_checksum(self):
        n = C.c_int64()
        s = self.L.orc_table_checksum(self.h, C.byref(n))
        return s, n.value

    def plan(self, start, goal, path_cap=20000):
        s = np.asarray(start, np.float32)
        g = np.asarray(goal, np.float32)
        res = np.zeros(1, RESULT_DT)
        self.L.orc_plan(self.h, s.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p))
        path = np.zeros((path_cap, 3), np.float32)
        n = self.L.orc_get_path(self.h, path.ctypes.data_as(C.c_void_p), C.c_size_t(path_cap))
        exp = np.zeros(max(1, int(res["nodes_expanded"][0])), np.int32)
        m = self.L.orc_get_expanded(self.h, exp.ctypes.data_as(C.c_void_p), C.c_size_t(exp.shape[0]))
        return res[0], path[:n].copy(), exp[:m].copy()

    def plan_r(self, start, goal, path_cap=40000, exp_cap=12000):
        """plan() with caller-owned outputs: safe to call from several threads on one context."""
        s = np.asarray(start, np.float32)
        g = np.asarray(goal, np.float32)
        res = np.zeros(1, RESULT_DT)
        path = np.zeros((path_cap, 3), np.float32)
        exp = np.zeros(exp_cap, np.int32)
        pn, en = C.c_int32(), C.c_int32()
        self.L.orc_plan_r(self.h, s.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p),
                          path.ctypes.data_as(C.c_void_p), C.c_size_t(path_cap), C.byref(pn), exp.ctypes.data_as(C.c_void_p),
                          C.c_size_t(exp_cap), C.byref(en))
        assert pn.value <= path_cap and en.value <= exp_cap
        return res[0], path[:pn.value].copy(), exp[:en.value].copy()

    def shortcut(self, path):
        path = np.ascontiguousarray(path, np.float32).reshape(-1, 3)
        idx = np.zeros(len(path) + 2, np.uint32)
        n = self.L.orc_shortcut(self.h, path.ctypes.data_as(C.c_void_p), C.c_uint32(len(path)), idx.ctypes.data_as(C.c_void_p))
        return idx[:n].copy()

    def plan_many(self, starts, goals):
        starts = np.ascontiguousarray(starts, np.float32)
        goals = np.ascontiguousarray(goals, np.float32)
        n = starts.shape[0]
        res = np.zeros(n, RESULT_DT)
        sec = self.L.orc_plan_many(self.h, starts.ctypes.data_as(C.c_void_p), goals.ctypes.data_as(C.c_void_p),
                                   C.c_size_t(n), res.ctypes.data_as(C.c_void_p))
        return sec, res

    # raw function access -------------------------------------------------
    def rs_distance(self, rho, q0, q1):
        a = np.asarray(q0, np.float64)
        b = np.asarray(q1, np.float64)
        return self.L.orc_rs_distance(C.c_double(rho), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))

    def rs_path(self, rho, q0, q1):
        a = np.asarray(q0, np.float64)
        b = np.asarray(q1, np.float64)
        ty = C.c_int()
        ln = np.zeros(5, np.float64)
        self.L.orc_rs_path(C.c_double(rho), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(ty),
                           ln.ctypes.data_as(C.c_void_p))
        return ty.value, ln

    def rs_sample(self, rho, q0, q1, step, cap=100000):
        a = np.asarray(q0, np.float64)
        b = np.asarray(q1, np.float64)
        length = C.c_double()
        pts = np.zeros((cap, 3), np.float64)
        n = self.L.orc_rs_sample(C.c_double(rho), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                 C.c_double(step), C.byref(length), pts.ctypes.data_as(C.c_void_p), C.c_size_t(cap))
        return length.value, pts[:n].copy()

    def dubins(self, q0, q1, rho):
        a = np.asarray(q0, np.float32)
        b = np.asarray(q1, np.float32)
        out = np.zeros(5, np.float32)
        rc = self.L.orc_dubins(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_float(rho),
                               out.ctypes.data_as(C.c_void_p))
        return rc, out

    def dubins_sample(self, q0, q1, rho, t):
        a = np.asarray(q0, np.float32)
        b = np.asarray(q1, np.float32)
        q = np.zeros(3, np.float32)
        rc = self.L.orc_dubins_sample(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_float(rho), C.c_float(t),
                                      q.ctypes.data_as(C.c_void_p))
        return rc, q

    def utility(self, op, args):
        a = np.asarray(args, np.float32)
        return self.L.orc_utility(int(op), a.ctypes.data_as(C.c_void_p))


def load_map(name):
    return np.load(ROOT / "tests" / "golden" / "maps" / f"{name}.npy")
