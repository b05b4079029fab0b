"""ctypes binding of libhyas.so (include/hyas.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable,
every compute entry point raises HyasError.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent


class HyasError(RuntimeError):
    pass


class Params(C.Structure):
    """hyas_params (include/hyas.h); names follow include/parameter_manager.h:10-42, :124-175."""

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
SUCC_DT = np.dtype([("x", "f4"), ("y", "f4"), ("t", "f4"), ("g", "f4"), ("h", "f4"), ("idx", "i4"),
                    ("prim", "i1"), ("free", "u1"), ("pad", "u2")])
RESULT_DT = np.dtype([("status", "i4"), ("nodes_explored", "i4"), ("nodes_expanded", "i4"), ("successors", "i4"),
                      ("path_len", "i4"), ("g_final", "f4"), ("shot_length", "f4"), ("path_length", "f4")])

STATUS_NO_PATH, STATUS_GOAL, STATUS_SHOT, STATUS_MAX_ITER, STATUS_BAD_POSE, STATUS_OVERFLOW = range(6)
# return codes of include/hyas.h
OK, EINVAL, ENOMEM, ENODEV, ESTATE, ERANGE = 0, -22, -12, -19, -1, -34
MODE_POLYGON, MODE_FOOTPRINT = 0, 1

# every symbol include/hyas.h declares
EXPORTS = [
    "hyas_default_params", "hyas_hot_path_params", "hyas_create", "hyas_destroy", "hyas_last_error",
    "hyas_set_costmap", "hyas_set_grid", "hyas_get_obstacles", "hyas_in_range_count", "hyas_in_range_list", "hyas_collision_batch",
    "hyas_collision_batch_dev", "hyas_footprint_table", "hyas_expand_batch", "hyas_num_successors",
    "hyas_shot_batch", "hyas_twod_size", "hyas_flood", "hyas_table_dims", "hyas_table_build_dev",
    "hyas_table_install_dev", "hyas_table_lookup", "hyas_plan_batch", "hyas_get_path", "hyas_get_expanded",
    "hyas_last_kernel_ms", "hyas_set_round_timing", "hyas_reset_kernel_timers", "hyas_set_stats", "hyas_get_stats", "hyas_nodes_in_device_memory",
    "hyas_table_install_noalloc_dev", "hyas_shortcut_batch", "hyas_expand_batch_from", "hyas_adaptive_cell",
]

_lib = None


def lib_path():
    import os

    return Path(os.environ.get("HYAS_LIB") or (_HERE / "libhyas.so"))


def load_library():
    """Loads libhyas.so (built in-tree by path_planner_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise HyasError(f"{p} is missing: build it with `make -C path_planner_amd/csrc` "
                        "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    # Kernel arguments in device memory: the HIP runtime reads this switch when it starts up (its first call), and it
    # shortens every launch by 2-3 us, which a search round (one launch per round trip) notices.  A value that is
    # already set is left alone.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    try:
        # torch bundles its own HIP runtime; if libhyas.so pulled in the system one first, a later
        # `import torch` would bring a second runtime into the process and find no GPU.  Loading torch
        # first makes both share one.
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(str(p))
    except OSError as e:  # e.g. libamdhip64 not found
        raise HyasError(f"cannot load {p}: {e}") from e
    L.hyas_create.restype = C.c_void_p
    L.hyas_create.argtypes = [C.POINTER(Params), C.c_int]
    L.hyas_destroy.argtypes = [C.c_void_p]
    L.hyas_last_error.restype = C.c_char_p
    L.hyas_last_error.argtypes = [C.c_void_p]
    L.hyas_default_params.argtypes = [C.POINTER(Params)]
    L.hyas_hot_path_params.argtypes = [C.POINTER(Params)]
    L.hyas_set_costmap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_int]
    L.hyas_set_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.hyas_get_obstacles.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.hyas_in_range_count.argtypes = [C.c_void_p, C.c_uint32]
    L.hyas_in_range_list.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.hyas_collision_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.hyas_collision_batch_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.hyas_footprint_table.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.hyas_expand_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.hyas_num_successors.argtypes = [C.c_void_p]
    L.hyas_shot_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.hyas_twod_size.argtypes = [C.c_void_p]
    L.hyas_flood.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hyas_table_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.hyas_table_build_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.hyas_table_install_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.hyas_table_install_noalloc_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.hyas_table_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.hyas_plan_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.hyas_shortcut_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.hyas_get_path.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.hyas_get_expanded.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.hyas_last_kernel_ms.restype = C.c_double
    L.hyas_last_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    L.hyas_reset_kernel_timers.argtypes = [C.c_void_p]
    L.hyas_set_stats.argtypes = [C.c_void_p, C.c_int]
    L.hyas_get_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.hyas_nodes_in_device_memory.argtypes = [C.c_void_p]
    L.hyas_set_round_timing.argtypes = [C.c_void_p, C.c_int]
    L.hyas_expand_batch_from.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.hyas_adaptive_cell.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_size_t]
    _lib = L
    return L


def default_params(**kw):
    p = Params()
    load_library().hyas_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def hot_path_params(**kw):
    p = Params()
    load_library().hyas_hot_path_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# The parameter file the reference ships (param/parameters.yaml:1-140), restricted to the keys this path reads.
# cd_min_turning_radius follows min_turning_radius: both are read from the same yaml key
# (src/parameter_manager.cpp:149-150, :233-234).
SHIPPED_YAML = dict(
    reverse=0, short_cut=1, piecewise_cubic_bezier_interpolation=1, number_of_successors=5, curve_type=2,
    adaptive_steering_angle_and_step_size=1, adaptive_step_size=0, constant_density=0.6, obstacle_detection_range=1.0,
    add_one_more_successor=0, add_one_more_successor_only_in_free_angle_range=1, map_boundary_obstacle=0,
    steering_angle_towards_free_angle_range_for_obstacle_angle_range=1, fixed_number_of_steering_angle_in_free_angle_range=1,
    analytical_expansion=1, analytical_expansion_every_point=0, consider_steering_angle_range_for_obstacle_density=0,
    position_resolution=1, vehicle_width=0.25, vehicle_length=0.5, max_steering_angle=30.0, tie_breaker=0.01, goal_range=1.0,
    penalty_turning=1.5, penalty_reverse=1.2, penalty_change_of_direction=1.0, steering_angle=3.0, curve_step_size=0.3,
    headings=72, max_iterations=100000, min_turning_radius=0.35, cd_min_turning_radius=0.35)


def shipped_yaml_params(**kw):
    """hyas_default_params overridden by the reference's shipped parameter file, then by kw."""
    return default_params(**{**SHIPPED_YAML, **kw})


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Hyas:
    """One planner context on one GPU (hyas_ctx)."""

    def __init__(self, params=None, device=0):
        self.L = load_library()
        self.params = params if params is not None else hot_path_params()
        h = self.L.hyas_create(C.byref(self.params), int(device))
        if not h:
            raise HyasError(f"hyas_create failed on device {device}: no usable GPU or unsupported parameters "
                            "(see stderr); this library has no CPU path")
        self.h = C.c_void_p(h)
        self.w = self.hgt = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.hyas_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise HyasError(f"{what} failed ({rc}): {self.L.hyas_last_error(self.h).decode()}")
        return rc

    # -- map ---------------------------------------------------------------
    def set_costmap(self, charmap, res=1.0, ox=0.0, oy=0.0, build_table=True):
        cm = np.ascontiguousarray(charmap, dtype=np.uint8)
        self.hgt, self.w = cm.shape
        self._check(self.L.hyas_set_costmap(self.h, _vp(cm), self.w, self.hgt, float(res), float(ox), float(oy),
                                            int(build_table)), "hyas_set_costmap")

    def set_grid(self, charmap):
        cm = np.ascontiguousarray(charmap, dtype=np.uint8)
        self.hgt, self.w = cm.shape
        self._check(self.L.hyas_set_grid(self.h, _vp(cm), self.w, self.hgt), "hyas_set_grid")

    def obstacles(self):
        n = self._check(self.L.hyas_get_obstacles(self.h, None, 0), "hyas_get_obstacles")
        out = np.zeros((n, 8), np.float32)
        self.L.hyas_get_obstacles(self.h, _vp(out), n)
        return out

    def in_range_count(self, key):
        return self.L.hyas_in_range_count(self.h, int(key))

    def in_range_list(self, key):
        n = max(0, self.L.hyas_in_range_count(self.h, int(key)))
        out = np.zeros((n, 8), np.float32)
        self.L.hyas_in_range_list(self.h, int(key), _vp(out), n)
        return out

    def footprint_table(self):
        n = self._check(self.L.hyas_footprint_table(self.h, None, 0), "hyas_footprint_table")
        out = np.zeros((n, 129), np.int32)
        self.L.hyas_footprint_table(self.h, _vp(out), n)
        return out

    # -- batched kernels -----------------------------------------------------
    def collision(self, xyt, pred_xy=None, mode=MODE_POLYGON):
        xyt = np.ascontiguousarray(xyt, np.float32).reshape(-1, 3)
        n = xyt.shape[0]
        out = np.zeros(n, np.uint8)
        pp = None
        if pred_xy is not None:
            pred_xy = np.ascontiguousarray(pred_xy, np.float32).reshape(-1, 2)
            assert pred_xy.shape[0] == n
            pp = _vp(pred_xy)
        self._check(self.L.hyas_collision_batch(self.h, int(mode), _vp(xyt), pp, n, _vp(out)), "hyas_collision_batch")
        return out

    def collision_dev(self, xyt_ptr, pred_ptr, n, out_ptr, mode=MODE_POLYGON):
        """Device pointers (e.g. torch tensor .data_ptr()); synchronous on the context's stream."""
        self._check(self.L.hyas_collision_batch_dev(self.h, int(mode), C.c_void_p(xyt_ptr),
                                                    C.c_void_p(pred_ptr) if pred_ptr else None, int(n),
                                                    C.c_void_p(out_ptr)), "hyas_collision_batch_dev")

    def num_successors(self):
        return self._check(self.L.hyas_num_successors(self.h), "hyas_num_successors")

    def twod_size(self):
        return self._check(self.L.hyas_twod_size(self.h), "hyas_twod_size")

    def expand(self, preds, goals, twod=None, starts=None):
        """Successors of n (node, goal) pairs; starts = the start pose of the search each pair belongs to (needed by
        adaptive_steering_angle_and_step_size).  count 0xFF marks a node with more candidates than its slots hold."""
        preds = np.ascontiguousarray(preds, NODE_DT)
        goals = np.ascontiguousarray(goals, GOAL_DT)
        n = preds.shape[0]
        K = self.num_successors()
        out = np.zeros((n, K), SUCC_DT)
        cnt = np.zeros(n, np.uint8)
        tp = None
        if twod is not None:
            twod = np.ascontiguousarray(twod, np.float32)
            assert twod.size == n * self.twod_size()
            tp = _vp(twod)
        if starts is not None:
            starts = np.ascontiguousarray(starts, GOAL_DT)
            assert starts.shape[0] == n
            self._check(self.L.hyas_expand_batch_from(self.h, _vp(preds), _vp(goals), _vp(starts), tp, n, _vp(out), _vp(cnt)),
                        "hyas_expand_batch_from")
        else:
            self._check(self.L.hyas_expand_batch(self.h, _vp(preds), _vp(goals), tp, n, _vp(out), _vp(cnt)),
                        "hyas_expand_batch")
        return out, cnt

    def adaptive_cell(self, key, cap=256):
        """(minimum obstacle distance, normalised density, [(distance, start, width), ...]) of one grid key."""
        md, de = C.c_float(), C.c_float()
        r = np.zeros((cap, 3), np.float32)
        n = self._check(self.L.hyas_adaptive_cell(self.h, C.c_uint32(int(key)), C.byref(md), C.byref(de), _vp(r), C.c_size_t(cap)),
                        "hyas_adaptive_cell")
        return md.value, de.value, r[:min(n, cap)].copy()

    def shot(self, nodes, goals):
        nodes = np.ascontiguousarray(nodes, NODE_DT)
        goals = np.ascontiguousarray(goals, GOAL_DT)
        n = nodes.shape[0]
        ok = np.zeros(n, np.uint8)
        ln = np.zeros(n, np.float32)
        self._check(self.L.hyas_shot_batch(self.h, _vp(nodes), _vp(goals), n, _vp(ok), _vp(ln)), "hyas_shot_batch")
        return ok, ln

    def flood(self, goal_xyt):
        g = np.zeros(1, GOAL_DT)
        g["x"], g["y"], g["t"] = goal_xyt
        out = np.zeros(self.twod_size(), np.float32)
        self._check(self.L.hyas_flood(self.h, _vp(g), _vp(out)), "hyas_flood")
        return out

    # -- heuristic table -------------------------------------------------------
    def table_dims(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.L.hyas_table_dims(self.h, C.byref(a), C.byref(b), C.byref(c)), "hyas_table_dims")
        return a.value, b.value, c.value

    def table_build_dev(self, curve, k0, k1, out_ptr):
        self._check(self.L.hyas_table_build_dev(self.h, int(curve), int(k0), int(k1), C.c_void_p(out_ptr)),
                    "hyas_table_build_dev")

    def table_install_dev(self, curve, raw_ptr):
        self._check(self.L.hyas_table_install_dev(self.h, int(curve), C.c_void_p(raw_ptr)), "hyas_table_install_dev")

    def table_install_noalloc_dev(self, curve, raw_ptr):
        self._check(self.L.hyas_table_install_noalloc_dev(self.h, int(curve), C.c_void_p(raw_ptr)), "hyas_table_install_noalloc_dev")

    def table_lookup(self, xyt):
        xyt = np.ascontiguousarray(xyt, np.float32).reshape(-1, 3)
        out = np.zeros(xyt.shape[0], np.float32)
        self._check(self.L.hyas_table_lookup(self.h, _vp(xyt), xyt.shape[0], _vp(out)), "hyas_table_lookup")
        return out

    # -- planning ----------------------------------------------------------------
    def plan_batch(self, starts, goals):
        starts = np.ascontiguousarray(starts, np.float32).reshape(-1, 3)
        goals = np.ascontiguousarray(goals, np.float32).reshape(-1, 3)
        n = starts.shape[0]
        if goals.shape[0] != n:
            raise HyasError(f"plan_batch: {n} starts but {goals.shape[0]} goals")
        res = np.zeros(n, RESULT_DT)
        self._check(self.L.hyas_plan_batch(self.h, _vp(starts), _vp(goals), n, _vp(res)), "hyas_plan_batch")
        return res

    def shortcut(self, paths):
        """ShortCut for a list of [len_i, 3] pose arrays; returns the kept node indices of every path."""
        offs = np.zeros(len(paths) + 1, np.uint32)
        offs[1:] = np.cumsum([len(p) for p in paths])
        allp = np.ascontiguousarray(np.concatenate(paths) if len(paths) else np.zeros((0, 3)), np.float32)
        idx = np.zeros(int(offs[-1]) + len(paths), np.uint32)
        cnt = np.zeros(len(paths), np.uint32)
        self._check(self.L.hyas_shortcut_batch(self.h, _vp(allp), _vp(offs), len(paths), _vp(idx), _vp(cnt)), "hyas_shortcut_batch")
        return [idx[int(offs[i]) + i: int(offs[i]) + i + int(cnt[i])].copy() for i in range(len(paths))]

    def path(self, i):
        n = self._check(self.L.hyas_get_path(self.h, int(i), None, 0), "hyas_get_path")
        out = np.zeros((n, 3), np.float32)
        if n:
            self.L.hyas_get_path(self.h, int(i), _vp(out), n)
        return out

    def expanded(self, i):
        n = self._check(self.L.hyas_get_expanded(self.h, int(i), None, 0), "hyas_get_expanded")
        out = np.zeros(n, np.int32)
        if n:
            self.L.hyas_get_expanded(self.h, int(i), _vp(out), n)
        return out

    def kernel_ms(self, which):
        n = C.c_int64()
        ms = self.L.hyas_last_kernel_ms(self.h, int(which), C.byref(n))
        return ms, n.value

    def reset_kernel_timers(self):
        self.L.hyas_reset_kernel_timers(self.h)

    def nodes_in_device_memory(self):
        return self.L.hyas_nodes_in_device_memory(self.h) == 1

    def set_round_timing(self, every):
        """0 (default): the search rounds carry no timing events; k > 0: every 2k-th launch of a query group is timed."""
        self._check(self.L.hyas_set_round_timing(self.h, int(every)), "hyas_set_round_timing")

    def set_stats(self, mode):
        """0 off, 1 work counters, 2 phase clocks."""
        self._check(self.L.hyas_set_stats(self.h, int(mode)), "hyas_set_stats")

    def stats(self):
        out = np.zeros(24, np.uint64)
        self._check(self.L.hyas_get_stats(self.h, _vp(out)), "hyas_get_stats")
        names = ["samples_tested", "samples_serial", "sample_list_entries", "succ_evaluated", "succ_list_entries",
                 "succ_emitted", "nodes", "shots", "clk_solve", "clk_interp", "clk_collide", "clk_to_frame", "clk_to_candidates",
                 "clk_to_winner", "clk_to_prefix", "clk_to_inputs", "clk_workgroup", "clk_pose_wave", "clk_segment_wave", "rounds"]
        return dict(zip(names, (int(v) for v in out)))
