"""CPU suite: host-side logic of the product.

The device geometry header and the list pruning are compiled for the host
(tests/host_geom, test scaffolding) and compared with the oracle on adversarial
poses: points on vertex rows and columns, within 1e-3 of cell borders,
horizontal segments through vertex rows, long segments that bypass pruning.
The planner mirror's argument handling is checked without a device."""
import ctypes as C
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

import orc

ROOT = Path(__file__).resolve().parent.parent
HG = ROOT / "tests" / "host_geom"


@pytest.fixture(scope="module")
def host_geom():
    so, src = HG / "libhost_geom.so", HG / "host_geom.cpp"
    deps = [src, ROOT / "path_planner_amd/csrc/hyas_dev.h", ROOT / "path_planner_amd/csrc/hyas_prune.h", ROOT / "include/hyas_math.h",
            ROOT / "path_planner_amd/csrc/hyas_rs.h", ROOT / "path_planner_amd/csrc/hyas_flood.h",
            ROOT / "path_planner_amd/csrc/hyas_bezier.h", ROOT / "path_planner_amd/csrc/hyas_adaptive.h",
            ROOT / "path_planner_amd/csrc/hyas_open_list.h", ROOT / "path_planner_amd/csrc/hyas_hugepool.h"]
    if not so.exists() or any(d.stat().st_mtime > so.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-D__host__=", "-D__device__=",
                        "-D__forceinline__=inline", f"-I{ROOT / 'include'}", "-o", str(so), str(src)], check=True)
    return C.CDLL(str(so))


def adversarial(W, H, n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, W, n)
    y = rng.uniform(0, H, n)
    t = rng.uniform(0, 2 * math.pi, n)
    k = n // 3
    eps = [0, 1e-4, -1e-4, 5e-4, -9e-4, 2e-3]
    y[:k] = np.round(y[:k]) + rng.choice(eps, k)  # vertex rows
    x[k:2 * k] = np.round(x[k:2 * k]) + rng.choice(eps, k)  # vertex columns
    xyt = np.stack([x, y, t], 1).astype(np.float32)
    d = rng.uniform(0.02, 0.3, n)
    a = rng.uniform(0, 2 * math.pi, n)
    pred = (xyt[:, :2] + np.stack([d * np.cos(a), d * np.sin(a)], 1)).astype(np.float32)
    pred[:k:2, 1] = xyt[:k:2, 1]  # horizontal segments on vertex rows
    m = len(xyt[k::5])
    pred[k::5] = xyt[k::5, :2] + rng.uniform(-3, 3, (m, 2)).astype(np.float32)  # long segments: unpruned lists
    return xyt, pred


@pytest.mark.parametrize("map_name", ["map_parking", "map_maze", "map_large", "map_slalom"])
def test_device_geometry_and_pruning_match_oracle(map_name, host_geom, oracle_built):
    mp = orc.load_map(map_name)
    H, W = mp.shape
    o = orc.Oracle(orc.hot_path_params(position_resolution=1))
    o.set_costmap(mp, build_table=False)
    Q = np.ascontiguousarray(o.obstacles(), np.float32)
    xyt, pred = adversarial(W, H, 60000, 7)
    want = o.collision(xyt, pred)
    want_pose = o.collision(xyt)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    ent = np.zeros(3, np.uint32)
    for prune in (1, 0):
        got = np.zeros(len(xyt), np.uint8)
        host_geom.hg_collision(vp(Q), len(Q), vp(mp), W, H, C.c_float(2), C.c_float(1), C.c_float(5), vp(xyt), vp(pred), len(xyt),
                               vp(got), prune, vp(ent))
        assert int((got != want).sum()) == 0, (map_name, prune)
        got2 = np.zeros(len(xyt), np.uint8)
        host_geom.hg_collision(vp(Q), len(Q), vp(mp), W, H, C.c_float(2), C.c_float(1), C.c_float(5), vp(xyt), None, len(xyt),
                               vp(got2), prune, vp(ent))
        assert int((got2 != want_pose).sum()) == 0, (map_name, prune)
    assert 0.02 < want.mean() < 0.98
    assert ent[1] <= ent[0] and ent[2] <= 2 * ent[0]  # pruning never adds pose entries


def test_side_by_side_reeds_shepp_candidates_equal_the_plain_form(host_geom):
    """The shot kernel evaluates the 44 Reeds-Shepp candidates in a form laid out for lanes running side by side; it
    must give the bits of the plain form (src/ReedsShepp.cpp:73-503 restated) for every candidate, valid or not."""
    rng = np.random.default_rng(77)
    n = 60000
    g = np.stack([rng.uniform(-60, 60, n), rng.uniform(-60, 60, n), rng.uniform(-2 * math.pi, 2 * math.pi, n)], 1)
    k = n // 6
    g[:k, :2] = rng.uniform(-4.5, 4.5, (k, 2))  # inside the circles where the CCC / CCCC forms exist
    g[k:2 * k, 2] = rng.choice([0.0, -0.0, math.pi, -math.pi, math.pi / 2, -math.pi / 2, 2 * math.pi], k)
    g[2 * k:3 * k, 1] = rng.choice([0.0, -0.0, 1.0, -1.0, 2.0], k)
    g[3 * k:3 * k + 8] = [[0, 0, 0], [-0.0, 0, 0], [0, -0.0, -0.0], [1e-12, 0, 0], [0, 2, math.pi], [4, 0, 0], [1e6, -1e6, 1], [2, 0, 0]]
    g = np.ascontiguousarray(g, np.float64)
    valid = C.c_long(0)
    host_geom.hg_rs_staged_mismatches.restype = C.c_long
    bad = host_geom.hg_rs_staged_mismatches(g.ctypes.data_as(C.c_void_p), n, C.byref(valid))
    assert valid.value > 4 * n  # several candidates exist for a typical goal
    assert bad == 0
    # the table kernel's walk over the side-by-side candidates picks the reference walk's path
    host_geom.hg_rs_solve_uniform_mismatches.restype = C.c_long
    assert host_geom.hg_rs_solve_uniform_mismatches(g.ctypes.data_as(C.c_void_p), n) == 0
    # the heuristic-table kernel's length-only form: candidates regrouped by symmetry, shared polar forms
    host_geom.hg_rs_shortest_length_mismatches.restype = C.c_long
    assert host_geom.hg_rs_shortest_length_mismatches(g.ctypes.data_as(C.c_void_p), n) == 0
    tgrid = np.stack(np.meshgrid(np.arange(-52, 52, 0.7) / 6.0, np.arange(-52, 52, 0.9) / 6.0, np.arange(0, 73) * np.float32(2 * math.pi / 72),
                                 indexing="ij"), -1).reshape(-1, 3)
    tgrid = np.ascontiguousarray(tgrid, np.float64)  # a coarse copy of the table's own grid (origin and axes included)
    assert host_geom.hg_rs_shortest_length_mismatches(tgrid.ctypes.data_as(C.c_void_p), len(tgrid)) == 0
    # ... and of the 180 x 60 m table at turning radius 0.35 (goals hundreds of radii away: the far field, where most
    # candidates are dropped by the signs of their arcs)
    far = np.stack(np.meshgrid(np.arange(0, 180, 1.3) / np.float32(0.35), np.arange(0, 60, 0.9) / np.float32(0.35),
                               np.arange(0, 73) * np.float32(2 * math.pi / 72), indexing="ij"), -1).reshape(-1, 3)
    far = np.ascontiguousarray(far, np.float64)
    assert host_geom.hg_rs_shortest_length_mismatches(far.ctypes.data_as(C.c_void_p), len(far)) == 0
    T = 2 * math.pi
    m = np.concatenate([rng.uniform(-60, 60, 400000), rng.uniform(-1e-9, 1e-9, 100), np.arange(-9, 10) * T, np.nextafter(np.arange(-9, 10) * T, 0),
                        np.nextafter(np.arange(-9, 10) * T, 1e9), [0.0, -0.0, math.pi, -math.pi, 1e9, -1e9, 1e300, np.inf, -np.inf, np.nan]])
    host_geom.hg_mod2pi_mismatches.restype = C.c_long
    assert host_geom.hg_mod2pi_mismatches(np.ascontiguousarray(m).ctypes.data_as(C.c_void_p), len(m)) == 0
    host_geom.hg_rs_interp_mismatches.restype = C.c_long
    assert host_geom.hg_rs_interp_mismatches(g.ctypes.data_as(C.c_void_p), 20000, 40, C.c_double(0.35)) == 0
    x = np.concatenate([rng.uniform(-100, 100, 200000), rng.uniform(-1e-3, 1e-3, 1000), [0.0, -0.0, math.pi / 4, -math.pi / 4, 1e5, -1e5]])
    host_geom.hg_sincos_mismatches.restype = C.c_long
    assert host_geom.hg_sincos_mismatches(np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), len(x)) == 0


def test_open_list_pops_in_cost_then_push_order(host_geom):
    """The search's open list (a sorted array of the best entries in front of a binary heap) must pop exactly like any
    container ordered by (f, push order): mixes of pushes, pops and look-aheads with few and with many distinct costs,
    rising costs (a search moving away from its start) and falling ones."""
    rng = np.random.default_rng(11)
    host_geom.hg_open_list_mismatches.restype = C.c_long
    for n, levels, p_pop, trend in ((3000, 3, 0.45, 0.0), (60000, 50, 0.3, 0.0), (60000, 100000, 0.48, 1e-3), (60000, 1000, 0.2, -1e-3),
                                    (500, 1, 0.5, 0.0), (40000, 7, 0.6, 0.0)):
        f = np.ascontiguousarray((rng.integers(0, levels, n) * 0.25 + trend * np.arange(n)).astype(np.float32))
        u = rng.random(n)
        op = np.ascontiguousarray(np.where(u < p_pop, 1, np.where(u < p_pop + 0.05, 2, 0)).astype(np.uint8))
        assert host_geom.hg_open_list_mismatches(f.ctypes.data_as(C.c_void_p), op.ctypes.data_as(C.c_void_p), n) == 0


def test_huge_page_pool_hands_out_sound_blocks(host_geom):
    """The pool behind the searches' host state (open lists, cell maps, arenas): sizes from one byte to beyond the largest
    pooled block; blocks distinct, aligned, intact, reused after a free, and all regions released by trim() at the end."""
    rng = np.random.default_rng(5)
    sizes = np.concatenate([np.arange(1, 70), 2 ** np.arange(6, 25), 2 ** np.arange(6, 24) + 1, 5 * 2 ** np.arange(8, 20),
                            rng.integers(1, 300000, 3000), [(16 << 20) + 1, 40 << 20]]).astype(np.uint32)
    host_geom.hg_hugepool_violations.restype = C.c_long
    assert host_geom.hg_hugepool_violations(np.ascontiguousarray(sizes).ctypes.data_as(C.c_void_p), len(sizes)) == 0


def test_flood_queue_pops_like_the_reference_priority_queue(host_geom):
    """The 2D flood's open queue restates libstdc++'s binary heap under the reference's `>=` comparator
    (include/hybrid_a_star.h:45-53): with many equal priorities it must pop the same cell at every step."""
    rng = np.random.default_rng(3)
    host_geom.hg_heap_mismatches.restype = C.c_long
    for n, levels in ((2000, 3), (50000, 40), (50000, 100000), (300, 1)):
        prio = np.ascontiguousarray(rng.integers(0, levels, n).astype(np.float32) * np.float32(0.25))
        is_pop = np.ascontiguousarray((rng.random(n) < 0.45).astype(np.uint8))
        assert host_geom.hg_heap_mismatches(prio.ctypes.data_as(C.c_void_p), is_pop.ctypes.data_as(C.c_void_p), n) == 0


@pytest.mark.parametrize("case", [("map_maze", 1.0, 0.0, 0.0), ("map_parking", 1.0, 0.0, 0.0), ("map_large", 1.0, 0.0, 0.0),
                                  ("map_basic", 0.5, -3.25, 2.5), ("map_small", 0.3, 1.7, -0.4)])
def test_flood_algorithm_matches_oracle(case, host_geom, oracle_built):
    """The flood kernel's formulation (parent index per cell, edge added at pop time from stored coordinates, restated
    queue), run serially on the host, against the oracle's BuildAStarCostMap (src/hybrid_a_star.cpp:1020-1074)."""
    name, res, ox, oy = case
    mp = orc.load_map(name)
    H, W = mp.shape
    o = orc.Oracle(orc.hot_path_params(position_resolution=1))
    o.set_costmap(mp, res, ox, oy, build_table=False)
    rng = np.random.default_rng(11)
    goals = np.stack([rng.uniform(ox, ox + W * res, 6), rng.uniform(oy, oy + H * res, 6)], 1).astype(np.float32)
    goals[0] = [ox + 0.5 * res, oy + 0.5 * res]
    goals[1] = [ox + (W - 0.01) * res, oy + (H - 0.01) * res]
    for gx, gy in goals:
        want = o.flood((gx, gy, 0.0))
        got = np.zeros_like(want)
        host_geom.hg_flood(W, H, C.c_double(res), C.c_double(ox), C.c_double(oy), C.c_float(gx), C.c_float(gy),
                           got.ctypes.data_as(C.c_void_p))
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (name, gx, gy)
        assert (want >= 0).sum() >= W * H - 1


def test_posestamped_yaw_round_trip():
    from path_planner_amd.planner import PoseStamped

    for yaw in (0.0, 0.3, math.pi / 2, math.pi, 4.0, 2 * math.pi - 1e-3):
        p = PoseStamped.from_yaw(1.5, -2.0, yaw)
        assert abs(((p.yaw() - yaw + math.pi) % (2 * math.pi)) - math.pi) < 1e-6
        assert p.x == 1.5 and p.y == -2.0 and p.frame_id == "map"


def test_costmap_accessors():
    from path_planner_amd.planner import Costmap2D

    m = np.zeros((30, 60), np.uint8)
    c = Costmap2D(m, resolution=0.5, origin_x=1.0, origin_y=-2.0)
    assert c.getSizeInCellsX() == 60 and c.getSizeInCellsY() == 30
    assert c.getResolution() == 0.5 and c.getOriginX() == 1.0 and c.getOriginY() == -2.0
    assert c.getCharMap().shape == (30, 60)


def test_make_plan_before_initialize_returns_false():
    """src/hybrid_a_star_planner.cpp:101-105: an uninitialised planner refuses, it does not throw."""
    from path_planner_amd.planner import HybridAStarPlanner, PoseStamped

    pl = HybridAStarPlanner()
    plan = ["stale"]
    assert pl.makePlan(PoseStamped.from_yaw(1, 1, 0), PoseStamped.from_yaw(2, 2, 0), plan) is False
    with pytest.raises(RuntimeError):
        pl.makePlanBatch(np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32))


def test_bezier_header_equals_oracle(host_geom, oracle_built):
    """The cubic Bezier header the kernels are built from (power-basis shot curve, Bernstein table curve), run on the
    host, against the oracle's restatement of src/cubic_bezier.cpp and include/bezier.h: same bits."""
    L = orc.Oracle(orc.hot_path_params()).L  # the portable-math build, as the kernels
    L.orc_bezier_table_cost.restype = C.c_float
    L.orc_bezier_table_cost.argtypes = [C.c_float] * 3
    L.orc_bezier_eval.restype = C.c_float
    L.orc_bezier_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    host_geom.hg_bezier_table_cost.restype = C.c_float
    host_geom.hg_bezier_table_cost.argtypes = [C.c_float] * 3
    host_geom.hg_bezier_eval.restype = C.c_float
    host_geom.hg_bezier_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    rng = np.random.default_rng(12)
    for _ in range(3000):
        x, y, th = float(rng.integers(0, 600)) / 10, float(rng.integers(0, 300)) / 10, float(rng.uniform(0, 2 * math.pi))
        a, b = L.orc_bezier_table_cost(x, y, th), host_geom.hg_bezier_table_cost(x, y, th)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (x, y, th, a, b)
    vp = lambda v: v.ctypes.data_as(C.c_void_p)
    for k in range(3000):
        s = np.array([rng.uniform(0, 60), rng.uniform(0, 30), rng.uniform(0, 2 * math.pi)], np.float32)
        g = np.array([rng.uniform(0, 60), rng.uniform(0, 30), rng.uniform(0, 2 * math.pi)], np.float32)
        if k % 50 == 0:
            g[:2] = s[:2]  # zero chord: all four control points coincide
        t = np.float32(rng.uniform(0, 1)) if k % 7 else np.float32(0)
        oa, ob = np.zeros(5, np.float32), np.zeros(4, np.float32)
        la = L.orc_bezier_eval(vp(s), vp(g), t, vp(oa))
        lb = host_geom.hg_bezier_eval(vp(s), vp(g), t, vp(ob))
        assert np.float32(la).view(np.uint32) == np.float32(lb).view(np.uint32)
        assert oa[[0, 1, 3, 4]].tobytes() == ob.tobytes(), (s, g, t, oa, ob)


def test_adaptive_header_equals_oracle(host_geom, oracle_built):
    """The adaptive step-size / steering header (angle-range algebra, ray distance, pair selection), run on the host,
    against the oracle's restatement of src/collisiondetection.cpp:688-938, :1263-1484 and src/utility.cpp: same bits."""
    vp = lambda v: v.ctypes.data_as(C.c_void_p)
    o0 = orc.Oracle(orc.hot_path_params())
    host_geom.hg_ray_distance.restype = C.c_float
    host_geom.hg_ray_distance.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(21)
    for k in range(20000):
        ox, oy = float(rng.integers(0, 20)), float(rng.integers(0, 20))
        wd, ht = 1.0, float(rng.integers(1, 6))
        px, py = (float(rng.integers(0, 25)), float(rng.integers(0, 25))) if k % 3 else (float(rng.uniform(0, 25)), float(rng.uniform(0, 25)))
        ang = float(np.float32(rng.integers(0, 361) * np.float32(math.pi / 180))) if k % 2 else float(rng.uniform(0, 6.4))
        quad = np.array([ox, oy, ox + wd, oy, ox + wd, oy + ht, ox, oy + ht], np.float32)
        a = o0.utility(29, [ox, oy, px, py, ang, wd, ht])
        b = host_geom.hg_ray_distance(vp(quad), px, py, ang)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (ox, oy, ht, px, py, ang, a, b)
    # pair selection over the oracle's own maps, every combination of the four switches on one map, defaults on two
    cases = [("map_maze", m, f) for m in (1, 2) for f in range(16)] + [("map_parking", 1, 0b1101), ("map_large", 1, 0b1101)]
    for name, mode, flags in cases:
        if mode == 2 and flags != 0b1101:
            continue
        kw = dict(add_one_more_successor=flags & 1, add_one_more_successor_only_in_free_angle_range=(flags >> 1) & 1,
                  steering_angle_towards_free_angle_range_for_obstacle_angle_range=(flags >> 2) & 1,
                  fixed_number_of_steering_angle_in_free_angle_range=(flags >> 3) & 1)
        kw["adaptive_steering_angle_and_step_size" if mode == 1 else "adaptive_step_size"] = 1
        p = orc.hot_path_params(position_resolution=1, **kw)
        o = orc.Oracle(p)
        grid = orc.load_map(name)
        o.set_costmap(grid, build_table=False)
        H, W = grid.shape
        off, rngs, dens = [0], [], np.zeros(W * H, np.float32)
        for key in range(W * H):
            c = o.adaptive_cell(key)
            if c is not None:
                raw = np.ascontiguousarray(c[2], np.float32)
                out = np.zeros((max(len(raw), 1), 3), np.float32)
                n = host_geom.hg_cut_to_circle(vp(raw), len(raw), vp(out))
                rngs.append(out[:n])
                dens[key] = c[1]
            off.append(off[-1] + (len(rngs[-1]) if c is not None else 0))
        off = np.array(off, np.uint32)
        rr = np.ascontiguousarray(np.concatenate(rngs), np.float32)
        n = 4000
        nodes = np.stack([rng.uniform(0.5, W - 0.5, n), rng.uniform(0.5, H - 0.5, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
        goals = np.stack([rng.uniform(0.5, W - 0.5, n), rng.uniform(0.5, H - 0.5, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
        starts = np.stack([rng.uniform(0.5, W - 0.5, n), rng.uniform(0.5, H - 0.5, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
        goals[:200, :2] = nodes[:200, :2] + rng.uniform(-0.5, 0.5, (200, 2)).astype(np.float32)  # goal within one step
        nodes[200:400, :2] = np.round(nodes[200:400, :2])  # nodes on cell corners
        dsg = np.sqrt(((starts[:, 0] - goals[:, 0]) ** 2 + (starts[:, 1] - goals[:, 1]) ** 2).astype(np.float32)).astype(np.float32)
        d = np.float32(o0.utility(17, [p.steering_angle]))  # ConvertDegToRad(steering_angle)
        steer = np.array([0] + [np.float32(s) * (np.float32(i) * d) for i in range(1, (p.number_of_successors - 1) // 2 + 1) for s in (1, -1)], np.float32)
        MAXP = 31  # HYAS_AD_MAX_PAIRS: forward pairs a node can have (32 successor slots, one is the reverse pair's)
        pairs = np.zeros((n, MAXP, 2), np.float32)
        counts = np.zeros(n, np.int32)
        checked = C.c_int(0)
        host_geom.hg_step_steer_pairs(W, H, C.c_double(1.0), C.c_double(0.0), C.c_double(0.0), vp(off), vp(rr), vp(dens), C.c_float(p.obstacle_detection_range),
                                      C.c_float(p.constant_density), mode, flags, vp(steer), len(steer), vp(nodes), vp(goals), vp(dsg), n, vp(pairs), vp(counts),
                                      C.byref(checked))
        # the lane-parallel form of the selection was compared with the serial one wherever it applies
        assert checked.value > n // 2 or mode != 1 or not (flags & 8), (name, mode, flags, checked.value)
        most = 0
        for i in range(n):
            want, nw = o.step_steer_pairs(nodes[i], goals[i], starts[i])
            want, nw = want[:nw - 1], nw - 1  # the reverse pair comes from the fixed rule
            if nw > MAXP:
                assert counts[i] == -1
                continue
            assert counts[i] == nw, (name, mode, flags, i, counts[i], nw)
            assert pairs[i, :nw].tobytes() == want.tobytes(), (name, mode, flags, i, pairs[i, :nw], want)
            most = max(most, nw)
        assert most >= 3
