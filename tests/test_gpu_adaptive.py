"""GPU parity of the adaptive step-size / steering modes (HybridAStar::CreateSuccessor, src/hybrid_a_star.cpp:547-599,
with CollisionDetection::FindStepSizeAndSteeringAngle / FindStepSize and the maps built per costmap,
src/collisiondetection.cpp:387-500, :688-938, :1013-1151, :1263-1484) against the CPU oracle, bit for bit."""
import math

import numpy as np
import pytest

import orc
from path_planner_amd import _lib as hl
from test_gpu_parity import Pair, _nodes, _plan_compare

pytestmark = pytest.mark.gpu

COMBINED = dict(adaptive_steering_angle_and_step_size=1)
STEP_ONLY = dict(adaptive_step_size=1)


@pytest.fixture(scope="module")
def maze_combined(oracle_built):
    return Pair("map_maze", position_resolution=2, max_iterations=400, **COMBINED)


@pytest.fixture(scope="module")
def parking_step(oracle_built):
    return Pair("map_parking", position_resolution=2, max_iterations=400, **STEP_ONLY)


@pytest.mark.parametrize("name,res,ox,oy", [("map_maze", 1.0, 0.0, 0.0), ("map_parking", 1.0, 0.0, 0.0), ("map_large", 1.0, 0.0, 0.0),
                                            ("map_basic", 0.5, -3.25, 2.5)])
def test_adaptive_maps(oracle_built, name, res, ox, oy):
    """Minimum obstacle distance, normalised density and the angle ranges of every grid key."""
    p = Pair(name, res=res, ox=ox, oy=oy, position_resolution=1, **COMBINED)
    seen = 0
    for key in range(p.W * p.H):
        want = p.o.adaptive_cell(key, cut=True)
        md, de, rng = p.h.adaptive_cell(key)
        if want is None:
            assert len(rng) == 0 and de == 0
            continue
        seen += 1
        assert np.float32(want[0]).view(np.uint32) == np.float32(md).view(np.uint32), key
        assert np.float32(want[1]).view(np.uint32) == np.float32(de).view(np.uint32), key
        assert want[2].tobytes() == rng.tobytes(), (key, want[2], rng)
    assert seen >= p.W * p.H - 1


@pytest.mark.parametrize("keep", ["0", "2", "21"])
def test_adaptive_maps_whatever_the_passes_keep(oracle_built, monkeypatch, keep):
    """The map build counts the ranges of every key, then fills them in; the counting pass keeps each key's first few
    ranges so that the filling pass need not sweep again (HYAS_ADAPTIVE_KEEP, default 8).  None kept, fewer than most
    cells have, and more than any cell has: the maps are the oracle's."""
    monkeypatch.setenv("HYAS_ADAPTIVE_KEEP", keep)
    p = Pair("map_maze", position_resolution=1, **COMBINED)
    most = 0
    for key in range(p.W * p.H):
        want = p.o.adaptive_cell(key, cut=True)
        md, de, rng = p.h.adaptive_cell(key)
        if want is None:
            assert len(rng) == 0 and de == 0
            continue
        most = max(most, len(rng))
        assert np.float32(want[0]).view(np.uint32) == np.float32(md).view(np.uint32), key
        assert want[2].tobytes() == rng.tobytes(), (key, want[2], rng)
    assert most > 8  # the default keeps eight: the maze has cells beyond that, so test_adaptive_maps covers both ways


def test_adaptive_modes_with_map_boundary_obstacle(oracle_built):
    """The boundary rectangles of map_boundary_obstacle = 1 take part in the sweeps and the density like any obstacle."""
    p = Pair("map_parking", position_resolution=1, max_iterations=300, map_boundary_obstacle=1, **COMBINED)
    for key in range(p.W * p.H):
        want = p.o.adaptive_cell(key, cut=True)
        md, de, rng = p.h.adaptive_cell(key)
        if want is None:
            assert len(rng) == 0 and de == 0
            continue
        assert np.float32(want[0]).view(np.uint32) == np.float32(md).view(np.uint32), key
        assert np.float32(want[1]).view(np.uint32) == np.float32(de).view(np.uint32), key
        assert want[2].tobytes() == rng.tobytes(), (key, want[2], rng)
    _plan_compare(p, p.free_poses(16, 41), p.free_poses(16, 42))


def _starts(p, n, seed):
    s = np.zeros(n, orc.GOAL_DT)
    xyt = p.free_poses(n, seed)
    s["x"], s["y"], s["t"] = xyt[:, 0], xyt[:, 1], xyt[:, 2]
    return s


@pytest.mark.parametrize("flags", [0b1101, 0b0000, 0b0111, 0b1010, 0b0101])
def test_expand_combined_mode(oracle_built, flags):
    """Successors with adaptive steering angle and step size; flags: add one more successor, only in free ranges,
    steer towards the free range, fixed number of steering angles in a free range (0b1101: the reference's defaults)."""
    kw = dict(add_one_more_successor=flags & 1, add_one_more_successor_only_in_free_angle_range=(flags >> 1) & 1,
              steering_angle_towards_free_angle_range_for_obstacle_angle_range=(flags >> 2) & 1,
              fixed_number_of_steering_angle_in_free_angle_range=(flags >> 3) & 1)
    p = Pair("map_maze", position_resolution=1, **COMBINED, **kw)
    assert p.h.num_successors() == 32 == p.o.num_successors()
    nodes, goals = _nodes(p, 3000, 40 + flags)
    starts = _starts(p, 3000, 90 + flags)
    # a node holds up to 31 forward {step, steering} pairs and the reverse step (two passes of sixteen candidates on the
    # GPU): the reference's default switches give at most 7, the 5-degree fans of fixed_number... = 0 about 20
    n_pairs = np.array([p.o.step_steer_pairs((nd["x"], nd["y"], nd["t"]), (g["x"], g["y"], g["t"]), (s["x"], s["y"], s["t"]))[1]
                        for nd, g, s in zip(nodes, goals, starts)])
    assert n_pairs.max() <= 32
    if flags == 0:
        assert (n_pairs > 16).sum() > 50  # nodes that need the second pass are in the sample
    if not flags & 0b1000:
        assert (n_pairs > 14).sum() > 5  # ... and in the other fan configurations nodes beyond the former limit
    some = np.concatenate([np.flatnonzero(n_pairs > 16)[:24], np.arange(48)])
    twod = np.stack([p.o.flood((g["x"], g["y"], g["t"])) for g in goals[some]])
    a, ca = p.o.expand(nodes[some], goals[some], twod, starts[some])
    b, cb = p.h.expand(nodes[some], goals[some], twod, starts[some])
    assert np.array_equal(ca, cb) and a.tobytes() == b.tobytes()
    a, ca = p.o.expand(nodes, goals, None, starts)
    b, cb = p.h.expand(nodes, goals, None, starts)
    assert not (cb == 0xFF).any()  # nothing flagged
    assert np.array_equal(ca, cb)
    assert a.tobytes() == b.tobytes()
    assert ca.max() >= 3 and (ca == 0).sum() < len(ca) // 2
    # the plain entry point has no start pose to weigh with
    with pytest.raises(hl.HyasError):
        p.h.expand(nodes[:4], goals[:4])


def test_expand_step_only_mode(parking_step, oracle_built):
    for p, seed in ((parking_step, 51), (Pair("map_maze", position_resolution=1, reverse=0, number_of_successors=7, **STEP_ONLY), 52)):
        nodes, goals = _nodes(p, 3000, seed)
        a, ca = p.o.expand(nodes, goals)
        b, cb = p.h.expand(nodes, goals)
        assert np.array_equal(ca, cb)
        assert a.tobytes() == b.tobytes()
        step = np.hypot(a["x"][:, 0] - nodes["x"], a["y"][:, 0] - nodes["y"])[ca > 0]
        assert step.max() > 1.0  # steps longer than the fixed 0.2


def test_plan_adaptive_modes(maze_combined, parking_step, oracle_built):
    """Whole searches: result record, expanded nodes in order and path, both modes, three maps."""
    large = Pair("map_large", position_resolution=1, max_iterations=300, **COMBINED)
    for p, seed, n in ((maze_combined, 71, 48), (parking_step, 72, 32), (large, 73, 32)):
        s, g = p.free_poses(n, seed), p.free_poses(n, seed + 1)
        res = _plan_compare(p, s, g)
        assert len(set(res["status"].tolist())) >= 2


def test_plan_with_steering_fans(oracle_built):
    """Whole searches with fixed_number_of_steering_angle_in_free_angle_range = 0 (a free angle range is covered by a fan of
    steering angles 5 degrees apart, src/collisiondetection.cpp:836-900): nodes with more than sixteen candidates take two
    passes on the GPU; nothing is flagged, and records, expansion order and paths equal the oracle's."""
    for mp, n, kw in (("map_maze", 32, {}), ("map_parking", 24, dict(add_one_more_successor=0))):
        p = Pair(mp, position_resolution=1, max_iterations=400, fixed_number_of_steering_angle_in_free_angle_range=0, **COMBINED, **kw)
        s, g = p.free_poses(n, 81), p.free_poses(n, 82)
        res = _plan_compare(p, s, g)
        assert not (res["status"] == hl.STATUS_OVERFLOW).any()
        assert res["nodes_expanded"].sum() > (1000 if mp == "map_maze" else 50)


def test_reference_defaults_plan(oracle_built):
    """The reference's struct defaults (adaptive_step_size = true, Dubins, three successors, short_cut) end to end."""
    po = orc.default_params(position_resolution=2, max_iterations=300)
    assert po.adaptive_step_size == 1 and hl.default_params().adaptive_step_size == 1
    p = Pair.__new__(Pair)
    ph = hl.Params()
    for name, _ in hl.Params._fields_:
        setattr(ph, name, getattr(po, name))
    p.map = orc.load_map("map_maze")
    p.o = orc.Oracle(po)
    p.o.set_costmap(p.map, build_table=True, nthreads=16)
    p.h = hl.Hyas(ph)
    p.h.set_costmap(p.map, build_table=True)
    p.H, p.W = p.map.shape
    p.res, p.ox, p.oy = 1.0, 0.0, 0.0
    s, g = p.free_poses(32, 75), p.free_poses(32, 76)
    _plan_compare(p, s, g)


@pytest.mark.parametrize("every_point", [1, 0])
def test_piecewise_bezier_interpolation(oracle_built, every_point):
    """HybridAStar::ConvertToPiecewiseCubicBezierPath (src/hybrid_a_star.cpp:830-883): the traced part of a path that an
    analytic shot closed is replaced by cubic Bezier curves between nodes at least one cell apart; with the adaptive step
    size (steps of several cells) and with the fixed 0.2 step (no two anchors: the traced part drops out)."""
    found = 0
    for kw in (dict(adaptive_step_size=1), dict(adaptive_steering_angle_and_step_size=1, curve_type=2, min_turning_radius=2.0,
                                                cd_min_turning_radius=2.0), dict()):
        p = Pair("map_maze", position_resolution=1, max_iterations=400, piecewise_cubic_bezier_interpolation=1,
                 analytical_expansion_every_point=every_point, short_cut=every_point, **kw)
        s, g = p.free_poses(40, 77), p.free_poses(40, 78)
        res = _plan_compare(p, s, g)
        shot = res["status"] == hl.STATUS_SHOT
        found += int(shot.sum())
        for i in np.flatnonzero(shot)[:8]:
            path = p.h.path(i)
            assert np.allclose(path[-1], g[i]) and np.isfinite(path).all()
    assert found >= 10


@pytest.mark.parametrize("name,iters,detection_range", [("map_maze", 1500, 1.0), ("map_parking", 1500, 1.0), ("map_large", 600, 1.0),
                                                        ("map_maze", 1500, 6.0), ("map_parking", 1500, 6.0), ("map_maze", 1500, 3.0)])
def test_shipped_parameter_file_plan(oracle_built, name, iters, detection_range):
    """The configuration the reference ships (param/parameters.yaml): cubic Bezier curves, adaptive steering angle and
    step size, 0.25 x 0.5 m vehicle with a 0.35 m turning radius, shots at the paper's schedule, short cut and piecewise
    Bezier interpolation of the result. Iterations are capped so that the scalar oracle finishes in seconds. On these
    1 m maps the shipped obstacle_detection_range of 1 m leaves most nodes with one straight successor (a sweep that
    meets no obstacle is stored with the cell's minimum distance, src/collisiondetection.cpp:480-487, and is then not
    recognised as free, :764): the searches end with an empty open list after a few nodes, here as there. The same
    file with the struct's range of 6 m, and with 3 m, searches properly."""
    over = {"max_iterations": iters, "obstacle_detection_range": detection_range}
    po = orc.default_params(**{**hl.SHIPPED_YAML, **over})
    ph = hl.shipped_yaml_params(**over)
    for f, _ in hl.Params._fields_:
        assert getattr(po, f) == getattr(ph, f), f
    p = Pair.__new__(Pair)
    p.map = orc.load_map(name)
    p.o = orc.Oracle(po)
    p.o.set_costmap(p.map, build_table=True, nthreads=16)
    p.h = hl.Hyas(ph)
    p.h.set_costmap(p.map, build_table=True)
    p.H, p.W = p.map.shape
    p.res, p.ox, p.oy = 1.0, 0.0, 0.0
    s, g = p.free_poses(32, 81), p.free_poses(32, 82)
    res = _plan_compare(p, s, g)
    if detection_range > 1.0 and name == "map_maze":
        assert (res["status"] == hl.STATUS_SHOT).sum() >= 4
