"""GPU parity: every C-ABI entry point of libhyas.so against the CPU oracle on
the same seeded inputs, bit for bit (verdicts, indices, f32/f64 values)."""
import math

import numpy as np
import pytest

import orc
from path_planner_amd import Hyas, Params
from path_planner_amd import _lib as hl

pytestmark = pytest.mark.gpu


def _params(**kw):
    o = orc.hot_path_params(**kw)
    h = Params()
    for name, _ in Params._fields_:
        setattr(h, name, getattr(o, name))
    return o, h


class Pair:
    def __init__(self, map_name, res=1.0, ox=0.0, oy=0.0, **kw):
        po, ph = _params(**kw)
        self.map = orc.load_map(map_name)
        self.o = orc.Oracle(po)
        self.o.set_costmap(self.map, res, ox, oy, build_table=True, nthreads=16)
        self.h = Hyas(ph)
        self.h.set_costmap(self.map, res, ox, oy, build_table=True)
        self.H, self.W = self.map.shape
        self.res, self.ox, self.oy = res, ox, oy

    def poses(self, n, seed, margin=0.0):
        rng = np.random.default_rng(seed)
        x = rng.uniform(self.ox - margin, self.ox + self.W * self.res + margin, n)
        y = rng.uniform(self.oy - margin, self.oy + self.H * self.res + margin, n)
        t = rng.uniform(0, 2 * math.pi, n)
        return np.stack([x, y, t], 1).astype(np.float32)

    def free_poses(self, n, seed):
        out = []
        k = 0
        while len(out) < n:
            p = self.poses(4 * n, seed + k)
            ok = self.o.collision(p)
            out.extend(p[ok == 1])
            k += 1
        return np.array(out[:n], np.float32)


@pytest.fixture(scope="module")
def parking(oracle_built):
    return Pair("map_parking", position_resolution=2)


@pytest.fixture(scope="module")
def maze(oracle_built):
    return Pair("map_maze", position_resolution=2, max_iterations=150)


def test_obstacles_and_lists(parking):
    p = parking
    a, b = p.o.obstacles(), p.h.obstacles()
    assert a.shape == b.shape and a.shape[0] > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for key in range(p.W * p.H):
        assert max(0, p.o.in_range_count(key)) == p.h.in_range_count(key), key


def _random_costmap(w, h, seed, share=0.12):
    rng = np.random.default_rng(seed)
    m = np.zeros((h, w), np.uint8)
    hit = rng.random((h, w)) < share
    m[hit] = rng.choice(np.array([1, 60, 128, 252, 253, 254, 255], np.uint8), int(hit.sum()))
    return m


def test_in_range_list_contents(parking, maze):
    """Every grid key's obstacle list, polygon by polygon in list order (the lists are assembled on the GPU: lengths,
    prefix sum, ids), on the shipped maps and on a random costmap of every cost class with half-metre cells, a shifted
    origin, more than 4096 keys and several hundred polygons."""
    for p in (parking, maze):
        for key in range(p.W * p.H):
            a, b = p.o.in_range_list(key), p.h.in_range_list(key)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
    m = _random_costmap(96, 70, 5)
    po, ph = _params(position_resolution=1, obstacle_detection_range=2.5)
    o, h = orc.Oracle(po), Hyas(ph)
    o.set_costmap(m, 0.5, -7.25, 3.5, build_table=False)
    h.set_costmap(m, 0.5, -7.25, 3.5, build_table=False)
    a, b = o.obstacles(), h.obstacles()
    assert a.shape[0] > 300 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    longest = 0
    for key in range(96 * 70):
        a, b = o.in_range_list(key), h.in_range_list(key)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
        longest = max(longest, a.shape[0])
    assert longest > 8
    rng = np.random.default_rng(6)
    xyt = np.stack([rng.uniform(-7.25, 40.75, 20000), rng.uniform(3.5, 38.5, 20000), rng.uniform(0, 6.28, 20000)], 1).astype(np.float32)
    assert np.array_equal(o.collision(xyt), h.collision(xyt))


def test_in_range_lists_from_tile_bins_on_a_large_map(monkeypatch):
    """The lists are picked from bins of polygons per 16 x 16 tile of keys instead of testing every polygon against every
    key.  On a random 800 x 600 costmap (about 14 000 polygons, where the oracle's own assembly would take minutes) they
    must equal the lists of a build whose bins hold every polygon, which is the exhaustive assembly the other list tests
    check against the oracle: every key's length, every fifth key's polygons in order, and the verdicts of poses."""
    rng = np.random.default_rng(2)
    m = np.zeros((600, 800), np.uint8)
    m[rng.random((600, 800)) < 0.03] = 254
    _, ph = _params(position_resolution=1)
    binned = Hyas(ph)
    binned.set_costmap(m, build_table=False)
    monkeypatch.setenv("HYAS_ALL_POLYGONS_IN_BINS", "1")
    exhaustive = Hyas(ph)
    exhaustive.set_costmap(m, build_table=False)
    monkeypatch.delenv("HYAS_ALL_POLYGONS_IN_BINS")
    assert binned.obstacles().shape[0] > 10000
    a = np.array([binned.in_range_count(k) for k in range(m.size)])
    b = np.array([exhaustive.in_range_count(k) for k in range(m.size)])
    assert np.array_equal(a, b) and a.max() > 8
    for key in range(0, m.size, 5):
        la, lb = binned.in_range_list(key), exhaustive.in_range_list(key)
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), key
    xyt = np.stack([rng.uniform(1, 799, 200000), rng.uniform(1, 599, 200000), rng.uniform(0, 6.28, 200000)], 1).astype(np.float32)
    assert np.array_equal(binned.collision(xyt), exhaustive.collision(xyt))


def test_pruned_lists_from_the_gpu_equal_the_host_form(monkeypatch):
    """The pose and segment lists (entries of the in-range lists that can still change a verdict) are cut on the GPU.
    With HYAS_CHECK_PRUNE the library also runs the host form of the same rules (hyas_prune.h, the form the CPU tests
    compare with the oracle's verdicts) and fails hyas_set_costmap when offsets or entries differ."""
    monkeypatch.setenv("HYAS_CHECK_PRUNE", "1")
    rng = np.random.default_rng(2)
    big = np.zeros((600, 800), np.uint8)
    big[rng.random((600, 800)) < 0.03] = 254
    cases = [(orc.load_map("map_maze"), 1.0, 0.0, 0.0, {}), (orc.load_map("map_parking"), 1.0, 0.0, 0.0, dict(map_boundary_obstacle=1)),
             (orc.load_map("map_large"), 1.0, 0.0, 0.0, dict(vehicle_length=4.5, vehicle_width=2.0)),
             (_random_costmap(96, 70, 5), 0.5, -7.25, 3.5, dict(obstacle_detection_range=2.5)), (big, 1.0, 0.0, 0.0, {}),
             (_random_costmap(64, 48, 9), 2.0, 10.0, -30.0, {})]
    for m, res, ox, oy, kw in cases:
        _, ph = _params(position_resolution=1, **kw)
        h = Hyas(ph)
        h.set_costmap(m, res, ox, oy, build_table=False)  # raises when the lists differ
        h.close()


def test_map_boundary_obstacle(oracle_built):
    """map_boundary_obstacle = 1 (CollisionDetection::AddMapBoundaryAsObstacle, src/collisiondetection.cpp:1223-1240):
    four rectangles around the grid, in cells whatever resolution and origin are, join the obstacle polygons behind the
    extracted ones; polygons, lists, verdicts near the border and whole searches against the oracle."""
    p = Pair("map_parking", position_resolution=2, max_iterations=200, map_boundary_obstacle=1)
    q = Pair("map_parking", position_resolution=2, max_iterations=200)
    a, b = p.o.obstacles(), p.h.obstacles()
    assert a.shape[0] == q.h.obstacles().shape[0] + 4 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(b[-4], np.array([0, -1, p.W, -1, p.W, 0, 0, 0], np.float32))
    for key in range(p.W * p.H):
        a, b = p.o.in_range_list(key), p.h.in_range_list(key)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
    xyt = p.poses(40000, 31, margin=0.5)
    rng = np.random.default_rng(32)
    xyt[:10000, 0] = rng.uniform(0, 1.2, 10000)  # along the left border
    xyt[10000:20000, 1] = rng.uniform(p.H - 1.2, p.H, 10000)  # along the top border
    pred = (xyt[:, :2] + rng.uniform(-0.3, 0.3, (xyt.shape[0], 2))).astype(np.float32)
    want = p.o.collision(xyt, pred)
    assert np.array_equal(want, p.h.collision(xyt, pred))
    assert (want != q.o.collision(xyt, pred)).sum() > 20  # the boundary rectangles decide verdicts of their own
    s, g = p.free_poses(24, 33), p.free_poses(24, 34)
    _plan_compare(p, s, g)
    # half-metre cells and a shifted origin: the rectangles stay where the reference puts them
    r = Pair("map_basic", res=0.5, ox=-3.25, oy=2.5, position_resolution=1, max_iterations=60, map_boundary_obstacle=1,
             vehicle_width=0.5, vehicle_length=1.0, min_turning_radius=2.0, cd_min_turning_radius=2.0)
    assert np.array_equal(r.o.obstacles().view(np.uint32), r.h.obstacles().view(np.uint32))
    xyt = r.poses(30000, 35, margin=0.7)
    pred = (xyt[:, :2] + np.float32(0.07)).astype(np.float32)
    assert np.array_equal(r.o.collision(xyt, pred), r.h.collision(xyt, pred))
    _plan_compare(r, r.free_poses(12, 36), r.free_poses(12, 37))


def test_footprint_table(parking):
    a, b = parking.o.footprint_table(), parking.h.footprint_table()
    assert a.shape == b.shape
    assert np.array_equal(a, b)


@pytest.mark.parametrize("with_pred", [False, True])
def test_collision_polygon(parking, maze, with_pred):
    for p, seed in ((parking, 11), (maze, 12)):
        xyt = p.poses(40000, seed, margin=1.5)
        pred = None
        if with_pred:
            rng = np.random.default_rng(seed + 100)
            pred = (xyt[:, :2] + rng.uniform(-0.3, 0.3, (xyt.shape[0], 2))).astype(np.float32)
        a = p.o.collision(xyt, pred)
        b = p.h.collision(xyt, pred)
        assert 0.05 < a.mean() < 0.95
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]


def test_collision_polygon_edge_cases(parking):
    p = parking
    # empty batch, poses on cell borders / map border / just outside
    assert p.h.collision(np.zeros((0, 3), np.float32)).shape == (0,)
    xs = np.array([0.0, 1e-7, 0.9999999, 1.0, 59.9995, 59.9999, 60.0, -1e-6, 30.0], np.float32)
    ys = np.array([0.0, 29.9999, 30.0, 15.0, 15.5, -0.0], np.float32)
    ts = np.array([0.0, 1e-5, math.pi / 2, math.pi, 6.2831, 3.0], np.float32)
    g = np.array(np.meshgrid(xs, ys, ts, indexing="ij")).reshape(3, -1).T.astype(np.float32)
    assert np.array_equal(p.o.collision(g), p.h.collision(g))
    pred = g[:, :2] + np.float32(0.1)
    assert np.array_equal(p.o.collision(g, pred), p.h.collision(g, pred))


def test_collision_footprint(parking):
    p = parking
    xyt = p.poses(50000, 21, margin=1.0)
    xyt[:, 0] = np.abs(xyt[:, 0])
    xyt[:, 1] = np.abs(xyt[:, 1])
    a = p.o.collision(xyt, mode=1)
    b = p.h.collision(xyt, mode=hl.MODE_FOOTPRINT)
    assert 0.05 < a.mean() < 0.99
    assert np.array_equal(a, b)
    # ragged tail (n not a multiple of 4)
    for n in (1, 2, 3, 5, 7):
        assert np.array_equal(p.o.collision(xyt[:n], mode=1), p.h.collision(xyt[:n], mode=1))
    # device buffers at any alignment: poses 4 bytes and verdicts 1 byte off a 16-byte boundary (hyas_collision_batch_dev)
    import torch

    n = 4099
    buf = torch.zeros(3 * n + 8, dtype=torch.float32, device="cuda")
    outb = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
    for off_in, off_out in ((0, 0), (1, 1), (3, 2), (2, 0), (0, 3)):
        src = buf[off_in:off_in + 3 * n]
        src.copy_(torch.from_numpy(xyt[:n].reshape(-1)))
        dst = outb[off_out:off_out + n]
        dst.zero_()
        p.h.collision_dev(src.data_ptr(), 0, n, dst.data_ptr(), mode=hl.MODE_FOOTPRINT)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), a[:n]), (off_in, off_out)


def test_table(parking):
    p = parking
    assert p.o.table_dims() == p.h.table_dims()
    rng = np.random.default_rng(5)
    q = np.stack([rng.uniform(0, p.W, 100000), rng.uniform(0, p.H, 100000), rng.uniform(-7, 7, 100000)], 1).astype(np.float32)
    a, b = p.o.table_lookup(q), p.h.table_lookup(q)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # exact grid points hit every key once
    nx, ny, ns = p.o.table_dims()
    gx, gy = np.meshgrid(np.arange(nx) * 0.5, np.arange(ny) * 0.5, indexing="ij")
    for t in (0.0, 1.0, 3.1, 6.2):
        g = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, t)], 1).astype(np.float32)
        assert np.array_equal(p.o.table_lookup(g).view(np.uint32), p.h.table_lookup(g).view(np.uint32))


def test_flood(parking, maze, monkeypatch):
    """2D cost-to-goal flood (src/hybrid_a_star.cpp:1020-1074) from the GPU kernel against the oracle: goals anywhere
    on the map incl. its corners; with the kernel's open queue held in LDS and with almost all of it pushed out into
    the spill area."""
    for heap, in_hbm in ((None, None), ("8", None), (None, "16"), ("8", "3")):
        if heap:
            monkeypatch.setenv("HYAS_FLOOD_HEAP", heap)
        if in_hbm:  # cell state in the workspace instead of LDS, that many queries per compute unit
            monkeypatch.setenv("HYAS_FLOOD_CELLS_IN_HBM", in_hbm)
        for p, seed in ((parking, 31), (maze, 32)):
            goals = list(p.free_poses(5, seed)) + [np.array([0.01, 0.01, 0], np.float32),
                                                   np.array([p.W - 0.01, p.H - 0.01, 1], np.float32),
                                                   np.array([p.W * 0.5, 0.0, 2], np.float32)]
            for g in goals[: (3 if heap else len(goals))]:
                a, b = p.o.flood(g), p.h.flood(g)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
                assert (a >= 0).sum() >= p.W * p.H - 1
            # a goal off the map floods nothing
            a, b = p.o.flood((-5.0, 3.0, 0.0)), p.h.flood((-5.0, 3.0, 0.0))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        monkeypatch.delenv("HYAS_FLOOD_HEAP", raising=False)
        monkeypatch.delenv("HYAS_FLOOD_CELLS_IN_HBM", raising=False)


def test_flood_large_and_shifted_maps(oracle_built, monkeypatch):
    """The flood on the largest in-tree map and on a map with 0.5 m cells and a shifted origin (cell coordinates that
    are not whole numbers), with the cell state in LDS and in the workspace; and a batch large enough for the
    workspace form to be chosen by itself (sixteen floods per compute unit side by side): every query's start-cell cost
    enters its start node's h, which the result records of a one-iteration search carry."""
    for name, res, ox, oy in (("map_large", 1.0, 0.0, 0.0), ("map_basic", 0.5, -3.25, 2.5)):
        p = Pair(name, res=res, ox=ox, oy=oy, position_resolution=1)
        rng = np.random.default_rng(5)
        for in_hbm in (None, "16"):
            if in_hbm:
                monkeypatch.setenv("HYAS_FLOOD_CELLS_IN_HBM", in_hbm)
            for _ in range(4):
                g = (ox + rng.uniform(0, p.W * res), oy + rng.uniform(0, p.H * res), 0.0)
                a, b = p.o.flood(g), p.h.flood(g)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, g, in_hbm)
            monkeypatch.delenv("HYAS_FLOOD_CELLS_IN_HBM", raising=False)
    p = Pair("map_large", position_resolution=1, max_iterations=40)
    xyt, goals = p.free_poses(1200, 61), p.free_poses(1200, 62)
    res = p.h.plan_batch(xyt, goals)
    for i in range(0, 1200, 9):
        r, path, exp = p.o.plan(xyt[i], goals[i])
        assert r.tobytes() == res[i].tobytes(), (i, r, res[i])
        assert np.array_equal(exp, p.h.expanded(i)) and path.tobytes() == p.h.path(i).tobytes(), i


def _nodes(p, n, seed):
    xyt = p.free_poses(n, seed)
    rng = np.random.default_rng(seed + 1)
    nodes = np.zeros(n, orc.NODE_DT)
    nodes["x"], nodes["y"], nodes["t"] = xyt[:, 0], xyt[:, 1], xyt[:, 2]
    nodes["g"] = (999 + rng.uniform(0, 50, n)).astype(np.float32)
    nodes["prim"] = rng.integers(0, 6, n)
    gxyt = p.free_poses(n, seed + 2)
    goals = np.zeros(n, orc.GOAL_DT)
    goals["x"], goals["y"], goals["t"] = gxyt[:, 0], gxyt[:, 1], gxyt[:, 2]
    # a few goals closer than one step to their node (step clipping, src/hybrid_a_star.cpp:607-614)
    k = n // 10
    goals["x"][:k] = nodes["x"][:k] + rng.uniform(-0.1, 0.1, k).astype(np.float32)
    goals["y"][:k] = nodes["y"][:k] + rng.uniform(-0.1, 0.1, k).astype(np.float32)
    goals["x"][0], goals["y"][0] = nodes["x"][0], nodes["y"][0]  # zero distance: every step skipped
    return nodes, goals


def test_expand(parking, maze):
    for p, seed in ((parking, 41), (maze, 42)):
        n = 3000
        nodes, goals = _nodes(p, n, seed)
        twod = np.stack([p.o.flood((g["x"], g["y"], g["t"])) for g in goals[:64]])
        a, ca = p.o.expand(nodes[:64], goals[:64], twod)
        b, cb = p.h.expand(nodes[:64], goals[:64], twod)
        assert np.array_equal(ca, cb)
        assert a.tobytes() == b.tobytes()
        a, ca = p.o.expand(nodes, goals)
        b, cb = p.h.expand(nodes, goals)
        assert np.array_equal(ca, cb)
        assert ca[0] == 0 and ca.max() == p.o.num_successors()
        assert a.tobytes() == b.tobytes()


def test_shot(parking, maze):
    for p, seed in ((parking, 51), (maze, 52)):
        nodes, goals = _nodes(p, 1500, seed)
        a, la = p.o.shot(nodes, goals)
        b, lb = p.h.shot(nodes, goals)
        assert 0.02 < a.mean() < 0.98
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))


def test_shot_long_curves(oracle_built):
    """Shots over tens of metres on the 180 x 60 map: curves of several hundred to well over a thousand samples, which
    the kernel tests in passes that each cover the whole curve, accepted and rejected ones, for all three curve types
    (verdict and length bit for bit)."""
    for curve, kw in ((1, {}), (0, {}), (2, {})):
        p = Pair("map_large", position_resolution=1, max_iterations=10, curve_type=curve, **kw)
        nodes, goals = _nodes(p, 600, 81 + curve)
        a, la = p.o.shot(nodes, goals)
        b, lb = p.h.shot(nodes, goals)
        long_ones = la > 25.6 * 2  # more than two passes of 256 samples 0.1 m apart (Dubins: 0.2 m)
        assert long_ones.sum() > 100 and a[long_ones].sum() > 5 and (a[long_ones] == 0).sum() > 50
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))


def _plan_compare(p, starts, goals):
    res = p.h.plan_batch(starts, goals)
    for i in range(len(starts)):
        r, path, exp = p.o.plan(starts[i], goals[i])
        assert r.tobytes() == res[i].tobytes(), (i, r, res[i])
        assert np.array_equal(exp, p.h.expanded(i)), i
        assert path.tobytes() == p.h.path(i).tobytes(), i
    return res


def test_plan_parking_reference_query(parking):
    # the reference's fixed start/goal, src/planner.cpp:193-207
    p = parking
    res = _plan_compare(p, np.array([[2, p.H - 2, 0]], np.float32), np.array([[p.W - 2, 2, 0]], np.float32))
    assert res[0]["status"] == hl.STATUS_SHOT


def test_plan_batch(parking, maze):
    for p, seed, n in ((parking, 61, 48), (maze, 62, 48)):
        s, g = p.free_poses(n, seed), p.free_poses(n, seed + 1)
        # include rejected poses (occupied start, goal off the map)
        s[0] = [p.W + 5, 1, 0]
        g[1] = [-3, 2, 1]
        res = _plan_compare(p, s, g)
        assert res[0]["status"] == hl.STATUS_BAD_POSE and res[1]["status"] == hl.STATUS_BAD_POSE
        assert len(set(res["status"].tolist())) >= 2


def test_resolution_and_origin(oracle_built):
    # 0.5 m cells with a shifted origin: float-accumulated cell axes, non-integer keys
    p = Pair("map_basic", res=0.5, ox=-3.25, oy=2.5, position_resolution=1, max_iterations=60,
             vehicle_width=0.5, vehicle_length=1.0, min_turning_radius=2.0, cd_min_turning_radius=2.0)
    a, b = p.o.obstacles(), p.h.obstacles()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    xyt = p.poses(30000, 71, margin=0.7)
    pred = (xyt[:, :2] + np.float32(0.07)).astype(np.float32)
    assert np.array_equal(p.o.collision(xyt, pred), p.h.collision(xyt, pred))
    s, g = p.free_poses(16, 72), p.free_poses(16, 73)
    _plan_compare(p, s, g)


def test_plan_golden_fixture():
    """The committed end-to-end vectors (tests/golden/plan_golden.npz, made by the CPU oracle): status, counters,
    ordered expanded-node indices and path poses from the GPU path, bit for bit."""
    from pathlib import Path

    g = np.load(Path(__file__).resolve().parent / "golden" / "plan_golden.npz")
    by_map = {}
    for name in g["names"]:
        by_map.setdefault((str(g[f"{name}/map"]), int(g[f"{name}/pr"][0])), []).append(str(name))
    for (mp, pr), names in by_map.items():
        _, ph = _params(position_resolution=pr)
        h = Hyas(ph)
        h.set_costmap(orc.load_map(mp), build_table=True)
        s = np.stack([g[f"{n}/start"] for n in names]).astype(np.float32)
        t = np.stack([g[f"{n}/goal"] for n in names]).astype(np.float32)
        res = h.plan_batch(s, t)
        for i, n in enumerate(names):
            assert res[i].tobytes() == g[f"{n}/result"].tobytes(), (n, res[i], g[f"{n}/result"])
            assert np.array_equal(h.expanded(i), g[f"{n}/expanded"]), n
            assert h.path(i).tobytes() == g[f"{n}/path"].tobytes(), n
            # path cost within 1e-6 relative is implied by the bitwise match; stated for the record
            assert abs(float(res[i]["g_final"]) - float(g[f"{n}/result"]["g_final"])) <= 1e-6 * max(1.0, float(res[i]["g_final"]))


@pytest.mark.parametrize("fp", [(0.18, 0.22), (1.2, 1.7), (2.0, 4.5)], ids=["in_tree", "mid", "car"])
def test_footprint_grid_only(fp, oracle_built):
    """hyas_set_grid + footprint-list test on a random grid whose size is not a multiple of the 8x8 bit tiles:
    short lists (packed table in LDS), lists of 5..7 cells, and long lists (general table walk)."""
    W, H = 1501, 1099
    grid = np.where(np.random.default_rng(1).random((H, W)) < 0.10, 254, 0).astype(np.uint8)
    grid[np.random.default_rng(2).random((H, W)) < 0.02] = 250  # the threshold value itself
    grid[np.random.default_rng(3).random((H, W)) < 0.02] = 249
    po, ph = _params(fp_width=fp[0], fp_length=fp[1])
    po.position_resolution = ph.position_resolution = 10
    o = orc.Oracle(po)
    o.set_grid(grid)
    h = Hyas(ph)
    h.set_grid(grid)
    ta, tb = o.footprint_table(), h.footprint_table()
    assert np.array_equal(ta, tb)
    rng = np.random.default_rng(4)
    n = 400003
    xyt = np.stack([rng.uniform(-3, W + 3, n), rng.uniform(-3, H + 3, n), rng.uniform(-0.5, 6.6, n)], 1).astype(np.float32)
    xyt[:1000, 0] = np.round(xyt[:1000, 0])  # cell borders
    xyt[1000:2000, 1] = np.round(xyt[1000:2000, 1])
    # values the int conversions cannot represent: off the grid for both sides
    xyt[2000:2008, 0] = [np.nan, 1e10, -1e10, 3e9, -3e9, np.inf, -np.inf, 2147483648.0]
    xyt[2008:2016, 1] = [np.nan, 1e10, -1e10, 3e9, -3e9, np.inf, -np.inf, 2147483648.0]
    xyt[2016:2020, 2] = [np.nan, 1e10, -1e10, np.inf]
    a = o.collision(xyt, mode=1)
    b = h.collision(xyt, mode=hl.MODE_FOOTPRINT)
    assert np.array_equal(a, b), int((a != b).sum())
    assert 0.05 < a.mean() < 0.95
    assert not a[2000:2020].any()
    if fp == (0.18, 0.22):
        assert int(ta[:, 0].max()) == 4


# ----------------------------------------------------------------- Dubins (curve_type 0)
@pytest.fixture(scope="module")
def parking_dubins(oracle_built):
    # turning radius small enough that shots connect inside the 60x30 map
    return Pair("map_parking", position_resolution=2, curve_type=0, min_turning_radius=2.0, cd_min_turning_radius=2.0,
                max_iterations=400)


@pytest.fixture(scope="module")
def maze_dubins_default(oracle_built):
    # the reference's struct defaults: Dubins, three forward successors, no reverse, radius 6
    po = orc.default_params(position_resolution=2, max_iterations=200)
    ph = Params()
    for name, _ in Params._fields_:
        setattr(ph, name, getattr(po, name))
    p = Pair.__new__(Pair)
    p.map = orc.load_map("map_maze")
    p.o = orc.Oracle(po)
    p.o.set_costmap(p.map, build_table=True, nthreads=16)
    p.h = Hyas(ph)
    p.h.set_costmap(p.map, build_table=True)
    p.H, p.W = p.map.shape
    p.res, p.ox, p.oy = 1.0, 0.0, 0.0
    return p


def test_dubins_table(parking_dubins):
    p = parking_dubins
    nx, ny, ns = p.h.table_dims()
    assert (nx, ny, ns) == p.o.table_dims()
    rng = np.random.default_rng(5)
    q = np.stack([rng.uniform(0, p.W, 100000), rng.uniform(0, p.H, 100000), rng.uniform(-7, 7, 100000)], 1).astype(np.float32)
    assert np.array_equal(p.o.table_lookup(q).view(np.uint32), p.h.table_lookup(q).view(np.uint32))
    gx, gy = np.meshgrid(np.arange(nx) * 0.5, np.arange(ny) * 0.5, indexing="ij")
    for t in (0.0, 1.0, 3.1, 6.2):
        g = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, t)], 1).astype(np.float32)
        a, b = p.o.table_lookup(g), p.h.table_lookup(g)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert (a < 1e6).mean() > 0.9


def test_dubins_shot_and_expand(parking_dubins, maze_dubins_default):
    for p, seed in ((parking_dubins, 81), (maze_dubins_default, 82)):
        nodes, goals = _nodes(p, 1500, seed)
        a, la = p.o.shot(nodes, goals)
        b, lb = p.h.shot(nodes, goals)
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
        sa, ca = p.o.expand(nodes, goals)
        sb, cb = p.h.expand(nodes, goals)
        assert np.array_equal(ca, cb)
        assert sa.tobytes() == sb.tobytes()
    assert parking_dubins.o.shot(*_nodes(parking_dubins, 1500, 81))[0].mean() > 0.02


def test_dubins_plan(parking_dubins, maze_dubins_default):
    for p, seed, n in ((parking_dubins, 91, 32), (maze_dubins_default, 92, 32)):
        s, g = p.free_poses(n, seed), p.free_poses(n, seed + 1)
        res = _plan_compare(p, s, g)
        assert len(res) == n


# ----------------------------------------------------------------- cubic Bezier (curve_type 2)
@pytest.fixture(scope="module")
def parking_bezier(oracle_built):
    # curvature limit 1 / 2: shots connect inside the 60x30 map
    return Pair("map_parking", position_resolution=2, curve_type=2, min_turning_radius=2.0, cd_min_turning_radius=2.0,
                max_iterations=400)


@pytest.fixture(scope="module")
def maze_bezier(oracle_built):
    return Pair("map_maze", position_resolution=2, curve_type=2, min_turning_radius=6.0, cd_min_turning_radius=6.0,
                max_iterations=300)


def test_bezier_table(parking_bezier):
    p = parking_bezier
    nx, ny, ns = p.h.table_dims()
    assert (nx, ny, ns) == p.o.table_dims()
    rng = np.random.default_rng(6)
    q = np.stack([rng.uniform(0, p.W, 100000), rng.uniform(0, p.H, 100000), rng.uniform(-7, 7, 100000)], 1).astype(np.float32)
    a, b = p.o.table_lookup(q), p.h.table_lookup(q)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    gx, gy = np.meshgrid(np.arange(nx) * 0.5, np.arange(ny) * 0.5, indexing="ij")
    for t in (0.0, 1.0, 3.1, 6.2):
        g = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, t)], 1).astype(np.float32)
        a, b = p.o.table_lookup(g), p.h.table_lookup(g)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        hit = a < 1e7  # the origin key is never filled (src/lookup_table.cpp:243-247)
        assert hit.mean() > 0.99
        # a polyline inscribed in the curve is at least as long as the chord
        chord = np.hypot(g[:, 0], g[:, 1])
        assert (a[hit] >= chord[hit] * (1 - 1e-5)).all()


def test_bezier_shot_and_expand(parking_bezier, maze_bezier):
    for p, seed in ((parking_bezier, 83), (maze_bezier, 84)):
        nodes, goals = _nodes(p, 1500, seed)
        a, la = p.o.shot(nodes, goals)
        b, lb = p.h.shot(nodes, goals)
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
        sa, ca = p.o.expand(nodes, goals)
        sb, cb = p.h.expand(nodes, goals)
        assert np.array_equal(ca, cb)
        assert sa.tobytes() == sb.tobytes()
    assert parking_bezier.o.shot(*_nodes(parking_bezier, 1500, 83))[0].mean() > 0.02


def test_bezier_plan(parking_bezier, maze_bezier):
    for p, seed, n in ((parking_bezier, 93, 32), (maze_bezier, 94, 32)):
        s, g = p.free_poses(n, seed), p.free_poses(n, seed + 1)
        res = _plan_compare(p, s, g)
        assert len(res) == n


def test_abi_error_behaviour():
    """Call-order and argument errors come back as codes with a message; nothing crashes, nothing is computed."""
    import ctypes as C

    L = hl.load_library()
    # unsupported parameter sets are refused at creation
    for kw in (dict(adaptive_step_size=1, consider_steering_angle_range_for_obstacle_density=1),
               dict(adaptive_steering_angle_and_step_size=1, obstacle_detection_range=0.0),
               dict(curve_type=3), dict(curve_type=-1),
               dict(vehicle_width=-1.0), dict(min_turning_radius=0.0), dict(number_of_successors=0), dict(headings=0)):
        p = hl.hot_path_params(**kw)
        assert L.hyas_create(C.byref(p), 0) is None, kw
    p = hl.hot_path_params()
    assert L.hyas_create(C.byref(p), 99) is None  # no such device
    h = Hyas(hl.hot_path_params(position_resolution=1))
    buf = np.zeros(64, np.float32)
    out = np.zeros(64, np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    # before a costmap: state error
    assert L.hyas_collision_batch(h.h, 0, vp(buf), None, 4, vp(out)) == hl.ESTATE
    assert L.hyas_plan_batch(h.h, vp(buf), vp(buf), 1, vp(out)) == hl.ESTATE
    assert L.hyas_twod_size(h.h) == hl.ESTATE
    assert L.hyas_set_costmap(h.h, None, 4, 4, 1.0, 0.0, 0.0, 0) == hl.EINVAL
    assert L.hyas_set_costmap(h.h, vp(out), 0, 4, 1.0, 0.0, 0.0, 0) == hl.EINVAL
    assert b"" != L.hyas_last_error(h.h)
    # costmap without table: searching needs the table
    h.set_costmap(orc.load_map("map_small"), build_table=False)
    assert L.hyas_plan_batch(h.h, vp(buf), vp(buf), 1, vp(out)) == hl.ESTATE
    assert L.hyas_collision_batch(h.h, 7, vp(buf), None, 4, vp(out)) == hl.EINVAL  # unknown mode
    assert L.hyas_collision_batch(h.h, 0, None, None, 4, vp(out)) == hl.EINVAL
    assert L.hyas_collision_batch(h.h, 0, vp(buf), None, 0, vp(out)) == hl.OK  # empty batch
    # grid-only contexts refuse the polygon test
    g = Hyas(hl.default_params())
    g.set_grid(np.zeros((16, 16), np.uint8))
    assert L.hyas_collision_batch(g.h, 0, vp(buf), None, 4, vp(out)) == hl.ESTATE
    assert L.hyas_collision_batch(g.h, 1, vp(buf), None, 4, vp(out)) == hl.OK
    # a context that had a costmap and then got a bare grid: nothing of the costmap is left to read or launch on
    g2 = Hyas(hl.hot_path_params(position_resolution=1))
    g2.set_costmap(orc.load_map("map_small"), build_table=True)
    assert g2.in_range_count(0) >= 0
    g2.set_grid(np.zeros((16, 16), np.uint8))
    a, b, c_ = C.c_int32(), C.c_int32(), C.c_int32()
    assert L.hyas_table_dims(g2.h, C.byref(a), C.byref(b), C.byref(c_)) == hl.ESTATE
    assert L.hyas_table_build_dev(g2.h, 1, 0, 1, vp(buf)) == hl.ESTATE
    assert L.hyas_table_install_dev(g2.h, 1, vp(buf)) == hl.ESTATE
    assert L.hyas_in_range_count(g2.h, 0) == -1
    assert L.hyas_get_obstacles(g2.h, None, 0) == hl.ESTATE
    assert L.hyas_flood(g2.h, vp(buf), vp(buf)) == hl.ESTATE
    assert L.hyas_collision_batch(g2.h, 1, vp(buf), None, 4, vp(out)) == hl.OK
    # install with an unknown curve type
    assert L.hyas_table_install_dev(h.h, 5, vp(buf)) == hl.EINVAL
    # an empty plan batch is fine once everything is set up
    h.set_costmap(orc.load_map("map_small"), build_table=True)
    res = h.plan_batch(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(res) == 0
    with pytest.raises(hl.HyasError):
        h.plan_batch(np.zeros((2, 3), np.float32), np.zeros((3, 3), np.float32))
