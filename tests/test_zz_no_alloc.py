"""hyas_table_install_noalloc_dev (per-key scratch in a static device array, no allocation) against
hyas_table_install_dev on the same raw tables, through the C ABI: two calls in a row on one GPU, the second on a
smaller table, each compared bit for bit through lookups on every grid point of every heading sample and at random
poses; a table beyond the static array's size is refused."""
import ctypes as C
import math

import numpy as np
import pytest

from path_planner_amd import Hyas, hot_path_params
from path_planner_amd import _lib as hl
from path_planner_amd.sharding import load_map

pytestmark = pytest.mark.gpu


def _probe_points(h, pr, W, H):
    nx, ny, ns = h.table_dims()
    gx, gy = np.meshgrid(np.arange(nx, dtype=np.float32) / np.float32(pr), np.arange(ny, dtype=np.float32) / np.float32(pr), indexing="ij")
    pts = [np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, np.float32(k * 2 * math.pi / 72 + 1e-3))], 1) for k in range(0, 72, 5)]
    rng = np.random.default_rng(17)
    pts.append(np.stack([rng.uniform(0, W, 200000), rng.uniform(0, H, 200000), rng.uniform(-7, 7, 200000)], 1))
    return np.concatenate(pts).astype(np.float32)


def _tables_equal(map_name, pr, curve, **kw):
    import torch

    mp = load_map(map_name)
    H, W = mp.shape
    p = hot_path_params(position_resolution=pr, curve_type=curve, **kw)
    a, b = Hyas(p), Hyas(p)
    a.set_costmap(mp, build_table=False)
    b.set_costmap(mp, build_table=False)
    nx, ny, ns = a.table_dims()
    raw = torch.zeros((ns, ny, nx), dtype=torch.float32, device="cuda")
    a.table_build_dev(curve, 0, ns, raw.data_ptr())
    torch.cuda.synchronize()
    a.table_install_dev(curve, raw.data_ptr())
    b.table_install_noalloc_dev(curve, raw.data_ptr())
    q = _probe_points(a, pr, W, H)
    ta, tb = a.table_lookup(q), b.table_lookup(q)
    assert (ta < 1e6).mean() > 0.5
    assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32)), (map_name, pr)
    # a search on the table installed without allocation gives the answers of the other context
    s = np.array([[2.0, H - 2.0, 0.0]], np.float32)
    g = np.array([[W - 2.0, 2.0, 0.0]], np.float32)
    assert a.plan_batch(s, g).tobytes() == b.plan_batch(s, g).tobytes()


def test_noalloc_install_matches_the_allocating_one_twice_in_a_row():
    _tables_equal("map_parking", 2, 1, cd_min_turning_radius=2.0)  # 525 720 keys
    _tables_equal("map_small", 1, 1, cd_min_turning_radius=0.35)   # a smaller table next: stale ranks must not leak
    _tables_equal("map_parking", 2, 0, cd_min_turning_radius=2.0)  # Dubins


def test_noalloc_install_refuses_large_tables():
    import torch

    p = hot_path_params(position_resolution=10)
    h = Hyas(p)
    h.set_costmap(load_map("map_maze"), build_table=False)  # 18 M keys
    buf = torch.zeros(16, dtype=torch.float32, device="cuda")
    rc = h.L.hyas_table_install_noalloc_dev(h.h, 1, C.c_void_p(buf.data_ptr()))
    assert rc == hl.ERANGE
    assert b"static scratch" in h.L.hyas_last_error(h.h)
    assert h.L.hyas_table_install_noalloc_dev(h.h, 7, C.c_void_p(buf.data_ptr())) == hl.EINVAL
