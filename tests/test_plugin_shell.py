"""The C++ nav_core plugin shell (plugin/): builds with plain g++ against the C ABI; on a GPU it answers makePlan
with the poses of the committed golden plans; without one it stays uninitialised and refuses."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import orc

ROOT = Path(__file__).resolve().parent.parent
PLUGIN = ROOT / "plugin"


@pytest.fixture(scope="module")
def demo():
    from path_planner_amd import _lib as hl

    if not hl.lib_path().exists():
        import __graft_entry__ as g

        g.build()
    subprocess.run(["make", "-s", "-C", str(PLUGIN)], check=True)
    return PLUGIN / "plugin_demo"


def _run(demo, tmp_path, map_name, pr, queries):
    mp = orc.load_map(map_name)
    f = tmp_path / f"{map_name}.bin"
    f.write_bytes(np.ascontiguousarray(mp, np.uint8).tobytes())
    args = [str(demo), str(f), str(mp.shape[1]), str(mp.shape[0]), "1.0", "0", "0", str(pr)]
    for s, g in queries:
        args += [repr(float(v)) for v in s] + [repr(float(v)) for v in g]
    return subprocess.run(args, capture_output=True, text=True, timeout=300)


def test_plugin_builds_and_exports_the_reference_class(demo):
    out = subprocess.run(["nm", "-DC", str(PLUGIN / "libhybrid_a_star_planner_lib.so")], capture_output=True, text=True, check=True).stdout
    for sig in ("HybridAStar::HybridAStarPlanner::HybridAStarPlanner()",
                "HybridAStar::HybridAStarPlanner::initialize(std::__cxx11::basic_string<char, std::char_traits<char>, std::allocator<char> >, costmap_2d::Costmap2DROS*)",
                "HybridAStar::HybridAStarPlanner::makePlan(geometry_msgs::PoseStamped const&, geometry_msgs::PoseStamped const&",
                "HybridAStar::HybridAStarPlanner::makePlanService(nav_msgs::GetPlan::Request&, nav_msgs::GetPlan::Response&)",
                "HybridAStar::HybridAStarPlanner::publishPlan("):
        assert sig in out, sig
    xml = (PLUGIN / "hybrid_a_star_planner_plugin.xml").read_text()
    assert 'name="HybridAStar/HybridAStarPlanner"' in xml and 'base_class_type="nav_core::BaseGlobalPlanner"' in xml
    assert "libhybrid_a_star_planner_lib" in xml


def test_plugin_refuses_without_gpu(demo, tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run(demo, tmp_path, "map_parking", 2, [((2, 28, 0), (58, 2, 0))])
    assert r.returncode == 1  # makePlan returned false
    assert r.stdout.split()[0] == "-1"
    assert "uninitialised" in r.stderr or "initialized" in r.stderr


@pytest.mark.gpu
def test_plugin_make_plan_matches_golden(demo, tmp_path):
    g = np.load(ROOT / "tests" / "golden" / "plan_golden.npz")
    for map_name, names in (("map_parking", ["parking_fixed"]), ("map_maze", ["maze_00", "maze_01"])):
        # yaws of these queries lie in [0, pi]: they survive the quaternion round trip of the ROS message bit for bit
        r = _run(demo, tmp_path, map_name, 2, [(g[f"{n}/start"], g[f"{n}/goal"]) for n in names])
        lines = r.stdout.strip().splitlines()
        k = 0
        for n in names:
            want = g[f"{n}/result"]
            status, explored, expanded, npts = (int(v) for v in lines[k].split())
            k += 1
            path = np.array([[float(v) for v in ln.split()] for ln in lines[k:k + npts]], np.float64).reshape(npts, 3)
            k += npts
            if int(want["path_len"]) == 0:  # no path: makePlan returns false, empty plan
                assert status == -1 and npts == 0
                continue
            assert status == int(want["status"]) and explored == int(want["nodes_explored"]) and expanded == int(want["nodes_expanded"])
            gp = g[f"{n}/path"].astype(np.float64)
            assert npts == len(gp)
            assert np.array_equal(path[:, :2].astype(np.float32), g[f"{n}/path"][:, :2])  # %.9g round-trips a float32
            d = np.abs(((path[:, 2] - gp[:, 2] + np.pi) % (2 * np.pi)) - np.pi)
            assert d.max() < 1e-6
        assert k == len(lines)
        assert r.returncode == (1 if any(int(g[f"{n}/result"]["path_len"]) == 0 for n in names) else 0)
