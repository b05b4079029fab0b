"""GPU tests at the benchmark configurations' full sizes, through properties that do not need the (slow) CPU
oracle on every element: determinism, independence of batch composition, paths that start and end where asked
and are collision-free, sharded table builds that tile the single build, verdict counts against the oracle where
the oracle is fast enough to run in full."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

import orc
from path_planner_amd import Hyas, Params, default_params, hot_path_params
from path_planner_amd import _lib as hl
from path_planner_amd.sharding import heading_slices, make_queries, result_digest

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def _compare_threaded(map_name, params_kw, S, G, res, got, which):
    """Queries `which` against the CPU oracle, one oracle context per thread: result record, ordered expanded cells and
    path poses must be bit-identical.  Returns the queries that differ."""
    import os
    import threading

    T = max(1, min(16, len(os.sched_getaffinity(0))))
    mp = orc.load_map(map_name)
    which = list(which)
    bad = []

    def work(t):
        o = orc.Oracle(orc.hot_path_params(**params_kw))
        o.set_costmap(mp, build_table=True, nthreads=2)
        for i in which[t::T]:
            r, path, exp = o.plan(S[i], G[i])
            if r.tobytes() != res[i].tobytes() or not np.array_equal(exp, got[i][1]) or path.tobytes() != got[i][0].tobytes():
                bad.append(i)

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    return sorted(bad)


@pytest.fixture(scope="module")
def maze_ctx():
    h = Hyas(hot_path_params())
    h.set_costmap(orc.load_map("map_maze"), build_table=True)
    return h


def test_maze_1024_queries_properties(maze_ctx, oracle_built):
    h = maze_ctx
    S, G = make_queries(h.collision, 1024, 20250228, 50, 50)
    res = h.plan_batch(S, G)
    paths = [h.path(i) for i in range(1024)]
    expanded = [h.expanded(i) for i in range(1024)]
    # every status occurs on this batch; counters are consistent
    assert set(res["status"].tolist()) == {hl.STATUS_NO_PATH, hl.STATUS_SHOT, hl.STATUS_MAX_ITER}
    assert (res["nodes_expanded"] <= res["nodes_explored"]).all()
    assert (res["nodes_explored"] <= 10001 + 500).all()  # pops of already closed nodes count too
    assert (res["successors"] <= 6 * res["nodes_expanded"]).all()
    for i in range(1024):
        assert len(expanded[i]) == res[i]["nodes_expanded"]
        assert len(paths[i]) == res[i]["path_len"]
    # paths: start at the start, a connected one ends on the goal, every pose is traversable
    done = np.flatnonzero(res["status"] == hl.STATUS_SHOT)
    assert len(done) > 500
    for i in done:
        p = paths[i]
        assert np.allclose(p[0], S[i], atol=1e-2)
        assert np.array_equal(p[-1, :2], G[i, :2])
        step = np.hypot(np.diff(p[:, 0]), np.diff(p[:, 1]))
        assert step.max() <= 0.2 + 1e-4
    allp = np.concatenate([paths[i][:-1] for i in done])  # the appended goal pose was pre-checked separately
    assert h.collision(allp).all()
    # determinism
    res2 = h.plan_batch(S, G)
    assert res.tobytes() == res2.tobytes()
    assert all(np.array_equal(paths[i], h.path(i)) for i in range(0, 1024, 37))
    # independence of batch composition and order: a shuffled subset gives the same per-query answers
    rng = np.random.default_rng(1)
    pick = rng.permutation(1024)[:96]
    res3 = h.plan_batch(S[pick], G[pick])
    for k, i in enumerate(pick):
        assert res3[k].tobytes() == res[i].tobytes(), i
        assert np.array_equal(h.expanded(k), expanded[i]) and np.array_equal(h.path(k), paths[i])
    # every one of the 1024 benchmark queries against the oracle (about a minute on sixteen cores)
    bad = _compare_threaded("map_maze", {}, S, G, res, list(zip(paths, expanded)), range(1024))
    assert not bad, bad
    # ... which makes this batch's digest a verified answer: bench.py prints the digest of what its timed steps
    # returned and compares it with the same file
    want = json.loads((GOLDEN / "bench_digest.json").read_text())["queries_maze"]
    assert result_digest(res) == want


def test_footprint_10m_checks_against_oracle(oracle_built):
    W = H = 4096
    n = 10_000_000
    grid = np.where(np.random.default_rng(1).random((H, W)) < 0.10, 254, 0).astype(np.uint8)
    rng = np.random.default_rng(2)
    xyt = np.stack([rng.uniform(8, W - 8, n), rng.uniform(8, H - 8, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
    h = Hyas(default_params())
    h.set_grid(grid)
    got = h.collision(xyt, mode=hl.MODE_FOOTPRINT)
    o = orc.Oracle(orc.default_params())
    o.set_grid(grid)
    want = o.collision(xyt, mode=1)
    assert int(got.sum()) == int(want.sum())
    assert np.array_equal(got, want)
    # the free fraction is what the occupancy predicts: 0.9 ** (mean listed cells) within sampling noise
    assert 0.80 < got.mean() < 0.90


def test_rs_table_slices_tile_the_full_build(oracle_built):
    """180x60 map, rho 0.35: building the 73 heading slices in 8 runs (as 8 GPUs would) equals one build; three
    slices are checked against the CPU oracle entry by entry; lookups hit what was installed."""
    import torch

    grid = np.zeros((60, 180), np.uint8)
    for pr in (1, 3):
        p = hot_path_params(position_resolution=pr, cd_min_turning_radius=0.35)
        h = Hyas(p)
        h.set_costmap(grid, build_table=False)
        nx, ny, ns = h.table_dims()
        po = orc.hot_path_params(position_resolution=pr, cd_min_turning_radius=0.35)
        o = orc.Oracle(po)
        o.set_costmap(grid, build_table=False)
        assert (nx, ny, ns) == o.table_dims()  # float-accumulated axes: 1/3 steps overshoot by one sample
        assert ns == 73 and nx >= 180 * pr and ny >= 60 * pr
        full = torch.zeros((ns, ny, nx), dtype=torch.float32, device="cuda")
        h.table_build_dev(1, 0, ns, full.data_ptr())
        parts = torch.zeros_like(full)
        slices, per = heading_slices(ns, 8)
        for k0, k1 in slices:
            if k1 > k0:
                h.table_build_dev(1, k0, k1, parts[k0:k1].data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(full.view(torch.int32), parts.view(torch.int32))
        for k in (0, 36, 72):
            want = o.table_build(1, k, k + 1)[0]
            assert np.array_equal(full[k].cpu().numpy().view(np.uint32), want.view(np.uint32)), (pr, k)
        h.table_install_dev(1, parts.data_ptr())
        q = np.array([[3.0, 2.0, 1.0], [100.0, 40.0, 3.0], [179.0, 59.0, 6.0]], np.float32)
        o.build_table(1, 16)
        assert np.array_equal(h.table_lookup(q).view(np.uint32), o.table_lookup(q).view(np.uint32))


def test_large_map_query_blocks(oracle_built):
    """map_large, benchmark sampling (seed 3): planning a batch in two contiguous blocks (two ranks) gives the same
    answers as planning it whole; spot-check against the oracle."""
    mp = orc.load_map("map_large")
    H, W = mp.shape
    h = Hyas(hot_path_params(max_iterations=400))
    h.set_costmap(mp, build_table=True)
    S, G = make_queries(h.collision, 4096, 3, W, H)
    whole = h.plan_batch(S, G)
    keep = {i: (h.path(i), h.expanded(i)) for i in range(0, 4096, 129)}
    a = h.plan_batch(S[:2048], G[:2048])
    first = {i: (h.path(i), h.expanded(i)) for i in keep if i < 2048}
    b = h.plan_batch(S[2048:], G[2048:])
    second = {i: (h.path(i - 2048), h.expanded(i - 2048)) for i in keep if i >= 2048}
    assert np.concatenate([a, b]).tobytes() == whole.tobytes()
    for i, (p, e) in {**first, **second}.items():
        assert np.array_equal(p, keep[i][0]) and np.array_equal(e, keep[i][1])
    o = orc.Oracle(orc.hot_path_params(max_iterations=400))
    o.set_costmap(mp, build_table=True, nthreads=16)
    for i in list(keep)[:12]:
        r, path, exp = o.plan(S[i], G[i])
        assert r.tobytes() == whole[i].tobytes(), i
        assert np.array_equal(exp, keep[i][1]) and path.tobytes() == keep[i][0].tobytes()


def test_search_schedule_does_not_change_answers(maze_ctx, monkeypatch):
    """How the host schedules the search (query groups in flight, worker threads, nodes expanded ahead of time per
    query, nodes sent ahead of their existence, launches per round, chunk size, successors dropped on the GPU or on the host) must not show in any output: result records, expansion order and paths are bit-identical."""
    h = maze_ctx
    S, G = make_queries(h.collision, 300, 77, 50, 50)
    base = h.plan_batch(S, G)
    default_path = h.nodes_in_device_memory()
    keep = [(h.path(i), h.expanded(i)) for i in range(300)]
    assert base["nodes_expanded"].sum() > 100000
    for env in (dict(HYAS_SPEC="1", HYAS_GROUPS="1", HYAS_THREADS="1"), dict(HYAS_SPEC="2", HYAS_GROUPS="3", HYAS_THREADS="5"),
                dict(HYAS_SPEC="8", HYAS_GROUPS="8", HYAS_THREADS="16"), dict(HYAS_SPEC="4", HYAS_CHUNK="97"),
                dict(HYAS_SPEC="8", HYAS_SPEC_ADAPT="0", HYAS_GROUPS="16", HYAS_DRIVERS="2"),
                dict(HYAS_SPEC="5", HYAS_SPEC_FLOOR="0", HYAS_DRIVERS="8"),
                dict(HYAS_DEPTH="1"), dict(HYAS_DEPTH="2"), dict(HYAS_DEPTH="3", HYAS_SPEC="3"), dict(HYAS_DEPTH="4", HYAS_SPEC="1", HYAS_GROUPS="2"),
                dict(HYAS_DEPTH="2", HYAS_CHUNK="97", HYAS_NODES_IN_HOST_MEMORY="1"),
                dict(HYAS_AHEAD="0"), dict(HYAS_AHEAD="1", HYAS_SPEC="1"), dict(HYAS_AHEAD="4", HYAS_SPEC="8"), dict(HYAS_AHEAD="3", HYAS_DEPTH="2"),
                dict(HYAS_AHEAD="1", HYAS_AHEAD_WIDTH="4"), dict(HYAS_PIN="0"), dict(HYAS_LIVE_WIDTHS="0"),
                dict(HYAS_GROUPS="1"), dict(HYAS_GROUPS="7", HYAS_CHUNK="61"),
                dict(HYAS_TREE="42", HYAS_SPEC="10", HYAS_SPEC_ADAPT="0"), dict(HYAS_TREE="7", HYAS_DEPTH="2"), dict(HYAS_TREE="48", HYAS_SPEC="1"), dict(HYAS_AHEAD="2", HYAS_AHEAD_WIDTH="2", HYAS_SPEC="8"),
                dict(HYAS_NODES_IN_HOST_MEMORY="1"), dict(HYAS_FLOOD_HEAP="16", HYAS_CHUNK="64"), dict(HYAS_FLOOD_CELLS_IN_HBM="16"), dict(HYAS_FLOOD_CELLS_IN_HBM="5", HYAS_FLOOD_HEAP="16", HYAS_FLOOD_BESIDE="4"),
                dict(HYAS_FILTER="0"), dict(HYAS_FILTER="0", HYAS_DEPTH="2"), dict(HYAS_FILTER="1", HYAS_NODES_IN_HOST_MEMORY="1", HYAS_GROUPS="3"),
                dict(HYAS_FILTER="1", HYAS_CHUNK="50", HYAS_TREE="20"), dict()):
        for k in ("HYAS_SPEC", "HYAS_GROUPS", "HYAS_THREADS", "HYAS_CHUNK", "HYAS_SPEC_ADAPT", "HYAS_SPEC_FLOOR", "HYAS_DRIVERS",
                  "HYAS_NODES_IN_HOST_MEMORY", "HYAS_FLOOD_HEAP", "HYAS_FLOOD_CELLS_IN_HBM", "HYAS_FLOOD_BESIDE", "HYAS_DEPTH", "HYAS_AHEAD", "HYAS_AHEAD_WIDTH", "HYAS_PIN", "HYAS_TREE", "HYAS_LIVE_WIDTHS",
                  "HYAS_FILTER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = h.plan_batch(S, G)
        # which way the popped nodes travelled: pinned host memory when asked for; otherwise device memory wherever
        # the GPU offers a host mapping (the last, empty case runs after the switch and must be back on that path)
        if "HYAS_NODES_IN_HOST_MEMORY" in env:
            assert not h.nodes_in_device_memory()
        elif env == {}:
            assert h.nodes_in_device_memory() == default_path
        assert res.tobytes() == base.tobytes(), env
        for i in range(0, 300, 7):
            assert np.array_equal(h.path(i), keep[i][0]) and np.array_equal(h.expanded(i), keep[i][1]), (env, i)


def test_two_contexts_plan_at_once_and_one_is_destroyed_meanwhile(maze_ctx):
    """Two contexts of one process planning at the same time from two host threads (one GPU each in a real process; both
    on this GPU here) share nothing but the pool behind the searches' host memory: both give the answers of a call on its
    own, also when a third context is created, used and destroyed while they are planning (its teardown asks the pool to
    release its regions, which it must refuse while blocks are in use)."""
    import threading

    mp = orc.load_map("map_maze")
    S, G = make_queries(maze_ctx.collision, 256, 99, 50, 50)
    want = maze_ctx.plan_batch(S, G)
    want_exp = [maze_ctx.expanded(i) for i in range(0, 256, 5)]
    ctxs = []
    for _ in range(2):
        h = Hyas(hot_path_params())
        h.set_costmap(mp, build_table=True)
        ctxs.append(h)
    got, errors = [None, None], []

    def plan(t):
        try:
            for _ in range(3):
                got[t] = ctxs[t].plan_batch(S, G)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=plan, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for _ in range(3):
        h3 = Hyas(hot_path_params())
        h3.set_costmap(mp, build_table=True)
        r3 = h3.plan_batch(S[:8], G[:8])
        assert r3.tobytes() == want[:8].tobytes()
        h3.close()
    for x in th:
        x.join()
    assert not errors, errors
    for t in range(2):
        assert got[t].tobytes() == want.tobytes()
        for k, i in enumerate(range(0, 256, 5)):
            assert np.array_equal(ctxs[t].expanded(i), want_exp[k])
        ctxs[t].close()


def test_round_timing_switch(maze_ctx):
    """hyas_set_round_timing: off (the default) the search rounds carry no timing events and report no kernel time; on,
    every second launch of a query group is timed and the sum scaled to all launches; the answers are the same."""
    h = maze_ctx
    S, G = make_queries(h.collision, 96, 77, 50, 50)
    h.set_round_timing(0)
    a = h.plan_batch(S[:96], G[:96])
    ms0, n0 = h.kernel_ms(0)
    h.set_round_timing(1)
    b = h.plan_batch(S[:96], G[:96])
    ms1, n1 = h.kernel_ms(0)
    h.set_round_timing(0)
    assert a.tobytes() == b.tobytes()
    assert ms0 == 0 and ms1 > 0 and n1 > 0
    assert 5e-3 < ms1 / n1 < 5.0  # a launch lasts tens of microseconds


def test_full_depth_searches_match_oracle(maze_ctx, oracle_built):
    """Searches at the reference's full iteration limit (10 000), where one wrong verdict, cost bit or tie anywhere
    along the way changes everything after it: result records, expansion order and paths against the CPU oracle,
    one oracle context per thread."""
    h = maze_ctx
    n = 96
    S, G = make_queries(h.collision, n, 4242, 50, 50)
    res = h.plan_batch(S, G)
    assert (res["nodes_expanded"] > 9000).sum() > 20  # many of them run to the limit
    got = [(h.path(i), h.expanded(i)) for i in range(n)]
    bad = _compare_threaded("map_maze", {}, S, G, res, got, range(n))
    assert not bad, sorted(bad)
