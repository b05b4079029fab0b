"""CPU suite: the multi-GPU partitioning, rehearsed with world_size 2 and 3 on gloo."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from path_planner_amd import sharding

ROOT = Path(__file__).resolve().parent.parent


def test_query_shards_tile_the_batch():
    for n in (0, 1, 7, 1024, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            edges = [sharding.query_shard(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1


def test_heading_slices_cover_73_samples():
    for world in (1, 2, 3, 4, 8, 16):
        sl, per = sharding.heading_slices(73, world)
        assert per * world >= 73
        covered = [k for a, b in sl for k in range(a, b)]
        assert covered == list(range(73))
        assert all(b - a <= per for a, b in sl)
    sl, per = sharding.heading_slices(73, 8)
    assert per == 10 and sl[7] == (70, 73)


def test_make_queries_is_seeded_and_rejects_blocked_poses():
    calls = []

    def verdict(p):
        calls.append(len(p))
        return (p[:, 0] > 10).astype(np.uint8)

    s1, g1 = sharding.make_queries(verdict, 100, 5, 50, 50)
    s2, g2 = sharding.make_queries(verdict, 100, 5, 50, 50)
    assert np.array_equal(s1, s2) and np.array_equal(g1, g2)
    assert (s1[:, 0] > 10).all() and (g1[:, 0] > 10).all()
    assert s1.dtype == np.float32 and s1.shape == (100, 3)
    s3, _ = sharding.make_queries(verdict, 100, 6, 50, 50)
    assert not np.array_equal(s1, s3)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks(world, tmp_path, oracle_built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]
    for r in res:
        assert r["agg_t"] == float(world)  # max over ranks of 1 + rank
        assert r["agg_counts"] == [10.0 * world * (world + 1) / 2, 7.0 * world]
        assert r["table_equal"], "all-gathered heading slices differ from the single-rank table"
        assert r["table_shape"][0] == 73
    assert [r["shard"] for r in res] == [list(sharding.query_shard(11, r, world)) for r in range(world)]
