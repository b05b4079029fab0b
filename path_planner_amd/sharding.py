"""Multi-GPU partitioning of the two parts of the path that shard: independent
planning queries (no data-path collective) and the heading slices of the
Reeds-Shepp heuristic table (one all-gather).  One process per GPU; the process
group is whatever torch.distributed was initialised with ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests)."""
import numpy as np


def load_map(name):
    """One of the occupancy maps the benchmarks and tests run on (uint8 [h][w], costmap_2d cost values), converted
    from the reference's maps/*.png by tests/golden/make_maps.py."""
    from pathlib import Path

    return np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "maps" / f"{name}.npy")


def launch_ranks(script, argv, n, timeout=None):
    """Starts `n` processes of `script argv...`, one per GPU, with the torch.distributed environment
    (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT) set, before this process has touched a
    GPU.  Returns (exit codes, rank 0's stdout).  Ranks other than 0 inherit stdout / stderr.  All ranks are watched
    together: as soon as one exits with a failure, or `timeout` seconds have passed, the others are terminated (they
    would otherwise sit in a rendezvous or a collective until its own timeout) and reaped; a rank that had to be stopped
    reports a negative code (its signal) or, after the time limit, 124."""
    import os
    import socket
    import subprocess
    import sys
    import threading
    import time

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    out0 = []
    reader = None
    timed_out = False
    try:
        for r in range(n):
            procs.append(subprocess.Popen([sys.executable, str(script)] + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                          stdout=subprocess.PIPE if r == 0 else None, text=True))
        # rank 0's pipe is drained on a thread of its own, so that a full pipe never blocks the rank
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            codes_now = [p.poll() for p in procs]  # every rank is asked, not only up to the first one still running
            if all(c is not None for c in codes_now) or any(c not in (None, 0) for c in codes_now):
                break
            if deadline is not None and time.monotonic() > deadline:
                timed_out = True
                break
            time.sleep(0.05)
    finally:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_kill = time.monotonic() + 5
        for p in alive:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if reader is not None:
            reader.join(timeout=5)
    codes = [124 if (timed_out and p.returncode is not None and p.returncode < 0) else p.returncode for p in procs]
    return codes, (out0[0] if out0 else "")


def query_shard(n_total, rank, world):
    """Contiguous block [begin, end) of n_total queries for `rank`; blocks differ by at most one query."""
    base, extra = divmod(n_total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def heading_slices(n_slices, world):
    """The table's heading samples (73 for 72 headings: src/lookup_table.cpp:182-224 runs theta from 0 while
    theta < 2*pi in float steps) split into `world` runs of `per` slices; trailing ranks may get fewer or none.
    Returns ([(k0, k1)] * world, per)."""
    per = -(-n_slices // world)
    return [(min(n_slices, r * per), min(n_slices, (r + 1) * per)) for r in range(world)], per


def gather_slices(local, n_slices, dist=None):
    """all-gather of the per-rank slice blocks.  `local` is a [per, ny, nx] float32 tensor whose first
    (k1 - k0) slices are valid; returns the [n_slices, ny, nx] table in heading order."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local[:n_slices].contiguous()
    world = dist.get_world_size()
    dev = local.device
    if dist.get_backend() != "nccl" and dev.type == "cuda":
        local = local.cpu()  # CPU rehearsal backends gather host tensors
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous())
    return full[:n_slices].to(dev).contiguous()


def build_table_sharded(ctx, curve, dist=None, device=None):
    """Each rank evaluates its heading slices on its GPU (hyas_table_build_dev), one all-gather assembles the
    raw table on every rank, and every rank installs it under the reference's keys with the first write winning
    (hyas_table_install_dev).  Returns (seconds_build, seconds_gather, seconds_install)."""
    import time

    import torch

    nx, ny, ns = ctx.table_dims()
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    slices, per = heading_slices(ns, world)
    k0, k1 = slices[rank]
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    local = torch.zeros((per, ny, nx), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if k1 > k0:
        ctx.table_build_dev(curve, k0, k1, local.data_ptr())
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    full = gather_slices(local, ns, dist)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    ctx.table_install_dev(curve, full.data_ptr())
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


def aggregate(dist, local_seconds, local_counts):
    """MAX of the ranks' times and SUM of their work counters."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_seconds, [float(v) for v in local_counts]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([local_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(v) for v in local_counts], dtype=torch.float64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(v) for v in c.tolist()]


def result_digest(res):
    """What a planning batch answered, in two numbers that a benchmark line and a test can both state: the CRC-32 of the
    result records (status, counters, path length and costs of every query, in query order) and the nodes expanded."""
    import zlib

    res = np.ascontiguousarray(res)
    return {"crc32_of_result_records": int(zlib.crc32(res.tobytes())), "nodes_expanded": int(res["nodes_expanded"].sum()),
            "queries": int(len(res))}


def make_queries(collision_fn, n, seed, width, height):
    """n start/goal pairs: x uniform in [1, width-1), y in [1, height-1), theta in [0, 2*pi), float32; poses the
    polygon-mode traversability test rejects are redrawn.  `collision_fn(xyt) -> uint8 verdicts`."""
    import math

    rng = np.random.default_rng(seed)

    def draw(k):
        out = np.zeros((0, 3), np.float32)
        while out.shape[0] < k:
            m = 2 * k
            p = np.stack([rng.uniform(1, width - 1, m), rng.uniform(1, height - 1, m), rng.uniform(0, 2 * math.pi, m)], 1).astype(np.float32)
            ok = collision_fn(p)
            out = np.concatenate([out, p[ok == 1]])
        return out[:k].copy()

    return draw(n), draw(n)
