"""The benchmark configurations besides the headline one (bench.py --workload ...):

 footprint_checks  4096 x 4096 random-occupancy grid (10 % occupied), 10 M footprint-list collision checks
                   per GPU per step, poses resident in HBM;
 rs_table          Reeds-Shepp heuristic table of a 180 x 60 map (rho 0.35), heading slices sharded over the
                   GPUs and assembled with one all-gather, then installed on every GPU;
 queries_large     65 536 planning queries on maps/map_large sharded over the GPUs (strong scaling);
 single_query      the maze queries one per call, as the reference's makePlan hands them over: latency per query;
 modes             the maze batch under the reference's other configurations: adaptive steering angle and step size,
                   adaptive step size, cubic Bezier curves, the shipped parameter file (obstacle range 6 m), and the
                   path shortcut on its own (hyas_shortcut_batch), each with the map set-up time.

Same output contract as bench.py: one JSON line on rank 0.
"""
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0


def _dist_setup():
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the expansion path has no CPU fallback")
    backend = os.environ.get("HYAS_BENCH_BACKEND", "nccl")  # "gloo" + HYAS_BENCH_DEVICE=0: rehearsal on one GPU
    if "HYAS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HYAS_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return torch, dist, rank, local_rank, world


def _barrier(torch, dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _newest_profile(pattern):
    """The newest profiles/rNN_<pattern>.json (rocprofv3 counter summaries kept in the tree), or None."""
    for f in sorted((ROOT / "profiles").glob(f"r*_{pattern}.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
            d["source"] = f"profiles/{f.name}"
            return d
        except Exception:
            continue
    return None


# footprints of the footprint_checks workloads: (width, length) of the vehicle rectangle in cells
FOOTPRINTS = {"in_tree": (0.18, 0.22), "car": (2.0, 4.5)}


def _footprint_l2(prof, avg_ms):
    """What bounds the kernel: every random 8-byte gather from the occupancy bits moves a whole 128-byte line from L2
    to a compute unit's L1 (the bits of a 4096 x 4096 grid are 2 MB: resident in L2, far too large for L1 or LDS)."""
    if not prof:
        return None
    req = prof["counters_per_launch"].get("TCC_REQ_sum")
    if not req:
        return None
    gbs = req * 128.0 / (avg_ms * 1e-3) / 1e9
    return {"l2_requests_per_launch": req, "bytes_per_request": 128, "achieved_GBps": gbs, "peak_GBps": 34500.0, "frac": gbs / 34500.0,
            "note": "requests x 128 B over the launch duration measured in this run; L2 peak = 16 channels x 128 B/clk x 8 XCDs x 2.1 GHz"}


def footprint_checks(args, footprint="in_tree"):
    torch, dist, rank, local_rank, world = _dist_setup()
    from path_planner_amd import Hyas, default_params
    from path_planner_amd._lib import MODE_FOOTPRINT
    from path_planner_amd.sharding import aggregate

    W = H = 4096
    n = 10_000_000
    fpw, fpl = FOOTPRINTS[footprint]
    grid = np.where(np.random.default_rng(1).random((H, W)) < 0.10, 254, 0).astype(np.uint8)
    rng = np.random.default_rng(2 + 7919 * rank)
    xyt = np.stack([rng.uniform(8, W - 8, n), rng.uniform(8, H - 8, n), rng.uniform(0, 2 * math.pi, n)], 1).astype(np.float32)
    # in_tree: the constants of include/constants.h (7200 lists of at most 4 cells); car: a 2 x 4.5 cell rectangle
    # (lists of up to 64 cells, the capacity of Constants::config)
    ctx = Hyas(default_params(fp_width=fpw, fp_length=fpl), device=local_rank)
    ctx.set_grid(grid)
    fp_len = ctx.footprint_table()[:, 0]
    d_xyt = torch.from_numpy(xyt).cuda()
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        ctx.collision_dev(d_xyt.data_ptr(), None, n, d_out.data_ptr(), MODE_FOOTPRINT)
    _barrier(torch, dist)
    ctx.reset_kernel_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.collision_dev(d_xyt.data_ptr(), None, n, d_out.data_ptr(), MODE_FOOTPRINT)
    _barrier(torch, dist)
    local_s = time.perf_counter() - t0
    kms, kl = ctx.kernel_ms(2)
    free = int(d_out.sum().item())
    total_s, (checks,) = aggregate(dist, local_s, [float(n) * args.steps])
    if rank != 0:
        return None
    # algorithmic bytes: 12 B pose + 1 B verdict + one grid byte per footprint cell visited
    it = (xyt[:, 2] / np.float32(2 * math.pi / 72)).astype(np.int64).clip(0, 71)
    ix = ((xyt[:, 0] - np.floor(xyt[:, 0])) * 10).astype(np.int64).clip(0, 9)
    iy = ((xyt[:, 1] - np.floor(xyt[:, 1])) * 10).astype(np.int64).clip(0, 9)
    mean_len = float(np.minimum(fp_len, 64)[(iy * 10 + ix) * 72 + it].mean())  # the test reads at most 64 cells of a list
    bytes_per_launch = n * (13.0 + mean_len)
    avg_ms = kms / max(kl, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    prof = _newest_profile(f"footprint_{footprint}_traffic")
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "tests"))
        import orc

        o = orc.Oracle(orc.default_params(fp_width=fpw, fp_length=fpl))
        o.set_grid(grid)
        m = 2_000_000
        t0 = time.perf_counter()
        ref = o.collision(xyt[:m], mode=1)
        cs = time.perf_counter() - t0
        same = bool(np.array_equal(ref, d_out[:m].cpu().numpy()))
        cpu = {"value": m / cs, "unit": "checks/s", "cores": 1, "kind": "port",
               "sample": f"first {m} of the {n} poses on one thread ({cs:.2f} s); verdicts equal to the GPU's: {same}"}
    return {
        "metric": "footprint collision checks/sec", "value": checks / total_s, "unit": "checks/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * total_s / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {"workload": f"4096x4096 grid, 10 % occupied, 10M footprint-list checks per GPU, {footprint} footprint "
                               f"({fpl} x {fpw} cells, lists of up to {int(fp_len.max())} cells)",
                   "free_fraction_rank0": free / n, "mean_cells_per_check": mean_len},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_of_achievable_6300": achieved / 6300.0,
                     "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
                     "kernel": (prof or {}).get("kernel", "k_collision_footprint_m8" if footprint == "in_tree" else "k_collision_footprint_window"),
                     "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "l2": _footprint_l2(prof, avg_ms), "counters": prof},
        "cpu_baseline": cpu,
    }


def rs_table(args):
    torch, dist, rank, local_rank, world = _dist_setup()
    from path_planner_amd import Hyas, hot_path_params
    from path_planner_amd.sharding import aggregate, build_table_sharded

    pr = args.table_pr
    grid = np.zeros((60, 180), np.uint8)
    ctx = Hyas(hot_path_params(position_resolution=pr, cd_min_turning_radius=0.35), device=local_rank)
    ctx.set_costmap(grid, build_table=False)
    nx, ny, ns = ctx.table_dims()
    entries = nx * ny * ns
    parts = [0.0, 0.0, 0.0]
    for _ in range(args.warmup):
        build_table_sharded(ctx, 1, dist)
    _barrier(torch, dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b, g, i = build_table_sharded(ctx, 1, dist)
        parts[0] += b
        parts[1] += g
        parts[2] += i
    _barrier(torch, dist)
    local_s = time.perf_counter() - t0
    total_s, _ = aggregate(dist, local_s, [0.0])
    # spot-check: the installed table answers lookups
    probe = np.array([[3.0, 2.0, 1.0], [100.0, 40.0, 3.0]], np.float32)
    cost = ctx.table_lookup(probe)
    if rank != 0:
        return None
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "tests"))
        import orc

        o = orc.Oracle(orc.hot_path_params(position_resolution=pr, cd_min_turning_radius=0.35))
        o.set_costmap(grid, build_table=False)
        k1 = 1 if pr > 1 else 8
        t0 = time.perf_counter()
        o.table_build(1, 0, k1)
        cs = time.perf_counter() - t0
        cpu = {"value": k1 * nx * ny / cs, "unit": "entries/s", "cores": 1, "kind": "port",
               "sample": f"{k1} of the {ns} heading slices on one thread ({cs:.2f} s)"}
    per_build = total_s / args.steps
    kms, kl = ctx.kernel_ms(3)
    # FP64 side: vector f64 operations per entry from the rocprofv3 counter pass (profiles/collect_table_f64.sh)
    f64 = _newest_profile("table_rs_f64")
    compute = None
    if f64 and kl:
        k_s = (kms / kl) * 1e-3  # one k_table_rs launch = this rank's slices
        my_entries = entries / world if world > 1 else entries
        tf = f64["f64_flops_per_entry"] * my_entries / k_s / 1e12
        compute = {"bound": "valu_f64", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                   "f64_flops_per_entry": f64["f64_flops_per_entry"], "kernel_ms": kms / kl, "source": f64["source"],
                   "note": "peak = MI355X FP64 vector spec; flops = SQ f64 add + mul + 2 x fma + transcendental wave "
                           "instructions x 64 lanes, i.e. issued lane slots: lanes switched off inside a closed form count"}
    return {
        "metric": "Reeds-Shepp heuristic table entries/sec", "value": entries / per_build, "unit": "entries/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * per_build, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"180x60 map, position_resolution {pr}, rho 0.35: {entries} entries, {ns} heading slices over {world} GPU(s)",
                   "seconds_build_gather_install_rank0": [p / args.steps for p in parts], "probe_costs": [float(v) for v in cost]},
        "roofline": {"bound": "hbm", "achieved": 4.0 * entries / (parts[0] / args.steps) / 1e9 if parts[0] > 0 else 0.0,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 4.0 * entries / (parts[0] / args.steps) / 1e9 / HBM_PEAK_GBS if parts[0] > 0 else 0.0,
                     "traffic": None, "kernel": "k_table_rs", "compute": compute,
                     "note": "4 B stored per entry; the kernel is f64 ALU bound (44 closed-form candidates per entry), the HBM fraction is low by construction; see roofline.compute"},
        "cpu_baseline": cpu,
    }


def _round_kernel_roofline(ctx, calls, what):
    """The search-round kernel's roofline object for a workload given as a list of (starts, goals) calls, measured in
    two untimed passes over them: one with HIP events round the launches (duration), one with the work counters on
    (algorithmic bytes, as bench.py counts them)."""
    from bench import HBM_PEAK_GBS, algorithmic_bytes

    ctx.set_round_timing(1)
    kms = kl = 0
    for s_, g_ in calls:
        ctx.plan_batch(s_, g_)
        a, b = ctx.kernel_ms(0)
        kms, kl = kms + a, kl + b
    ctx.set_round_timing(0)
    ctx.set_stats(1)
    launches = 0
    for s_, g_ in calls:
        ctx.plan_batch(s_, g_)
        launches += ctx.kernel_ms(0)[1]
    st = ctx.stats()
    ctx.set_stats(0)
    avg_ms = kms / max(kl, 1)
    per_launch = algorithmic_bytes(st) / max(launches, 1)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": "k_round", "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": per_launch, "launches": launches,
            "measured_on": what,
            "note": "HIP events on the launching streams, in passes of their own; bytes are those of the reference's serial scan "
                    "(bench.py algorithmic_bytes), every node a launch processed counted; no counter pass for this workload"}


def queries_large_run(torch, dist, rank, local_rank, world, total_q, scaling, warmup, steps):
    """The map_large batch on an initialised process group: every rank plans its block (strong) or its own total_q
    queries (weak); returns the aggregated time and counters and rank-local objects for the caller's extras."""
    from path_planner_amd import Hyas, hot_path_params
    from path_planner_amd.sharding import aggregate, load_map, make_queries, query_shard

    mp = load_map("map_large")
    H, W = mp.shape
    ctx = Hyas(hot_path_params(), device=local_rank)
    ctx.set_costmap(mp, build_table=True)
    if scaling == "strong":
        starts, goals = make_queries(ctx.collision, total_q, 3, W, H)  # every rank draws the same batch, then takes its block
        b, e = query_shard(total_q, rank, world)
        starts, goals = starts[b:e], goals[b:e]
    else:
        starts, goals = make_queries(ctx.collision, total_q, 3 + 7919 * rank, W, H)  # total_q queries per GPU
    for _ in range(warmup):
        ctx.plan_batch(starts[: min(len(starts), 2048)], goals[: min(len(goals), 2048)])
    _barrier(torch, dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = ctx.plan_batch(starts, goals)
    _barrier(torch, dist)
    local_s = time.perf_counter() - t0
    succ = float(res["successors"].sum()) * steps
    expd = float(res["nodes_expanded"].sum()) * steps
    total_s, (succ, expd, nq) = aggregate(dist, local_s, [succ, expd, float(len(res)) * steps])
    return dict(total_s=total_s, succ=succ, expd=expd, nq=nq, res=res, ctx=ctx, starts=starts, goals=goals, mp=mp)


def north_star_curve(torch, dist, rank, local_rank, world, total_q=65536):
    """One pass of the batched-query strong-scaling workload (total_q map_large queries in contiguous blocks over the
    ranks), for the default bench line to carry beside its own metric.  Every rank must call it."""
    r = queries_large_run(torch, dist, rank, local_rank, world, total_q, "strong", 1, 1)
    r["ctx"].close()
    if rank != 0:
        return None
    return {"workload": f"map_large 180x60, {total_q} queries in contiguous blocks over {world} GPU(s), Reeds-Shepp shot every pop; one pass",
            "scaling": "strong", "seconds": r["total_s"], "queries_per_s": r["nq"] / r["total_s"], "successors_per_s": r["succ"] / r["total_s"],
            "nodes_expanded_per_s": r["expd"] / r["total_s"]}


def queries_large(args):
    torch, dist, rank, local_rank, world = _dist_setup()
    scaling = args.scaling or "strong"
    total_q = args.queries if args.queries != 1024 else 65536
    r = queries_large_run(torch, dist, rank, local_rank, world, total_q, scaling, args.warmup, args.steps)
    total_s, succ, expd, nq, res, ctx, starts, goals, mp = (r[k] for k in ("total_s", "succ", "expd", "nq", "res", "ctx", "starts", "goals", "mp"))
    if rank != 0:
        return None
    status = {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}
    sub = min(len(starts), 2048)
    roofline = _round_kernel_roofline(ctx, [(starts[:sub], goals[:sub])], f"the first {sub} queries of rank 0's block, one call")
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from bench import cpu_baseline

        cpu = cpu_baseline(mp, starts, goals, budget_s=12.0)
    return {
        "metric": "collision-checked node expansions/sec", "value": succ / total_s, "unit": "successors/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * total_s / args.steps, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"map_large 180x60, {total_q} queries " + ("in contiguous blocks over" if scaling == "strong" else "on each of")
                               + f" {world} GPU(s), Reeds-Shepp shot every pop"},
        "queries_per_s": nq / total_s, "nodes_expanded_per_s": expd / total_s, "status_counts_rank0": status,
        "round_kernel_avg_us_rank0": roofline["avg_launch_us"],
        "roofline": roofline, "cpu_baseline": cpu,
    }


def single_query(args):
    """One query per hyas_plan_batch call (HybridAStarPlanner::makePlan, src/hybrid_a_star_planner.cpp:99-124, plans one
    start/goal pair per call): a step is one pass over the first 64 benchmark queries of every GPU, one after the other."""
    torch, dist, rank, local_rank, world = _dist_setup()
    from path_planner_amd import Hyas, hot_path_params
    from path_planner_amd.sharding import aggregate, load_map, make_queries

    mp = load_map("map_maze")
    ctx = Hyas(hot_path_params(), device=local_rank)
    ctx.set_costmap(mp, build_table=True)
    nq = 64 if args.queries == 1024 else args.queries
    starts, goals = make_queries(ctx.collision, nq, 20250228 + 7919 * rank, 50, 50)
    for _ in range(args.warmup):
        ctx.plan_batch(starts[:1], goals[:1])
    _barrier(torch, dist)
    lat = []
    succ = expd = launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for i in range(nq):
            t1 = time.perf_counter()
            r = ctx.plan_batch(starts[i:i + 1], goals[i:i + 1])
            lat.append(time.perf_counter() - t1)
            succ += int(r["successors"][0])
            expd += int(r["nodes_expanded"][0])
            launches += ctx.kernel_ms(0)[1]
    _barrier(torch, dist)
    local_s = time.perf_counter() - t0
    total_s, (succ, expd, n_done, launches) = aggregate(dist, local_s, [float(succ), float(expd), float(nq * args.steps), float(launches)])
    if rank != 0:
        return None
    lat = np.array(lat) * 1e3
    roofline = _round_kernel_roofline(ctx, [(starts[i:i + 1], goals[i:i + 1]) for i in range(nq)], f"rank 0's {nq} queries, one call each")
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from bench import cpu_baseline

        cpu = cpu_baseline(mp, starts, goals, budget_s=12.0)
    return {
        "metric": "collision-checked node expansions/sec", "value": succ / total_s, "unit": "successors/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * total_s / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"map_maze 50x50, {nq} queries per GPU, one query per call, Reeds-Shepp shot every pop"},
        "queries_per_s": n_done / total_s, "nodes_expanded_per_s": expd / total_s,
        "latency_ms_rank0": {"mean": float(lat.mean()), "median": float(np.median(lat)), "p90": float(np.percentile(lat, 90)), "max": float(lat.max())},
        "rounds_per_query": launches / n_done, "expansions_per_round": expd / max(launches, 1.0),
        "roofline": roofline, "cpu_baseline": cpu,
    }


def modes(args):
    """The 1024-query maze batch under the configurations beside the benchmark's (one GPU): per configuration the map
    set-up time (obstacle lists, adaptive maps, heuristic table), the batch time and rates, and the CPU oracle on a
    bounded sample of the same queries; then hyas_shortcut_batch on the paths the batch found."""
    torch, dist, rank, local_rank, world = _dist_setup()
    from path_planner_amd import Hyas, hot_path_params
    from path_planner_amd import _lib as hl
    from path_planner_amd.sharding import load_map, make_queries

    mp = load_map("map_maze")
    H, W = mp.shape
    base = Hyas(hot_path_params(), device=local_rank)
    base.set_costmap(mp, build_table=True)
    nq = args.queries
    starts, goals = make_queries(base.collision, nq, 20250228, W, H)
    shipped = {k: v for k, v in hl.SHIPPED_YAML.items()}
    shipped.update(obstacle_detection_range=6.0, max_iterations=10000)
    variants = {
        "benchmark (Reeds-Shepp, fixed motion set)": {},
        "adaptive steering angle and step size": dict(adaptive_steering_angle_and_step_size=1),
        "adaptive step size": dict(adaptive_step_size=1),
        "cubic Bezier curves": dict(curve_type=2),
        "Dubins curves, shots at the paper's schedule": dict(curve_type=0, analytical_expansion_every_point=0),
        "shipped parameter file with a 6 m obstacle range": shipped,
    }
    rows = {}
    cpu_on = world == 1 and not args.no_cpu_baseline
    if cpu_on:
        sys.path.insert(0, str(ROOT / "tests"))
        import orc
    keep_paths = None
    for name, kw in variants.items():
        make = hl.default_params if kw is shipped else hot_path_params
        ctx = Hyas(make(**kw), device=local_rank)
        t0 = time.perf_counter()
        ctx.set_costmap(mp, build_table=True)
        setup_ms = (time.perf_counter() - t0) * 1e3
        ctx.plan_batch(starts[:64], goals[:64])
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = ctx.plan_batch(starts, goals)
        sec = (time.perf_counter() - t0) / args.steps
        ctx.set_round_timing(1)  # the timed calls carry no timing events: one more pass for the kernel's duration
        ctx.plan_batch(starts, goals)
        kms, kl = ctx.kernel_ms(0)
        ctx.set_round_timing(0)
        status = {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}
        row = {"map_setup_ms": setup_ms, "ms_per_batch": sec * 1e3, "queries_per_s": nq / sec,
               "successors_per_s": float(res["successors"].sum()) / sec, "nodes_expanded_per_s": float(res["nodes_expanded"].sum()) / sec,
               "round_kernel_avg_us": 1e3 * kms / max(kl, 1), "rounds_per_group": kl, "status_counts": status}
        if name.startswith("benchmark"):
            keep_paths = [ctx.path(i) for i in range(nq) if res["status"][i] in (hl.STATUS_SHOT, 1) and res["path_len"][i] >= 3]
        if cpu_on:
            make_o = orc.default_params if kw is shipped else orc.hot_path_params
            o = orc.Oracle(make_o(**kw))
            t0 = time.perf_counter()
            o.set_costmap(mp, build_table=True, nthreads=1)
            o_setup = time.perf_counter() - t0
            order = np.random.default_rng(1).permutation(nq)
            t0 = time.perf_counter()
            done = succ = 0
            for i in order:
                r, _, _ = o.plan(starts[i], goals[i])
                done += 1
                succ += int(r["successors"])
                if time.perf_counter() - t0 > 4.0:
                    break
            cs = time.perf_counter() - t0
            row["cpu_oracle"] = {"map_setup_ms": o_setup * 1e3, "queries_per_s": done / cs, "successors_per_s": succ / cs, "cores": 1,
                                 "sample": f"{done} of the {nq} queries (seeded shuffle) in {cs:.1f} s on one thread"}
        rows[name] = row
        ctx.close()
    # ShortCut on its own: the paths of the benchmark configuration (short_cut is off there)
    sc = None
    if keep_paths:
        base.shortcut(keep_paths[:8])
        t0 = time.perf_counter()
        for _ in range(args.steps):
            kept = base.shortcut(keep_paths)
        sec = (time.perf_counter() - t0) / args.steps
        poses = int(sum(len(p) for p in keep_paths))
        sc = {"paths": len(keep_paths), "poses": poses, "ms_per_call": sec * 1e3, "paths_per_s": len(keep_paths) / sec,
              "poses_per_s": poses / sec, "poses_kept": int(sum(len(k) for k in kept))}
        if cpu_on:
            o = orc.Oracle(orc.hot_path_params())
            o.set_costmap(mp, build_table=False)
            t0 = time.perf_counter()
            done = 0
            for p_ in keep_paths:
                o.shortcut(p_)
                done += 1
                if time.perf_counter() - t0 > 4.0:
                    break
            cs = time.perf_counter() - t0
            sc["cpu_oracle"] = {"paths_per_s": done / cs, "cores": 1, "sample": f"{done} paths in {cs:.1f} s on one thread"}
    if rank != 0:
        return None
    head = rows["shipped parameter file with a 6 m obstacle range"]
    return {
        "metric": "collision-checked node expansions/sec", "value": head["successors_per_s"], "unit": "successors/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_batch"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"map_maze 50x50, {nq} queries, one GPU; `value` is the shipped-parameter-file row"},
        "rows": rows, "shortcut": sc, "roofline": None, "cpu_baseline": head.get("cpu_oracle"),
    }


def run(args):
    fn = {"single_query": single_query, "modes": modes, "footprint_checks": footprint_checks, "footprint_checks_car": lambda a: footprint_checks(a, "car"), "rs_table": rs_table,
          "queries_large": queries_large}[args.workload]
    line = fn(args)
    if line is not None:
        print(json.dumps(line))
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
