#!/usr/bin/env python3
"""Benchmark of the Hybrid A* expansion path on MI355X.

Workload: maps/map_maze.png (50x50 cells, 1 m), a
batch of 1024 random start/goal queries, Reeds-Shepp shot evaluated at every
popped node, reference struct-default parameters with 5 forward + 1 reverse
primitive.  One "step" = one hyas_plan_batch over the whole batch.  At N GPUs
every rank plans its own 1024 queries (weak scaling, no data-path collective).
Every round the search sends each query's popped node and its next best open
nodes to the GPU and uses a result only for the node it pops next, so the
answers are those of expanding one node at a time.

`python bench.py --gpus N` starts the N ranks itself (one process per GPU, the
torch.distributed environment set for each) unless a launcher has already done
so (RANK / WORLD_SIZE in the environment, as under torch.distributed.run).

Prints ONE JSON line on rank 0.  `value` = successors that received a full
collision verdict + g + h, per second, summed over ranks; the map, tables and
query poses are resident before the timed region (per-round node/successor
records cross PCIe inside it, as the host keeps the open/closed sets).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

# as path_planner_amd does when it loads the library; here because torch is imported first (see _lib.load_library)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
CHECKER_NOTE = ("parity checker = CPU restatement of the reference on include/hyas_math.h (bit-exact target); against the "
                "same restatement on glibc libm at most 5 of 50 000 random collision verdicts differ")


def algorithmic_bytes(st):
    """Bytes the round kernel has to move for the work in `st` (hyas_get_stats): per expanded node 20 B in; per emitted
    successor 28 B out + 4 B table + 4 B 2D-cost read; per collision-tested successor / curve sample 12 B pose and
    32 B (4 vertices x 8 B) per obstacle-list entry walked; 1 bit verdict per sample."""
    return (20 * st["nodes"] + 36 * st["succ_emitted"] + 12 * st["succ_evaluated"] + 32 * st["succ_list_entries"]
            + 12 * st["samples_serial"] + 32 * st["sample_list_entries"] + st["samples_serial"] // 8)


def sample_order(n, seed=0):
    """Order in which the CPU baseline takes queries: a seeded shuffle, so that a time-bounded prefix of it is spread
    over the whole batch."""
    return np.random.default_rng(seed).permutation(n)


def cpu_baseline(map_arr, starts, goals, budget_s=15.0):
    """The CPU oracle (serial restatement of the reference) on a bounded sample of the same queries."""
    sys.path.insert(0, str(ROOT / "tests"))
    import orc

    po = orc.hot_path_params()
    o = orc.Oracle(po)
    o.set_costmap(map_arr, build_table=True, nthreads=min(32, os.cpu_count() or 1))
    succ = exp = nq = 0
    sec = 0.0
    for i in sample_order(len(starts)):
        if sec >= budget_s:
            break
        s, res = o.plan_many(starts[i:i + 1], goals[i:i + 1])
        sec += s
        succ += int(res["successors"].sum())
        exp += int(res["nodes_expanded"].sum())
        nq += 1
    return {"value": succ / sec if sec > 0 else 0.0, "unit": "successors/s", "cores": 1, "kind": "port",
            "sample": f"{nq} of the {len(starts)} benchmark queries, drawn across the batch by a seeded shuffle, one at a time "
                      f"on one thread ({sec:.1f} s, {exp} nodes expanded, {nq / sec:.3f} queries/s)"}


def cpu_baseline_all_cores(map_arr, starts, goals, budget_s=12.0, cores=None):
    """The same CPU oracle on every core this process may use: one context and one thread per core, each planning
    its own share of the same queries one at a time (the foreign calls release the interpreter lock)."""
    import threading

    sys.path.insert(0, str(ROOT / "tests"))
    import orc

    cores = cores or max(1, min(64, len(os.sched_getaffinity(0))))
    ctxs = []
    for _ in range(cores):
        o = orc.Oracle(orc.hot_path_params())
        o.set_costmap(map_arr, build_table=True, nthreads=min(32, cores))
        ctxs.append(o)
    tot = [[0, 0, 0] for _ in range(cores)]
    order = sample_order(len(starts), seed=1)
    t0 = time.perf_counter()

    def work(t):
        k = t
        while k < len(order) and time.perf_counter() - t0 < budget_s:
            i = order[k]
            _, res = ctxs[t].plan_many(starts[i:i + 1], goals[i:i + 1])
            tot[t][0] += int(res["successors"].sum())
            tot[t][1] += int(res["nodes_expanded"].sum())
            tot[t][2] += 1
            k += cores

    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    sec = time.perf_counter() - t0
    succ, exp, nq = (sum(c[k] for c in tot) for k in range(3))
    return {"value": succ / sec, "unit": "successors/s", "cores": cores, "kind": "port",
            "sample": f"{nq} of the {len(starts)} benchmark queries (seeded shuffle), one context and one thread per core, each "
                      f"query on one thread ({sec:.1f} s, {exp} nodes expanded, {nq / sec:.3f} queries/s)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=1024, help="queries per GPU per step (queries_large: in all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="queries_maze",
                    choices=["queries_maze", "queries_large", "footprint_checks", "footprint_checks_car", "rs_table", "single_query", "modes"],
                    help="queries_maze is the benchmark the metric is quoted on; the others are the remaining configurations")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: every GPU gets --queries queries; strong: the batch is split into contiguous blocks "
                         "(default: weak for queries_maze, strong for queries_large and rs_table)")
    ap.add_argument("--table-pr", type=int, default=1, help="rs_table: position_resolution (1 or 10)")
    ap.add_argument("--no-large", dest="with_large", action="store_false", default=True,
                    help="queries_maze: leave out the one pass of the map_large strong-scaling batch (65 536 queries over all GPUs) "
                         "that the line otherwise carries as `queries_large_strong`")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, join the process group (gloo) and aggregate placeholder counts without touching a GPU")
    return ap.parse_args(argv)


def dry_run(args):
    """Launcher rehearsal without a GPU: every rank joins a gloo group, contributes its rank's share, rank 0 prints
    the line.  Exercises launch_ranks + aggregate + query_shard exactly as the real run uses them."""
    import torch.distributed as dist

    from path_planner_amd.sharding import aggregate, query_shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("HYAS_DRY_RUN_FAIL_RANK") == str(rank):  # launcher test: one rank dies before the rendezvous
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo")
    b, e = query_shard(args.queries, rank, world) if (args.scaling or "weak") == "strong" else (0, args.queries)
    total_s, (nq, ranks) = aggregate(dist if world > 1 else None, 1.0 + 0.01 * rank, [float(e - b), 1.0])
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": nq / total_s, "unit": "queries/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * total_s, "higher_is_better": True,
                          "scaling": args.scaling or "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "dry run"}, "dry_run": True, "ranks_joined": int(ranks), "queries_total": nq}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    # --gpus N without a launcher: this process starts the N ranks (before it imports torch or touches a GPU), waits
    # for them and hands on rank 0's line
    if args.gpus > 1 and "RANK" not in os.environ:
        from path_planner_amd.sharding import launch_ranks

        codes, out0 = launch_ranks(Path(__file__).resolve(), sys.argv[1:], args.gpus)
        sys.stdout.write(out0 or "")
        sys.stdout.flush()
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad:
            raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}")
        return
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and args.gpus != world_env:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world_env} rank(s); reporting n_gpus = {world_env}", file=sys.stderr)
    if args.dry_run:
        return dry_run(args)
    if args.workload != "queries_maze":
        import bench_extra

        return bench_extra.run(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # rehearsal on a one-GPU box: HYAS_BENCH_BACKEND=gloo HYAS_BENCH_DEVICE=0 puts every rank on GPU 0 without RCCL
    backend = os.environ.get("HYAS_BENCH_BACKEND", "nccl")
    if "HYAS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HYAS_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the expansion path has no CPU fallback")
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from path_planner_amd import Hyas, hot_path_params
    from path_planner_amd.sharding import aggregate, load_map, make_queries, query_shard, result_digest

    scaling = args.scaling or "weak"
    map_arr = load_map("map_maze")
    ctx = Hyas(hot_path_params(), device=local_rank)
    t0 = time.time()
    ctx.set_costmap(map_arr, build_table=True)
    setup_s = time.time() - t0
    table_ms, _ = ctx.kernel_ms(3)
    if scaling == "weak":
        # rank r draws its own queries; rank 0 uses the seed the workload is defined with
        starts, goals = make_queries(ctx.collision, args.queries, 20250228 + 7919 * rank, 50, 50)
    else:
        starts, goals = make_queries(ctx.collision, args.queries, 20250228, 50, 50)
        b, e = query_shard(args.queries, rank, world)
        starts, goals = starts[b:e], goals[b:e]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.plan_batch(starts, goals)

    # the timed steps carry no timing events (they lengthen every launch call); the kernel's duration is taken in an
    # untimed pass of the same batch below
    ctx.set_round_timing(0)
    barrier()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_step = time.perf_counter()
        res = ctx.plan_batch(starts, goals)
        step_s.append(time.perf_counter() - t_step)
    barrier()
    local_s = time.perf_counter() - t0
    # untimed: the same batch once more with HIP events round the launches (every second launch of a query group,
    # on the stream it is launched on; the sum is scaled to all launches)
    ctx.set_round_timing(1)
    ctx.plan_batch(starts, goals)
    kern_ms, kern_launches = ctx.kernel_ms(0)
    ctx.set_round_timing(0)

    succ = float(res["successors"].sum()) * args.steps
    expd = float(res["nodes_expanded"].sum()) * args.steps
    nq = float(len(res)) * args.steps
    total_s, (succ, expd, nq, kern_ms_sum, kern_launches_sum) = aggregate(dist, local_s, [succ, expd, nq, kern_ms, kern_launches])

    # roofline of the dominant kernel (expansion step): one instrumented, untimed pass counts the work
    roofline = None
    cpu = None
    cpu_all = None
    counters = None
    if rank == 0:
        ctx.set_stats(1)
        res1 = ctx.plan_batch(starts, goals)
        st = ctx.stats()
        _, launches_1 = ctx.kernel_ms(0)
        ctx.set_stats(0)
        consumed = int(res1["nodes_expanded"].sum())
        counters = {k: st[k] for k in ("nodes", "shots", "samples_tested", "samples_serial", "sample_list_entries", "succ_evaluated",
                                       "succ_list_entries", "succ_emitted")}
        counters["nodes_consumed_by_the_search"] = consumed
        counters["successors_consumed_by_the_search"] = int(res1["successors"].sum())
        bytes_per_launch = algorithmic_bytes(st) / max(launches_1, 1)
        avg_ms = kern_ms / max(kern_launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        traffic_source = None
        # HBM-side bytes per launch from the rocprofv3 counter passes (profiles/collect_traffic.sh), newest file: taken
        # on the builder's box with a shorter workload (300 iterations per query), not in this run
        for tf in sorted((ROOT / "profiles").glob("r*_step_kernel_traffic.json"), reverse=True):
            try:
                traffic = json.loads(tf.read_text()).get("hbm_bytes_per_launch")
                traffic_source = f"profiles/{tf.name}: counter passes of an earlier run (1024 maze queries, 300 iterations each), not of this one"
                break
            except Exception:
                traffic = None
        l2 = None
        # L2-side requests of the same kernel (profiles/collect_l2.sh), newest file
        for lf in sorted((ROOT / "profiles").glob("r*_step_kernel_l2.json"), reverse=True):
            try:
                d = json.loads(lf.read_text())
                l2 = {k: d[k] for k in ("l2_requests_per_launch", "l2_hit_rate", "l2_GBps_at_64B_per_request",
                                        "l2_GBps_at_128B_per_request", "l2_peak_GBps")}
                l2["source"] = f"profiles/{lf.name}: counter passes of an earlier run (1024 maze queries, 300 iterations each), not of this one"
                break
            except Exception:
                l2 = None
        frac = achieved / HBM_PEAK_GBS
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
                    "useful_frac": frac * consumed / max(st["nodes"], 1),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "moved_bytes_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                    "kernel": "k_round", "avg_launch_us": avg_ms * 1e3,
                    "avg_launch_us_source": "HIP events on the launching streams in an untimed pass of the same batch (the timed steps carry no events)",
                    "algorithmic_bytes_per_launch": bytes_per_launch, "launches_per_step": launches_1, "l2": l2,
                    "note": "map, obstacle lists and tables are <= 30 MB and cache-resident: HBM and L2 fractions are low by "
                            "construction; a launch lasts as long as its slowest workgroup's dependent chain (solve, pick, "
                            "interpolate, list walk), stretched by the launches that share the SIMDs. Bytes count every node a "
                            "launch processed, including nodes expanded ahead of time (counters.nodes); `value` and useful_frac "
                            "count only what the search consumed (counters.nodes_consumed_by_the_search). The bytes are those "
                            "of the reference's serial scan (every shot sample up to the first colliding one with its obstacle "
                            "lists); the kernel decides most shots from one cost byte per sample and reads far fewer list "
                            "entries (compare `traffic` and roofline.l2), so `achieved` is a rate of algorithmic work, not "
                            "of bytes moved"}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(map_arr, starts, goals)
            cpu_all = cpu_baseline_all_cores(map_arr, starts, goals)

    # the batched-query strong-scaling workload beside the line's own metric (every rank takes part)
    curve = None
    if args.with_large:
        import bench_extra

        ctx.close()
        curve = bench_extra.north_star_curve(torch, dist, rank, local_rank, world)

    if rank == 0:
        # what the timed steps answered; for the workload as defined (1024 queries, seed 20250228) the GPU test
        # tests/test_gpu_fullsize.py::test_maze_1024_queries_properties compares every query of this very batch with
        # the CPU oracle and pins the same digest in tests/golden/bench_digest.json
        digest = result_digest(res)
        verified = None
        if args.queries == 1024 and (scaling == "weak" or world == 1):
            try:
                verified = digest == json.loads((ROOT / "tests" / "golden" / "bench_digest.json").read_text())["queries_maze"]
            except Exception:
                verified = None
        status = {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}
        per = "per GPU" if scaling == "weak" else f"in all, contiguous blocks over {world} GPU(s)"
        line = {
            "metric": "collision-checked node expansions/sec",
            "value": succ / total_s,
            "unit": "successors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * total_s / args.steps,
            "ms_per_step_median_rank0": 1e3 * float(np.median(step_s)),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"map_maze 50x50, {args.queries} random start/goal queries {per}, Reeds-Shepp shot every pop",
                       "queries_per_gpu": len(starts), "parallelism": f"queries sharded over {world} GPU(s)",
                       "params": "reference struct defaults except curve_type=1, 5+1 primitives, short_cut=false; max_iterations=10000",
                       "checker": CHECKER_NOTE},
            "queries_per_s": nq / total_s,
            "nodes_expanded_per_s": expd / total_s,
            "status_counts_rank0": status,
            "result_digest_rank0": digest,
            "result_digest_verified": verified,
            "setup_s": setup_s,
            "table_build_ms": table_ms,
            "counters_rank0_one_step": counters,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "queries_large_strong": curve,
        }
        print(json.dumps(line))
    # The line is out.  The process holds two HIP runtimes (torch's own and the one the library links), and one run in
    # several hundred did not come back from tearing things down after its line: if the rest of the process (the last
    # barrier, the context's teardown, the interpreter's) takes longer than two minutes, the alarm's default action ends it.
    sys.stdout.flush()
    sys.stderr.flush()
    import signal

    signal.alarm(120)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
