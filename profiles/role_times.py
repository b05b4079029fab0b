"""Times of the two roles of a search round alone on the GPU: k_shot and k_expand on n random maze nodes."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from path_planner_amd import Hyas, hot_path_params
from path_planner_amd._lib import NODE_DT, GOAL_DT
from path_planner_amd.sharding import load_map, make_queries

h = Hyas(hot_path_params(), device=0)
h.set_costmap(load_map("map_maze"), build_table=True)
for n in (64, 570, 2048):
    S, G = make_queries(h.collision, n, 7, 50, 50)
    nodes = np.zeros(n, NODE_DT); goals = np.zeros(n, GOAL_DT)
    nodes["x"], nodes["y"], nodes["t"], nodes["g"] = S[:, 0], S[:, 1], S[:, 2], 10.0
    goals["x"], goals["y"], goals["t"] = G[:, 0], G[:, 1], G[:, 2]
    h.shot(nodes, goals); h.expand(nodes, goals)
    ts, te = [], []
    for _ in range(20):
        ok, ln = h.shot(nodes, goals); ms, l = h.kernel_ms(0); ts.append(1e3 * ms / max(l, 1))
        out, cnt = h.expand(nodes, goals); ms, l = h.kernel_ms(1); te.append(1e3 * ms / max(l, 1))
    print(n, "shot us", round(float(np.median(ts)), 1), "ok", int(ok.sum()), "mean len", round(float(ln.mean()), 1), "expand us", round(float(np.median(te)), 1), "succ", int(cnt.sum()))
# failing shots only, as in a search
for n in (64, 570):
    S, G = make_queries(h.collision, 4 * n, 9, 50, 50)
    nodes = np.zeros(4 * n, NODE_DT); goals = np.zeros(4 * n, GOAL_DT)
    nodes["x"], nodes["y"], nodes["t"], nodes["g"] = S[:, 0], S[:, 1], S[:, 2], 10.0
    goals["x"], goals["y"], goals["t"] = G[:, 0], G[:, 1], G[:, 2]
    ok, ln = h.shot(nodes, goals)
    keep = np.flatnonzero(ok == 0)[:n]
    nodes, goals = nodes[keep].copy(), goals[keep].copy()
    ts = []
    for _ in range(20):
        ok, ln = h.shot(nodes, goals); ms, l = h.kernel_ms(0); ts.append(1e3 * ms / max(l, 1))
    print(n, "failing shots only: us", round(float(np.median(ts)), 1), "min", round(float(np.min(ts)), 1), "ok", int(ok.sum()), "mean len", round(float(ln.mean()), 1))
    short = np.argsort(ln)[: n // 2]
    ts = []
    for _ in range(20):
        h.shot(nodes[short].copy(), goals[short].copy()); ms, l = h.kernel_ms(0); ts.append(1e3 * ms / max(l, 1))
    print(n // 2, "shortest half: us", round(float(np.median(ts)), 1), "mean len", round(float(ln[short].mean()), 1))
# tail: subsets of the 570 failing shots
per = []
for s0 in range(0, len(nodes), 19):
    sub_n, sub_g = nodes[s0:s0 + 19].copy(), goals[s0:s0 + 19].copy()
    ts = []
    for _ in range(8):
        h.shot(sub_n, sub_g); ms, l = h.kernel_ms(0); ts.append(1e3 * ms / max(l, 1))
    per.append(round(float(np.median(ts)), 1))
print("subsets of 19 failing shots, us:", sorted(per))
