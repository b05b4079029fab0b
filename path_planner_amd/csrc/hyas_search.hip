// hyas_search.hip -- batched Hybrid A* driver: the open/closed-set bookkeeping
// of HybridAStar::GetPath (src/hybrid_a_star.cpp:41-311) stays on the host,
// one state machine per query; every round each live query pops one node and
// all popped nodes go through ONE expansion-step kernel launch (successors +
// verdict + g + h + analytic shot).  Queries are independent, so the serial
// pop order of every single query is untouched and its expansion sequence is
// the one a one-query-at-a-time run produces.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include <emmintrin.h>
#include <sched.h>

#include "hyas_ctx.h"
#include "hyas_open_list.h"

namespace hyas
{
namespace
{

#define HYAS_MAX_SPEC 24   // most open-list entries of one query in one launch
#define HYAS_MAX_TREE 100  // most nodes of one query in one launch that do not exist yet
#define HYAS_MAX_NODES 124 // most nodes of one query in one launch
constexpr int SHARE_MIN = 7, SHARE_MAX = 64, SHARE_FIXED = 11; // a query's share of them: least, most, fixed widths from here up
#define HYAS_MAX_DEPTH 4 // most launches per round: each expands the lowest-f successors of the one before

struct SNode
{
    float x, y, t, g, h;
    int16_t prim;
    int16_t cand;     // which entry of the motion set produced the node, -1 = unknown (start node, adaptive modes)
    int32_t pred;     // arena index, -1 = none (shared_ptr chain in the reference)
    int32_t dup_next; // chain of the duplicate-check hash, -1 = end
    int32_t dup_prev; // >= 0 arena index, < 0: -(bucket + 2), head of its bucket
};

// idx -> {g, open, closed}: the three Node3D fields GetPath reads back from
// nodes3D[] (:101, :110, :264, :271).  Open addressing, grows at 50 % load.
struct CellMap
{
    struct Slot
    {
        int32_t key;
        float g;
        uint8_t open, closed;
    };
    HVec<Slot> slots;
    size_t used = 0;
    void init(size_t cap)
    {
        slots.assign(cap, Slot{-1, 999.f, 0, 0});
        used = 0;
    }
    // Successors of a node fall into the same or neighbouring cells: keeping neighbouring keys in neighbouring
    // slots makes the lookups of one expansion share cache lines.  Keys are dense, so runs rarely collide.
    // (Tried: the heading bins of one position side by side in windows of 128 slots, positions scattered over the
    // windows.  Slower: a search visits few positions with many headings each, and the windows fill up unevenly.)
    static size_t hash(int32_t k) { return (size_t)(uint32_t)k; }
    Slot *find(int32_t key)
    {
        size_t mask = slots.size() - 1, i = hash(key) & mask;
        while (slots[i].key != -1)
        {
            if (slots[i].key == key)
                return &slots[i];
            i = (i + 1) & mask;
        }
        return nullptr;
    }
    Slot &get(int32_t key)
    {
        if ((used + 1) * 2 > slots.size())
            grow();
        size_t mask = slots.size() - 1, i = hash(key) & mask;
        while (slots[i].key != -1 && slots[i].key != key)
            i = (i + 1) & mask;
        if (slots[i].key == -1)
        {
            slots[i] = Slot{key, 999.f, 0, 0}; // Node3D() default g, include/node3d.h:17
            ++used;
        }
        return slots[i];
    }
    void grow()
    {
        HVec<Slot> old;
        old.swap(slots);
        slots.assign(old.size() * 2, Slot{-1, 999.f, 0, 0});
        used = 0;
        for (const Slot &s : old)
            if (s.key != -1)
                get(s.key) = s;
    }
};

struct Query
{
    uint32_t qi = 0; // index within the chunk: row of the goals, the 2D cost maps and the successor filter
    hyas_goal goal;
    SNode start;
    uint32_t dist_start_goal_bits = 0; // bits of the float GetDistance(start, goal), for the adaptive steering selection
    HVec<SNode> arena;
    OpenList open;
    uint32_t seq = 0;
    CellMap cells;
    HVec<int32_t> dup_head; // spatial hash over the open list
    int iterations = 0, counter = 0, N = 0;
    int cur = -1, cur_idx = 0; // node being expanded this round
    // nodes handed to the kernel this round: spec[0] = cur, the others are the open list's next best entries,
    // expanded ahead of time in case one of them is popped right after cur
    // Behind them: nodes that do not exist yet.  Entry i (spec[i] = -1) is the successor that motion-set entry
    // ahead_cand[i] will produce from entry ahead_parent[i] of this list, formed on the host from the parent's pose
    // alone (successor pose and cost depend on nothing else); its answer is used if that successor, bit for bit,
    // is pushed and then popped during the round.
    int32_t spec[HYAS_MAX_NODES];
    int8_t ahead_parent[HYAS_MAX_NODES], ahead_cand[HYAS_MAX_NODES];
    // What this search has been doing: follow[a][b] counts how often, right after a node made by motion-set entry a,
    // the node popped next was that node's successor by entry b; seen[a] counts the nodes made by entry a.  Only
    // used to choose which successor to send ahead.
    uint16_t follow[8][8] = {}, seen[8] = {};
    // ... and how it has paid: nodes sent ahead for a parent made by entry a, how many of them were used, and how
    // many times sending was skipped since
    uint16_t ahead_sent[2][8] = {}, ahead_used[2][8] = {}, ahead_skipped[2][8] = {}; // [popped node / another][entry a]
    int8_t ahead_kind[HYAS_MAX_NODES], ahead_made_by[HYAS_MAX_NODES];
    int nspec = 0, n_open = 0; // nodes in the list; how many of them are open-list entries (they come first)
    int spec_w = 2; // how many of them to send next round: follows how far down the list the search has been landing
    bool shoot = false;
    bool done = false;
    int shot_from = -1; // >= 0: finished through an analytic shot from this node
    int end_node = -1, end_status = 0; // >= 0: finished at this node by the goal test or the iteration limit
    hyas_result res{};
    std::vector<int32_t> expanded; // (plain pages: with these lists in the huge-page pool as well most of its gain was lost, measured)
    std::vector<float> path;
};

// Changes of cell state a worker has seen this round, for the successor filter on the GPU (CellFilter, hyas_kernels.h).
// The sink lives on the worker's stack and writes into a buffer of its own: nothing here shares a cache line with
// another worker.  What does not fit is left out (a bound that stays too high only drops less).
struct UpdSink
{
    CellUpd *p;
    uint32_t n, cap;
    void add(uint32_t query, int32_t idx, int32_t bits)
    {
        if (n < cap)
            p[n++] = CellUpd{query, idx, bits, 0};
    }
};

// HYAS_TRACE: where the host time of taking successors goes, per thread, flushed by the worker loop
struct ConsumeProbe
{
    uint64_t dup = 0, cell = 0, push = 0, chain = 0, heads = 0, succ = 0, pushed = 0;
    uint64_t drop_closed = 0, drop_worse = 0, drop_tie = 0, drop_dup = 0; // successors dropped, by the test that dropped them
    uint64_t waited = 0, waits = 0; // cycles spent waiting for answer units that had not arrived, and how often
    uint64_t waited_shot = 0, waits_shot = 0; // ... of which for the shot's answer (the rest: successor count and records)
};
static thread_local ConsumeProbe t_probe;

struct Driver
{
    hyas_ctx *c;
    const DevMap &m;
    const hyas_params &P;
    float rad3; // ConvertDegToRad(3), src/hybrid_a_star.cpp:973
    const bool probe_on; // HYAS_TRACE: the worker threads count cycles and drops into t_probe
    Driver(hyas_ctx *ctx, bool probe) : c(ctx), m(ctx->dm), P(ctx->P), rad3(deg2rad(3)), probe_on(probe) {}

    // ---- duplicate check -------------------------------------------------
    // DuplicateCheck (:968-980) asks whether ANY open-list entry lies within
    // 0.05 m / 0.05 m / 3 deg of the candidate.  The open list is mirrored in
    // a hash of 0.1 m buckets; every entry of the 3x3 neighbourhood is put
    // through the reference's own comparison, so the answer is the linear
    // scan's answer.
    static int32_t bucket_coord(float v) { return (int32_t)std::floor((double)v * 10.0); }
    // buckets in tiles of 4 x 4 (0.4 m x 0.4 m, one cache line of bucket heads): a lookup's neighbourhood touches one or
    // two lines, and so do the lookups of the next expansions, which are a step of 0.2 m further on
    static size_t bucket_hash(int32_t bx, int32_t by, size_t mask)
    {
        const uint32_t tile = ((uint32_t)(by >> 2) * 2654435761u + (uint32_t)(bx >> 2)) * 0x9E3779B1u >> 4;
        return (size_t)(tile << 4 | (uint32_t)(by & 3) << 2 | (uint32_t)(bx & 3)) & mask;
    }
    void dup_link(Query &q, int32_t id)
    {
        SNode &n = q.arena[id];
        size_t b = bucket_hash(bucket_coord(n.x), bucket_coord(n.y), q.dup_head.size() - 1);
        n.dup_next = q.dup_head[b];
        n.dup_prev = -(int32_t)b - 2;
        if (n.dup_next >= 0)
            q.arena[n.dup_next].dup_prev = id;
        q.dup_head[b] = id;
    }
    void dup_insert(Query &q, int32_t id)
    {
        if (q.open.size() > q.dup_head.size())
        {
            // rebuild with four times the buckets; the open list holds the live set
            q.dup_head.assign(q.dup_head.size() * 4, -1);
            q.open.for_each([&](const HeapItem &e) {
                if (e.node != id)
                    dup_link(q, e.node);
            });
        }
        dup_link(q, id);
    }
    void dup_remove(Query &q, int32_t id)
    {
        SNode &n = q.arena[id];
        if (n.dup_prev >= 0)
            q.arena[n.dup_prev].dup_next = n.dup_next;
        else
            q.dup_head[(size_t)(-(n.dup_prev + 2))] = n.dup_next;
        if (n.dup_next >= 0)
            q.arena[n.dup_next].dup_prev = n.dup_prev;
        n.dup_next = -1, n.dup_prev = -1;
    }
    // Which neighbouring buckets can hold an entry within 0.05 (half a bucket) of v: the lower one only from the
    // bucket's lower half, the upper one only from its upper half.  The comparison below is made on the float
    // difference, which is within a rounding step of the true one: the margin of a thousandth of a bucket is orders
    // of magnitude wider.
    static void bucket_reach(float v, int32_t &b, int &lo, int &hi)
    {
        const double u = (double)v * 10.0, fl = std::floor(u), f = u - fl;
        b = (int32_t)fl;
        lo = f < 0.501 ? -1 : 0;
        hi = f > 0.499 ? 1 : 0;
    }
    bool dup_exists(const Query &q, float x, float y, float t) const
    {
        int32_t bx, by;
        int xlo, xhi, ylo, yhi;
        bucket_reach(x, bx, xlo, xhi), bucket_reach(y, by, ylo, yhi);
        const size_t mask = q.dup_head.size() - 1;
        // two neighbours may share a bucket; scanning it twice does not change the answer
        for (int dy = ylo; dy <= yhi; ++dy)
            for (int dx = xlo; dx <= xhi; ++dx)
            {
                const size_t b = bucket_hash(bx + dx, by + dy, mask);
                if (probe_on && q.dup_head[b] >= 0)
                    t_probe.heads += 1;
                for (int32_t id = q.dup_head[b]; id >= 0; id = q.arena[id].dup_next)
                {
                    const SNode &e = q.arena[id];
                    if (probe_on)
                        t_probe.chain += 1;
                    if ((double)hy_fabsf(e.x - x) < 0.05 && (double)hy_fabsf(e.y - y) < 0.05 && hy_fabsf(e.t - t) < rad3)
                        return true;
                }
            }
        return false;
    }

    // ---- open list -------------------------------------------------------
    void push(Query &q, const SNode &n, int32_t idx)
    {
        q.arena.push_back(n);
        int32_t id = (int32_t)q.arena.size() - 1;
        q.open.push(HeapItem{n.g + n.h, q.seq++, id, idx}); // getTotalCost(), include/node3d.h:36
        dup_insert(q, id);
    }
    void pop(Query &q)
    {
        dup_remove(q, q.open.top().node);
        q.open.pop();
    }

    // The open list's best entries, at most `want` of them, in pop order; entries whose cell is already closed are passed
    // over (advance() would discard them).  Pops nothing (entries may move from the heap into the sorted array).
    int next_best(Query &q, int32_t *out, int want) const
    {
        OpenList &o = q.open;
        int got = 0, looked = 0;
        const int look = std::min(OpenList::HOT, 2 * want + 2); // how far down the list to look for them
        if (o.nh < look)
            o.refill(look - o.nh);
        const size_t cmask = q.cells.slots.size() - 1;
        // (the lines the following steps read are requested as soon as their addresses are known)
        for (int i = o.nh - 1, n = 0; i >= 0 && n < look; --i, ++n)
            __builtin_prefetch(&q.cells.slots[q.cells.hash(o.hot[i].idx) & cmask]);
        for (int i = o.nh - 1; i >= 0 && got < want && looked < look; --i, ++looked)
        {
            CellMap::Slot *cell = q.cells.find(o.hot[i].idx);
            if (cell && cell->closed)
                continue;
            out[got++] = o.hot[i].node;
            __builtin_prefetch(&q.arena[o.hot[i].node]);
        }
        return got;
    }

    // ---- path assembly -----------------------------------------------------
    // TracePath (:813-828): back to the first node that "equals" the start
    // (Node3D::operator==, include/node3d.h:89-98) or to a null predecessor.
    void trace(const Query &q, int32_t id, std::vector<float> &xyt) const
    {
        std::vector<int32_t> chain;
        while (id >= 0)
        {
            chain.push_back(id);
            const SNode &n = q.arena[id];
            if (std::abs(n.x - q.start.x) < 1e-2 && std::abs(n.y - q.start.y) < 1e-2 && std::abs(n.t - q.start.t) < 1e-2)
                break;
            id = n.pred;
        }
        xyt.clear();
        for (size_t i = chain.size(); i-- > 0;)
        {
            const SNode &n = q.arena[chain[i]];
            xyt.push_back(n.x), xyt.push_back(n.y), xyt.push_back(n.t);
        }
    }
    // Utility::RemoveDuplicates (src/utility.cpp:2091-2109): of two poses closer
    // than 1e-3 the EARLIER one is erased.
    static void remove_duplicates(std::vector<float> &p)
    {
        if (p.empty())
            return;
        for (uint32_t i = 0; i < p.size() / 3 - 1; i++)
        {
            float d = normf(sub(f2(p[3 * i], p[3 * i + 1]), f2(p[3 * i + 3], p[3 * i + 4])));
            if ((double)d < 1e-3)
            {
                p.erase(p.begin() + 3 * i, p.begin() + 3 * i + 3);
                i = i - 1;
            }
        }
    }
    // The anchor poses of HybridAStar::ConvertToPiecewiseCubicBezierPath (src/hybrid_a_star.cpp:839-862): path nodes at
    // least one cell from their successor, up to and including the node the analytic curve starts from.  Returns the
    // sample count per curve, round(distance(start, that node) / resolution) (:873); the node stays (0, 0, 0) when the
    // loop never reaches its index.  Fewer than two anchors: no curve (the reference counts to size() - 1 unsigned).
    uint32_t piecewise_anchors(const Query &q, size_t analytical_expansion_index, std::vector<float> &anchors) const
    {
        anchors.clear();
        const std::vector<float> &p = q.path;
        const size_t n = p.size() / 3;
        float ex = 0, ey = 0;
        for (uint32_t index = 0; index < n - 1; ++index)
        {
            if (index == analytical_expansion_index)
            {
                ex = p[3 * index], ey = p[3 * index + 1];
                anchors.insert(anchors.end(), p.begin() + 3 * index, p.begin() + 3 * index + 3);
                break;
            }
            if (normf(sub(f2(p[3 * index], p[3 * index + 1]), f2(p[3 * index + 3], p[3 * index + 4]))) < m.res_f)
                continue;
            anchors.insert(anchors.end(), p.begin() + 3 * index, p.begin() + 3 * index + 3);
        }
        const int number_of_points = (int)hy_roundf(normf(sub(f2(q.start.x, q.start.y), f2(ex, ey))) / m.res_f);
        return (uint32_t)number_of_points;
    }
    // n_rd: how many RemoveDuplicates passes are still due (GetPath's own, unless the caller has run it in front of
    // ShortCut, and calculatePath's, :994)
    void finish(Query &q, int status, int32_t id, int n_rd) const
    {
        q.res.status = status;
        q.res.g_final = q.arena[id].g;
        for (int i = 0; i < n_rd; ++i)
            remove_duplicates(q.path);
        q.res.path_len = (int32_t)(q.path.size() / 3);
        double len = 0;
        for (size_t i = 1; i < q.path.size() / 3; ++i)
            len += (double)normf(sub(f2(q.path[3 * i], q.path[3 * i + 1]), f2(q.path[3 * i - 3], q.path[3 * i - 2])));
        q.res.path_length = (float)len;
        q.done = true;
    }

    // ---- one query, host side of one round --------------------------------
    // Pops until a node has to be expanded (:80-147).  Returns true when the
    // query wants a slot in this round's kernel launch.
    // upd (optional): changes of cell state for the successor filter on the GPU (CellFilter, hyas_kernels.h)
    static constexpr int32_t CLOSED_BITS = (int32_t)0xFF800000u; // -inf
    bool advance(Query &q, UpdSink *upd = nullptr)
    {
        while (!q.open.empty())
        {
            const int32_t pid = q.open.top().node;
            q.res.nodes_explored++;
            const SNode &n = q.arena[pid];
            const int32_t iPred = q.open.top().idx;
            q.iterations++;
            CellMap::Slot &cell = q.cells.get(iPred);
            if (cell.closed)
            {
                pop(q);
                continue;
            }
            cell.closed = 1, cell.open = 0;
            if (upd)
                upd->add(q.qi, iPred, CLOSED_BITS);
            pop(q);
            if (q.iterations > P.max_iterations)
            {
                q.end_node = pid, q.end_status = HYAS_STATUS_MAX_ITER, q.done = true; // path assembled at the end of the chunk
                return false;
            }
            // Utility::IsCloseEnough, src/utility.cpp:1132-1163
            {
                const float dist = normf(sub(f2(n.x, n.y), f2(q.goal.x, q.goal.y)));
                const float dist_range = P.goal_range * m.res_f;
                const float ang_range = (float)(2 * HY_PI / P.headings);
                if (dist < dist_range && hy_fabsf(rad_norm(n.t - q.goal.t)) <= ang_range)
                {
                    q.end_node = pid, q.end_status = HYAS_STATUS_GOAL, q.done = true;
                    return false;
                }
            }
            q.res.nodes_expanded++;
            q.expanded.push_back(iPred);
            q.shoot = false;
            if (P.analytical_expansion)
            {
                if (P.analytical_expansion_every_point)
                    q.shoot = true;
                else if (q.counter == q.N)
                {
                    q.N = (int)n.h;
                    q.counter = 0;
                    q.shoot = true;
                }
                else
                    q.counter++;
            }
            q.cur = pid, q.cur_idx = iPred;
            return true;
        }
        q.res.status = HYAS_STATUS_NO_PATH;
        q.done = true;
        return false;
    }

    // Takes the kernel's answer for the node popped by advance() (:243-305).  best_child (optional): the arena index
    // of the successor the kernel expanded one level down, -1 when that one was not pushed.  wanted (optional): the
    // arena indices of the pushed successors that come from given motion-set entries.
    struct Wanted // successors to report by motion-set entry: in cand[], out id[] (arena index, -1: not pushed)
    {
        int n = 0;
        int cand[HYAS_MAX_NODES];
        int32_t id[HYAS_MAX_NODES];
    };
    // count: successors emitted | candidates that received a verdict << 8; 0xFF: the node's pair list overflowed
    void consume(Query &q, uint8_t shot_ok, float shot_len, uint32_t count, const hyas_succ *succ, int32_t *best_child = nullptr,
                 Wanted *wanted = nullptr, UpdSink *upd = nullptr)
    {
        const int32_t pid = q.cur;
        if (best_child)
            *best_child = -1;
        if (q.shoot && shot_ok)
        {
            q.shot_from = pid;
            q.res.shot_length = shot_len;
            q.done = true; // path assembled once the curve samples are fetched
            return;
        }
        const SNode pred = q.arena[pid];
        if ((count & 0xFFu) == 0xFFu) // more distinct steering candidates than a node's list holds (adaptive mode)
        {
            q.res.status = HYAS_STATUS_OVERFLOW;
            q.done = true;
            return;
        }
        q.res.successors += (int32_t)(count >> 8 & 0xFFu); // motion-set entries that received a verdict
        const int n_rec = (int)(count & 0xFFu);
        // the lookups below miss the cache independently of each other: request all their lines first
        {
            const size_t cmask = q.cells.slots.size() - 1;
            for (int k = 0; k < n_rec; ++k)
            {
                const hyas_succ &s = succ[k];
                __builtin_prefetch(&q.cells.slots[q.cells.hash(s.idx) & cmask]);
            }
        }
        // the successor the kernel has sent on to the next, deeper launch of this round (marked pad = 1)
        int kmin = -1;
        if (best_child)
            for (int k = 0; k < n_rec; ++k)
                if (succ[k].pad & 1)
                    kmin = k;
        // The reference tests, in this order: duplicate of an open-list entry, cell closed, not better than the cell's
        // g, same cell as the predecessor and beyond the tie breaker (:255-283).  Each of them only drops the
        // successor, so their order does not matter: the cell tests come first here because they are cheaper and
        // drop most successors, and the duplicate check (a walk over neighbouring open-list entries) runs for the rest.
        for (int k = 0; k < n_rec; ++k)
        {
            const hyas_succ &s = succ[k];
            const uint64_t p0 = probe_on ? __builtin_ia32_rdtsc() : 0;
            const int32_t iSucc = s.idx;
            CellMap::Slot *seen = q.cells.find(iSucc);
            const uint64_t p1 = probe_on ? __builtin_ia32_rdtsc() : 0;
            if (probe_on)
                t_probe.succ += 1, t_probe.cell += p1 - p0;
            const float newG = s.g;
            if (seen && (seen->closed || (seen->open && !(newG < seen->g))))
            {
                if (probe_on)
                    (seen->closed ? t_probe.drop_closed : t_probe.drop_worse) += 1;
                continue;
            }
            const bool same_cell = q.cur_idx == iSucc;
            if (same_cell && s.g + s.h > pred.g + pred.h + P.tie_breaker)
            {
                if (probe_on)
                    t_probe.drop_tie += 1;
                continue;
            }
            const bool dup = dup_exists(q, s.x, s.y, s.t);
            const uint64_t p2 = probe_on ? __builtin_ia32_rdtsc() : 0;
            if (probe_on)
                t_probe.dup += p2 - p1, t_probe.drop_dup += dup ? 1 : 0;
            if (dup)
                continue;
            SNode n{s.x, s.y, s.t, s.g, s.h, (int16_t)s.prim, (int16_t)((int)(s.pad >> 8) - 1), same_cell ? pred.pred : pid, -1, -1};
            CellMap::Slot &cell = seen ? *seen : q.cells.get(iSucc);
            cell.g = n.g, cell.open = 1, cell.closed = 0;
            push(q, n, iSucc);
            if (upd)
            {
                int32_t bits;
                std::memcpy(&bits, &n.g, sizeof bits);
                upd->add(q.qi, iSucc, bits);
            }
            if (probe_on)
                t_probe.push += __builtin_ia32_rdtsc() - p2, t_probe.pushed += 1;
            if (k == kmin)
                *best_child = (int32_t)q.arena.size() - 1;
            if (wanted)
                for (int w = 0; w < wanted->n; ++w)
                    if (wanted->cand[w] == n.cand)
                        wanted->id[w] = (int32_t)q.arena.size() - 1;
        }
    }
};

// Minimal fork-join pool: the calling thread takes part, workers spin between
// rounds (a round lasts tens of microseconds).
// Worker threads stay on one CPU each (a query is always handled by the same thread: its heap, cell map and
// duplicate hash then stay in that core's cache), neighbours in a pool on neighbouring CPUs of the process's affinity
// mask; the ranks of a node take consecutive runs of CPUs.  Off when the mask holds fewer CPUs than the ranks'
// threads together, or with HYAS_PIN=0.
static int local_rank_env(const char *name)
{
    const char *e = std::getenv(name);
    return e ? std::max(0, std::atoi(e)) : 0;
}
static bool pinning_on(int threads_in_all)
{
    if (const char *e = std::getenv("HYAS_PIN"))
        if (std::atoi(e) == 0)
            return false;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0)
        return false;
    const int ranks = std::max(1, local_rank_env("LOCAL_WORLD_SIZE"));
    return CPU_COUNT(&set) >= ranks * threads_in_all;
}
static void pin_self(int index, int threads_in_all)
{
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0)
        return;
    std::vector<int> cpus;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set))
            cpus.push_back(c);
    if (cpus.empty())
        return;
    const size_t at = (size_t)local_rank_env("LOCAL_RANK") * (size_t)threads_in_all + (size_t)index;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpus[at % cpus.size()], &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
}

class Pool
{
  public:
    // first_cpu >= 0: thread i stays on CPU number first_cpu + i of the `pin_total` this rank uses
    explicit Pool(int nthreads, int first_cpu = -1, int pin_total = 0) : n_(std::max(1, nthreads)), first_cpu_(first_cpu)
    {
        for (int i = 1; i < n_; ++i)
            th_.emplace_back([this, i, pin_total] {
                if (first_cpu_ >= 0)
                    pin_self(first_cpu_ + i, pin_total);
                worker(i);
            });
    }
    int size() const { return n_; }
    ~Pool()
    {
        stop_.store(true, std::memory_order_release);
        publish();
        for (auto &t : th_)
            t.join();
    }
    void run(size_t count, const std::function<void(size_t, size_t)> &fn)
    {
        if (count == 0)
            return;
        if (n_ == 1 || count < 64)
        {
            fn(0, count);
            return;
        }
        fn_ = &fn;
        count_ = count;
        chunk_ = std::max<size_t>(4, count / (size_t)(n_ * 4));
        next_.store(0, std::memory_order_relaxed);
        pending_.store(n_ - 1, std::memory_order_relaxed);
        publish();
        work();
        while (pending_.load(std::memory_order_acquire) != 0)
            cpu_relax();
    }

    // fn(w) once on every thread, w = 0 (the caller) .. size()-1: the same w is always the same thread, so
    // data a thread touched last round is still in its cache
    void run_parts(const std::function<void(int)> &fn)
    {
        if (n_ == 1)
        {
            fn(0);
            return;
        }
        part_fn_ = &fn;
        pending_.store(n_ - 1, std::memory_order_relaxed);
        publish();
        fn(0);
        while (pending_.load(std::memory_order_acquire) != 0)
            cpu_relax();
        part_fn_ = nullptr;
    }

  private:
    static void cpu_relax() { __builtin_ia32_pause(); }
    // announces new work; wakes the workers that gave up spinning
    void publish()
    {
        gen_.fetch_add(1, std::memory_order_acq_rel);
        if (sleepers_.load(std::memory_order_acquire) != 0)
        {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
    }
    void work()
    {
        for (;;)
        {
            size_t b = next_.fetch_add(chunk_, std::memory_order_relaxed);
            if (b >= count_)
                break;
            (*fn_)(b, std::min(count_, b + chunk_));
        }
    }
    void worker(int self)
    {
        uint64_t seen = 0;
        for (;;)
        {
            // rounds are tens of microseconds apart: spin for about that long, then block, so that a machine
            // with fewer free CPUs than workers does not burn its time slices in spin loops
            int spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen)
            {
                if (++spins < 6000)
                {
                    cpu_relax();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire))
                return;
            if (part_fn_)
                (*part_fn_)(self);
            else
                work();
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    int n_, first_cpu_;
    std::vector<std::thread> th_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
    std::atomic<size_t> next_{0};
    std::atomic<int> pending_{0};
    std::atomic<int> sleepers_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    const std::function<void(int)> *part_fn_ = nullptr;
    size_t count_ = 0, chunk_ = 1;
};

int host_threads()
{
    if (const char *e = std::getenv("HYAS_THREADS"))
    {
        int v = std::atoi(e);
        if (v >= 1)
            return std::min(v, 256);
    }
    // the CPUs this process may run on: the workers spin between rounds, so more threads than CPUs stall each other
    unsigned hc = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        hc = (unsigned)CPU_COUNT(&set);
    // one process per GPU: the ranks of a node share its CPUs
    if (const char *e = std::getenv("LOCAL_WORLD_SIZE"))
    {
        int lw = std::atoi(e);
        if (lw > 1)
            hc = std::max(2u, hc / (unsigned)lw);
    }
    return (int)std::min<unsigned>(std::max<unsigned>(hc, 1), 16);
}

} // namespace

// Worker threads of the search driver, kept in the context between calls: one pool for the set-up and finishing
// steps of a chunk, and one for the rounds.  Idle workers block after a short spin.
struct SearchPools
{
    std::unique_ptr<Pool> main;
    Pool &main_pool(int threads)
    {
        if (!main || main->size() != std::max(1, threads))
            main.reset(new Pool(threads));
        return *main;
    }
    // the worker threads of the rounds, each on a CPU of its own where the affinity mask has enough of them
    std::unique_ptr<Pool> round;
    bool round_pinned = false;
    Pool &round_pool(int threads)
    {
        const bool pin = pinning_on(threads);
        if (!round || round->size() != std::max(1, threads) || round_pinned != pin)
            round.reset(), round.reset(new Pool(threads, pin ? 0 : -1, threads)), round_pinned = pin;
        return *round;
    }
};

void free_search_pools(hyas_ctx *c)
{
    delete c->pools;
    c->pools = nullptr;
}

int plan_batch_impl(hyas_ctx *c, const float *starts, const float *goals, size_t n, hyas_result *out)
{
    HYAS_HIP(c, hipSetDevice(c->device));
    const bool trace_on = std::getenv("HYAS_TRACE") != nullptr;
    Driver D(c, trace_on);
    const DevMap &m = c->dm;
    const int K = m.n_slots;
    reset_timers(c);
    c->paths.assign(n, std::vector<float>());
    c->expanded.assign(n, std::vector<int32_t>());
    if (n == 0)
        return HYAS_OK;
    // the round kernel is looked up here, once, before several worker threads launch it at the same moment
    HYAS_HIP(c, resolve_round_kernel(m, c->stats_on, c->stats_mode));
    if (!c->pools)
        c->pools = new SearchPools();
    Pool &pool = c->pools->main_pool(host_threads());

    // Queries are processed in chunks: each query of a chunk owns a 2D cost map in device memory, so a chunk takes
    // at most a quarter of the free device memory (8192 queries at most; HYAS_CHUNK overrides).
    const size_t twod = c->twod_size;
    // Successor filter on the GPU (CellFilter, hyas_kernels.h): one bound per query and cell (Node3D::setIdx: heading
    // bins 0..headings, rows 0..h, columns 0..w), kept up to date by the pops and pushes of every round.  HYAS_FILTER=0
    // turns it off; the answers are the same either way, the host then sees every traversable successor.
    const bool filter_on = !(std::getenv("HYAS_FILTER") && std::atoi(std::getenv("HYAS_FILTER")) == 0);
    const size_t cells3d = (size_t)(c->P.headings + 1) * m.w * m.h + m.w + 1;
    // queries per chunk: 4096 (the host state of the live searches -- open lists, cell maps, duplicate hashes -- is what
    // the worker threads miss the caches on; map_large: 5700-5800 queries/s against 5150-5200 with chunks of 8192)
    size_t chunk_cap = 4096;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
            (void)hipGetLastError(), free_b = (size_t)8 << 30;
        // two sets of cost maps when chunks are pipelined, one row of cell bounds
        const size_t per_query = 2 * twod * sizeof(float) + 4096 + (filter_on ? cells3d * sizeof(int32_t) : 0);
        const size_t reuse = c->d_twod_bytes + c->d_bound_bytes; // already held by this context
        chunk_cap = std::min<size_t>(chunk_cap, std::max<size_t>(16, (free_b / 4 + reuse) / per_query));
    }
    if (const char *e = std::getenv("HYAS_CHUNK"))
        chunk_cap = std::max<size_t>(1, (size_t)std::atoll(e));
    // Query groups in flight at once.  A process has four hardware queues and kernels that share one run one after the
    // other: chunks of up to 4096 queries form four groups (a queue each), larger ones (HYAS_CHUNK) eight.
    size_t n_groups = 0; // 0: by chunk size
    if (const char *e = std::getenv("HYAS_GROUPS"))
        n_groups = std::min<size_t>(HYAS_MAX_GROUPS, std::max<size_t>(1, (size_t)std::atoll(e)));
    // Nodes of one query expanded per round: its popped node plus the next best open-list entries, whose
    // results are used only if the search pops exactly that node next (it often does on plateaus of equal cost),
    // which saves the round trip.  The answers are those of one node per round.
    // A query sends fewer extra nodes while its speculation is not paying off: two beyond the deepest entry the
    // search used last round, one less (down to HYAS_SPEC_FLOOR) after a round that used none.  HYAS_SPEC_ADAPT=0
    // sends the full width every round.
    // The fewer queries are alive (the plugin's makePlan brings one), the more room a launch has for nodes of each:
    // about 2000 nodes per round over all live queries is what the GPU takes before the rounds get longer faster than
    // they get fewer (measured on maze batches of 1 to 1024 queries).  Per query: at most 64 nodes -- 16 open-list
    // entries and the popped node's neighbourhood (HYAS_TREE below), 27 instead of 47 ms for one maze query per call --
    // down to the 6 + 1 of large batches; below 11 per query the width follows each query's recent use, above it is
    // fixed.  The share is worked out every round from the queries still alive (a batch thins out as its searches
    // end: the benchmark batch spends most of its rounds with a quarter of its queries), unless one of HYAS_SPEC,
    // HYAS_SPEC_ADAPT, HYAS_AHEAD, HYAS_AHEAD_WIDTH, HYAS_TREE fixes the widths for the whole call; widths worked out
    // once from the size of the call (HYAS_LIVE_WIDTHS=0) take 5632 nodes per round, which allows for the thinning.
    const size_t ROUND_NODES = std::getenv("HYAS_ROUND_NODES") ? (size_t)std::max(64, std::atoi(std::getenv("HYAS_ROUND_NODES"))) : 2048;
    auto share_of = [](size_t queries, size_t nodes) {
        return (int)std::min<size_t>(SHARE_MAX, std::max<size_t>(SHARE_MIN, nodes / std::max<size_t>(queries, 1)));
    };
    const bool widths_follow_live = !(std::getenv("HYAS_SPEC") || std::getenv("HYAS_SPEC_ADAPT") || std::getenv("HYAS_AHEAD") ||
                                      std::getenv("HYAS_AHEAD_WIDTH") || std::getenv("HYAS_TREE") ||
                                      (std::getenv("HYAS_LIVE_WIDTHS") && std::atoi(std::getenv("HYAS_LIVE_WIDTHS")) == 0));
    const int per_query = share_of(n, 5632);
    const bool few_queries = per_query >= SHARE_FIXED;
    const bool spec_adapt = std::getenv("HYAS_SPEC_ADAPT") ? std::atoi(std::getenv("HYAS_SPEC_ADAPT")) != 0 : !few_queries;
    const int spec_floor = std::getenv("HYAS_SPEC_FLOOR") ? std::max(0, std::atoi(std::getenv("HYAS_SPEC_FLOOR"))) : 2;
    int spec_cfg = few_queries ? std::min(16, 4 + (per_query + 2) / 5) : 6;
    if (const char *e = std::getenv("HYAS_SPEC"))
        spec_cfg = std::min(HYAS_MAX_SPEC, std::max(1, std::atoi(e)));
    // Launches per round.  The second launch expands, for every node of the first, its lowest-f successor (which the
    // first launch left in device memory), the third does the same for the second: a search that keeps following
    // its newest node is answered for several pops from one round trip.  The host consumes strictly in pop order and
    // takes a deeper answer only when the node it pops is exactly the one that was expanded.
    int depth_cfg = 1;
    if (const char *e = std::getenv("HYAS_DEPTH"))
        depth_cfg = std::min(HYAS_MAX_DEPTH, std::max(1, std::atoi(e)));
    const size_t DEPTH = (size_t)depth_cfg;
    // Nodes sent ahead of their existence (HYAS_AHEAD, default 1 per query): a search mostly keeps the steering it
    // has, so the successor that the popped node's own motion-set entry will produce from it (HYAS_AHEAD=2: and that
    // successor's successor by the same entry) is formed on the host (pose and cost follow from the parent alone) and expanded
    // in the same launch.  Fixed motion set only: the adaptive modes pick their steering from maps the host does not hold.
    int ahead_cfg = few_queries ? 2 : 1;
    if (const char *e = std::getenv("HYAS_AHEAD"))
        ahead_cfg = std::min(4, std::max(0, std::atoi(e)));
    if (m.ad_mode != 0)
        ahead_cfg = 0;
    int ahead_width = 1; // how many of a query's nodes get such successors: the popped node, the open-list entries behind it
    if (const char *e = std::getenv("HYAS_AHEAD_WIDTH"))
        ahead_width = std::min(4, std::max(1, std::atoi(e)));
    ahead_width = std::min(ahead_width, ahead_cfg ? std::max(1, 4 / ahead_cfg) : 1);
    // With a handful of queries the launch has room for the popped node's whole neighbourhood (HYAS_TREE nodes, 0-100,
    // default: what the call's per-query share leaves beside the open-list entries, 48 at most): its successors by every motion-set entry, their successors, ... breadth first, each
    // node's own entry first.  Takes the place of the single successors above.
    int tree_cfg = few_queries ? per_query - spec_cfg : 0;
    if (const char *e = std::getenv("HYAS_TREE"))
        tree_cfg = std::min(HYAS_MAX_TREE, std::max(0, std::atoi(e)));
    if (m.ad_mode != 0)
        tree_cfg = 0;
    const int ahead_nodes = tree_cfg ? tree_cfg : ahead_cfg * ahead_width; // nodes of a query that do not exist yet, at most
    // Timing events around the launches of a round, behind hyas_last_kernel_ms(ctx, 0): off unless hyas_set_round_timing
    // (or HYAS_EVENT_EVERY) asks for them; every 2 k-th launch of a group then carries a pair, and the figure is scaled
    // up from the timed launches.  (Two event records lengthen a launch call by 2-3 us: 3-4 % of a maze batch.)
    const int64_t ev_every = std::getenv("HYAS_EVENT_EVERY") ? std::max(0, std::atoi(std::getenv("HYAS_EVENT_EVERY"))) : std::max(0, c->round_timing);
    std::vector<float> shot_buf;
    // a failed HIP call may leave kernels of other groups in flight that write into the staging block: wait for them
    auto drain = [&](int rc) {
        for (int g = 0; g < HYAS_MAX_GROUPS; ++g)
            if (c->round_stream[g])
                (void)hipStreamSynchronize(c->round_stream[g]);
        (void)hipGetLastError();
        return rc;
    };

    // ---- staging, laid out once for the largest chunk (one pinned block, one device block) ----
    const size_t cn_max = std::min(chunk_cap, n);
    // the next chunk is set up while this one's rounds run: two sets of set-up buffers (HYAS_PIPELINE=0: one after the other)
    const bool pipelined = n > chunk_cap && !(std::getenv("HYAS_PIPELINE") && std::atoi(std::getenv("HYAS_PIPELINE")) == 0);
    const int n_bufs = pipelined ? 2 : 1;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // results of one group of `cap` queries: successors, shot length, shot verdict, successor count
    // results of one group of `cap` node slots: one block of 2 + 2 K sixteen-byte units per node (RoundOut, hyas_kernels.h)
    const size_t out_units = 2 + 2 * (size_t)K;
    auto res_stride = [&](size_t cap) { return (cap * out_units * 16 + 255) & ~(size_t)255; };
    const int spec_k = spec_cfg; // open-list entries of a query in one launch, at most (widths fixed for the call)
    // nodes of a query in one launch, at most
    const size_t KS = widths_follow_live ? (size_t)SHARE_MAX : (size_t)(spec_k + ahead_nodes);
    // nodes of one group of a chunk in one launch, at most: with widths that follow the live queries, SHARE_MIN each
    // while the group is full, its part of ROUND_NODES (rounded up by one query's share) once it has thinned out
    auto group_slots = [&](size_t gcap_, size_t ng_) {
        return widths_follow_live ? std::max<size_t>(SHARE_MIN * gcap_, ROUND_NODES / ng_ + SHARE_MAX) : gcap_ * KS;
    };
    static_assert(HYAS_MAX_SPEC + HYAS_MAX_TREE <= HYAS_MAX_NODES && HYAS_MAX_NODES <= 127, "positions are kept in int8_t");
    // groups of a chunk hold ceil(cn / groups) queries each: at most cn + groups slots in all
    const size_t slots_max = cn_max + HYAS_MAX_GROUPS;
    // nodes of all groups together, at most (the sum of group_slots over any split into at most HYAS_MAX_GROUPS groups)
    const size_t nodes_max = widths_follow_live ? SHARE_MIN * slots_max + ROUND_NODES + (size_t)SHARE_MAX * HYAS_MAX_GROUPS : slots_max * KS;
    // (two sets of node slots and cell updates: a launch may still be reading its own while the next one's are written)
    const size_t o_in = take(2 * nodes_max * sizeof(StepIn));
    const size_t o_succ = take(DEPTH * (res_stride(nodes_max) + HYAS_MAX_GROUPS * 512));
    const size_t child_stride = (nodes_max * sizeof(StepIn) + 255) & ~(size_t)255; // device side only
    const size_t o_child = take((DEPTH - 1) * child_stride + 256);
    // changes of cell state for the successor filter, per node of the launch before: K pushes and the pop that follows
    const size_t upd_per_node = (size_t)K + 1, upd_max = filter_on ? nodes_max * upd_per_node : 0;
    const size_t o_upd = take(2 * upd_max * sizeof(CellUpd) + 256);
    // set-up staging, per buffer set: goals, poses {start, goal}, relative poses for the table lookup, their results
    struct PrepOff
    {
        size_t goal, poses, rel, cost, ok, qlist, startc;
    } po[2];
    for (int bset = 0; bset < n_bufs; ++bset)
    {
        po[bset].goal = take(cn_max * sizeof(hyas_goal)), po[bset].poses = take(cn_max * 6 * sizeof(float));
        po[bset].rel = take(cn_max * 3 * sizeof(float)), po[bset].cost = take(cn_max * sizeof(float));
        po[bset].ok = take(cn_max * 2 + 4), po[bset].qlist = take(cn_max * sizeof(uint32_t)), po[bset].startc = take(cn_max * sizeof(float));
    }
    int rc = ensure_scratch(c, off, off);
    if (rc != HYAS_OK)
        return rc;
    const size_t twod_stride = (cn_max * twod + 63) & ~(size_t)63; // floats per buffer set
    if ((rc = ensure_twod(c, (size_t)n_bufs * twod_stride * sizeof(float))) != HYAS_OK)
        return rc;
    char *const hs = (char *)c->h_scratch, *const ds = (char *)c->d_scratch;
    StepIn *const h_in = (StepIn *)(hs + o_in);
    char *hd = nullptr; // the pinned host block as the GPU addresses it
    HYAS_HIP(c, hipHostGetDevicePointer((void **)&hd, c->h_scratch, 0));
    // the popped nodes go into device memory the host can store into, where there is such memory, else into the
    // pinned host block; r_in: where the host writes them, k_in: where the kernel reads them
    const size_t nodes_bytes = (2 * nodes_max * sizeof(StepIn) + 255) & ~(size_t)255;
    if ((rc = ensure_nodes(c, nodes_bytes + 2 * upd_max * sizeof(CellUpd) + 256)) != HYAS_OK)
        return rc;
    StepIn *const r_in = c->d_nodes ? (StepIn *)c->d_nodes : h_in;
    const StepIn *const k_in = c->d_nodes ? (const StepIn *)c->d_nodes : (const StepIn *)(hd + o_in);
    // the filter's updates travel the same way, behind the nodes
    CellUpd *const r_upd = c->d_nodes ? (CellUpd *)((char *)c->d_nodes + nodes_bytes) : (CellUpd *)(hs + o_upd);
    const CellUpd *const k_upd = c->d_nodes ? (const CellUpd *)((char *)c->d_nodes + nodes_bytes) : (const CellUpd *)(hd + o_upd);
    if (filter_on && (rc = ensure_buffer(c, (void **)&c->d_bound, &c->d_bound_bytes, cn_max * cells3d * sizeof(int32_t))) != HYAS_OK)
        return rc;
    if (!c->prep_stream)
    {
        // lowest priority: streams of another priority get a hardware queue of their own (the round streams share the
        // process's four queues, and a flood that runs for a large part of a second must not sit in front of them)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
            (void)hipGetLastError(), least = 0;
        if (hipStreamCreateWithPriority(&c->prep_stream, hipStreamNonBlocking, least) != hipSuccess)
        {
            (void)hipGetLastError();
            HYAS_HIP(c, hipStreamCreate(&c->prep_stream));
        }
    }

    // floods that run beside another chunk's rounds: queries per compute unit at most (HYAS_FLOOD_BESIDE)
    const int flood_beside = std::getenv("HYAS_FLOOD_BESIDE") ? std::max(1, std::atoi(std::getenv("HYAS_FLOOD_BESIDE"))) : 2; // map_large: 1: 5460-5670 queries/s, 2: 5720-5890, 4: 5680-5870, 8: 5330-5440
    // ---- set-up of one chunk: pose checks, table lookups, 2D floods, start nodes ----
    struct Prep
    {
        size_t base = 0, cn = 0;
        int buf = 0, rc = HYAS_OK;
        HVec<Query> Q;
        double t_gpu = 0, t_all = 0;
    };
    auto prepare = [&](Prep &P, bool overlapped) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t cn = P.cn, base = P.base;
        const PrepOff &o = po[P.buf];
        hipStream_t st = c->prep_stream;
        P.Q.clear();
        P.Q.resize(cn);
        HVec<Query> &Q = P.Q;
        hyas_goal *h_goal = (hyas_goal *)(hs + o.goal);
        // makePlan pre-checks (src/hybrid_a_star_planner.cpp:126-155): start and goal poses through the pose test;
        // the start node's curve cost (:63) from the heuristic table: both on the GPU, in one trip
        float *poses = (float *)(hs + o.poses), *rel = (float *)(hs + o.rel), *cost = (float *)(hs + o.cost),
              *start_cost = (float *)(hs + o.startc);
        uint8_t *pose_ok = (uint8_t *)(hs + o.ok);
        uint32_t *qlist = (uint32_t *)(hs + o.qlist);
        for (size_t i = 0; i < cn; ++i)
        {
            const float *s = starts + 3 * (base + i), *g = goals + 3 * (base + i);
            // Utility::TypeConversion(PoseStamped -> Node3D), src/utility.cpp:7-13
            float *p = poses + 6 * i;
            p[0] = s[0], p[1] = s[1], p[2] = rad_0_2pi(s[2]);
            p[3] = g[0], p[4] = g[1], p[5] = rad_0_2pi(g[2]);
            h_goal[i] = hyas_goal{p[3], p[4], p[5]};
            rel[3 * i] = hy_fabsf(p[3] - p[0]), rel[3 * i + 1] = hy_fabsf(p[4] - p[1]), rel[3 * i + 2] = p[5] - p[2];
        }
        HYAS_HIP(c, hipMemcpyAsync(ds + o.goal, hs + o.goal, o.cost - o.goal, hipMemcpyHostToDevice, st)); // goals, poses, rel
        launch_collision(m, HYAS_MODE_POLYGON, (const float *)(ds + o.poses), nullptr, cn * 2, (uint8_t *)(ds + o.ok), st);
        launch_table_lookup(m, (const float *)(ds + o.rel), cn, (float *)(ds + o.cost), st);
        HYAS_HIP(c, hipGetLastError());
        HYAS_HIP(c, hipMemcpyAsync(hs + o.cost, ds + o.cost, o.qlist - o.cost, hipMemcpyDeviceToHost, st)); // costs, verdicts
        HYAS_HIP(c, hipStreamSynchronize(st));
        size_t n_flood = 0;
        for (size_t i = 0; i < cn; ++i)
        {
            Query &q = Q[i];
            const float *p = poses + 6 * i;
            q.qi = (uint32_t)i;
            q.goal = h_goal[i];
            q.start = SNode{p[0], p[1], p[2], 999.f, 0.f, 0, -1, -1, -1, -1}; // Node3D(): g = 999, prim = 0
            {
                // Utility::GetDistance(start_, goal_), src/collisiondetection.cpp:1389
                const float d = normf(sub(f2(p[0], p[1]), f2(q.goal.x, q.goal.y)));
                std::memcpy(&q.dist_start_goal_bits, &d, sizeof d);
            }
            uint32_t gx, gy;
            bool bad = !pose_ok[2 * i];
            if (!world_to_map(m, p[3], p[4], gx, gy))
                bad = true;
            else if (pose_ok[2 * i + 1] && c->charmap[gx + gy * m.w] > 252)
                bad = true;
            if (bad)
            {
                q.res.status = HYAS_STATUS_BAD_POSE;
                q.done = true;
                continue;
            }
            qlist[n_flood++] = (uint32_t)i;
        }
        // 2D cost map of every goal (src/hybrid_a_star.cpp:59), on the GPU, straight into the buffer the round kernel
        // reads; the host only needs the cost at each start cell.  Beside the rounds of another chunk the flood keeps to
        // one workgroup per compute unit, so that the round kernels still find LDS on every unit.
        if (n_flood)
        {
            HYAS_HIP(c, hipMemcpyAsync(ds + o.qlist, qlist, n_flood * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            int frc = flood_dev(c, (const float *)(ds + o.poses), (const uint32_t *)(ds + o.qlist), (uint32_t)n_flood,
                                c->d_twod + (size_t)P.buf * twod_stride, (float *)(ds + o.startc), st, overlapped ? flood_beside : 0);
            if (frc != HYAS_OK)
                return frc;
            HYAS_HIP(c, hipMemcpyAsync(start_cost, ds + o.startc, cn * sizeof(float), hipMemcpyDeviceToHost, st));
            HYAS_HIP(c, hipStreamSynchronize(st));
        }
        P.t_gpu = since(t0);
        // per-query set-up: search structures and the start node (:61-78)
        auto init = [&](size_t a, size_t b) {
            for (size_t k = a; k < b; ++k)
            {
                const size_t i = qlist[k];
                Query &q = Q[i];
                q.cells.init(1024);
                q.dup_head.assign(1 << 11, -1);
                q.arena.reserve(1024);
                const float two_d = start_cost[i] >= 0 ? start_cost[i] : 0.f;
                q.start.h = fmaxf_(cost[i], two_d);
                q.N = (int)q.start.h;
                const int idx = idx_3d(m, q.start.x, q.start.y, q.start.t);
                D.push(q, q.start, idx);
                CellMap::Slot &cell = q.cells.get(idx);
                cell.g = q.start.g, cell.open = 1, cell.closed = 0;
            }
        };
        if (overlapped) // the worker pools are busy with the other chunk's rounds
            init(0, n_flood);
        else
            pool.run(n_flood, init);
        P.t_all = since(t0);
        return HYAS_OK;
    };

    Prep cur, nxt;
    // Chunk sizes: the first chunk's set-up is the only one that nothing hides, so a pipelined batch starts with an
    // eighth of a chunk and doubles from chunk to chunk (a chunk's set-up beside the rounds of one half its size takes
    // about as long as those rounds); HYAS_FIRST_CHUNK sets the first size.
    size_t first_cn = std::min(chunk_cap, n);
    if (pipelined)
    {
        first_cn = std::max<size_t>(std::min<size_t>(1024, chunk_cap), chunk_cap / 8);
        if (const char *e = std::getenv("HYAS_FIRST_CHUNK"))
            first_cn = std::min(chunk_cap, std::max<size_t>(1, (size_t)std::atoll(e)));
    }
    cur.base = 0, cur.cn = first_cn, cur.buf = 0;
    if ((rc = prepare(cur, false)) != HYAS_OK)
        return rc;
    for (;;)
    {
        const auto wall0 = std::chrono::steady_clock::now();
        const size_t cn = cur.cn, base = cur.base;
        HVec<Query> &Q = cur.Q;
        const size_t o_goal = po[cur.buf].goal;
        const float *const d_twod_cur = c->d_twod + (size_t)cur.buf * twod_stride;
        // the next chunk's set-up runs on a thread of its own (own stream, own buffers) while this chunk's rounds run
        const bool has_next = base + cn < n;
        std::thread prep_thread;
        struct Joiner
        {
            std::thread &t;
            ~Joiner()
            {
                if (t.joinable())
                    t.join();
            }
        } joiner{prep_thread};
        if (has_next)
        {
            nxt = Prep();
            nxt.base = base + cn, nxt.cn = std::min(std::min(chunk_cap, 2 * cn), n - nxt.base), nxt.buf = pipelined ? 1 - cur.buf : 0;
        }
        if (has_next && pipelined)
        {
            prep_thread = std::thread([&] {
                if (hipSetDevice(c->device) != hipSuccess)
                {
                    nxt.rc = HYAS_ENODEV;
                    return;
                }
                nxt.rc = prepare(nxt, true);
            });
        }
        const size_t want_groups = n_groups ? n_groups : (cn <= 4096 ? 4 : 8);
        const int ng = (int)std::max<size_t>(1, std::min<size_t>(want_groups, cn / 32 + 1));
        const size_t gcap = (cn + ng - 1) / ng; // queries per group
        const size_t gslots = group_slots(gcap, (size_t)ng); // node slots per group
        // worker threads of the rounds: one per eight queries of the chunk, at most the host threads of the call
        const int n_workers = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size(), cn / 8 + 1));
        const double wall_flood = cur.t_gpu;

        // rounds.  The chunk's queries form groups that advance independently: every group has its own stream and
        // slice of the staging block, and nothing is shared between groups but the map.  A round of a group is a host
        // part (push the successors of the launch that has come back, pop the next nodes, stage them) and a launch.
        // A host part is cut into pieces of a few queries.  Every worker thread has a home range of every group's pieces (the
        // same queries every round: their heaps, cell maps and duplicate hashes stay in that core's cache) and takes
        // those first, then pieces of the other workers' ranges that are still waiting, so that a host part is finished
        // by whoever is free and its next launch leaves as early as possible.  Host parts are released in the order in
        // which the launches come back and every worker goes through them in that order; no worker waits for another: the
        // one that finishes a host part's last piece launches the group and releases its next host part at once: nobody
        // waits for the launch to finish, a worker waits for the answer it needs next (see `unit`).
        enum : int { G_HOST = 0, G_FINISHED = 2 };
        struct Group
        {
            // the host part being worked on: result blocks, widths, who has still to do a share of it
            const char *out[HYAS_MAX_DEPTH] = {}; // results of the launch(es) before it, one block of out_units units per node
            uint32_t read_tag = 0;                // ... and the tag their units carry
            int r_spec_k = 0, r_tree = 0, r_ahead = 0;
            size_t r_nodes = 0; // node slots of every query in the launch that follows this host part
            bool r_adapt = false, have_results = false;
            std::atomic<uint32_t> pieces_left{0}; // pieces of this host part that are not finished
            uint32_t part_no = 0;
            // worker w's home range of pieces: {host part (24 bits), end (20), next (20)} in one word, claimed by compare-and-swap
            struct alignas(64) Cursor
            {
                std::atomic<uint64_t> v{0};
            };
            std::unique_ptr<Cursor[]> cur;
            std::atomic<int> state{G_HOST};
            std::atomic<uint64_t> host_ns{0}; // worker time spent in host parts (HYAS_TRACE)
            // HYAS_TRACE, per host part: when it was released, when the first worker began and the last one ended; sums over the parts
            uint64_t rel_ns = 0;
            std::atomic<uint64_t> first_ns{0}, last_ns{0};
            uint64_t sum_first = 0, sum_last = 0, parts = 0;
            std::vector<uint32_t> live, keep;
            std::vector<size_t> bounds;
            size_t s0 = 0, ns = 0; // first query, nodes in flight
            size_t base = 0;       // first node slot of the group in the staging blocks
            // successor filter: the cell changes each worker has seen this round, and their place in the launch
            std::vector<std::unique_ptr<CellUpd[]>> upd; // one buffer of upd_cap entries per worker
            std::atomic<uint32_t> upd_claim{0};
            size_t upd_base = 0, upd_cap = 0, n_upd = 0;
            double t_launch = 0, k_ms = 0, t_sub = 0, t_turn = 0, t_idle = 0, t_done = 0;
            int64_t launches = 0, rounds = 0, timed_launches = 0;
            bool timed = false; // the round in flight carries the two timing events
            int rc = HYAS_OK;
        } G[HYAS_MAX_GROUPS];
        for (int gi = 0; gi < ng; ++gi)
        {
            G[gi].s0 = (size_t)gi * gcap;
            G[gi].base = (size_t)gi * gslots;
            G[gi].upd_base = (size_t)gi * gslots * upd_per_node, G[gi].upd_cap = filter_on ? gslots * upd_per_node : 0;
            for (size_t i = G[gi].s0; i < std::min(cn, G[gi].s0 + gcap); ++i)
                G[gi].live.push_back((uint32_t)i);
        }
        std::vector<uint8_t> wants(cn);
        HVec<StepIn> stage(cn * KS); // next round's input of every query, filled by the worker that popped it
        HVec<uint32_t> slot0(cn);    // where a query's nodes lie in its group's launch
        const bool P_analytic = c->P.analytical_expansion != 0;
        std::atomic<uint64_t> miss_child{0}, miss_other{0};
        // what every worker counts for itself (summed after the rounds): answers used ahead of time, and with HYAS_TRACE
        // the nodes sent, host time by part in timestamp-counter cycles and the figures of taking successors
        struct alignas(64) Tally
        {
            uint64_t hits = 0, deep = 0, ahead = 0, ahead_sent = 0, open_sent = 0;
            uint64_t c_consume = 0, c_advance = 0, c_next = 0, c_copy = 0;
            ConsumeProbe probe;
        };
        std::vector<Tally> tally((size_t)n_workers);
        std::vector<std::atomic<uint64_t>> probe_hist(128);
        auto tick = [trace_on]() -> uint64_t { return trace_on ? __builtin_ia32_rdtsc() : 0; };
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        // where a group's results lie in the staging block
        auto out_offset = [&](const Group &g, size_t lvl) { return o_succ + ((&g - G) * DEPTH + lvl) * res_stride(gslots); };
        std::atomic<bool> abort_all{false};
        // One unit of a node's result block, once it carries the round's tag (in the low 23 bits of its last word shifted
        // down by `shift`).  The kernel stores a unit in one 16-byte store and the load here is one 16-byte load, so a
        // unit is seen whole or not at all.  False when the call is being abandoned.
        auto stuck = [&]() {
            for (int gi = 0; gi < ng; ++gi)
            {
                const hipError_t e = hipStreamQuery(c->round_stream[gi]);
                if (e != hipSuccess && e != hipErrorNotReady)
                {
                    G[gi].rc = HYAS_ENODEV;
                    abort_all.store(true, std::memory_order_release);
                }
                (void)hipGetLastError();
            }
        };
        auto unit = [&](const char *p, uint32_t tag, int shift, uint32_t w[4], bool is_shot = false) -> bool {
            uint64_t wait0 = 0;
            for (uint64_t spins = 0;; ++spins)
            {
                _mm_storeu_si128((__m128i *)w, _mm_load_si128((const __m128i *)p));
                if (((w[3] >> shift) & ROUND_TAG_MASK) == tag)
                {
                    if (spins && trace_on)
                    {
                        const uint64_t dt = __builtin_ia32_rdtsc() - wait0;
                        t_probe.waited += dt, t_probe.waits += 1;
                        if (is_shot)
                            t_probe.waited_shot += dt, t_probe.waits_shot += 1;
                    }
                    return true;
                }
                if (!spins && trace_on)
                    wait0 = __builtin_ia32_rdtsc();
                __builtin_ia32_pause();
                if ((spins & 0xFFFFF) == 0xFFFFF) // a launch that does not come back: ask the runtime what happened to it
                {
                    if (abort_all.load(std::memory_order_acquire))
                        return false;
                    if ((spins >> 20) % 64 == 63)
                        stuck();
                }
            }
        };
        // host parts in the order of their release: entry k names the group of the k-th; a group has at most one host part
        // outstanding, so no more than ng entries are ever ahead of the slowest worker
        constexpr uint64_t ORDER_CAP = 4 * HYAS_MAX_GROUPS;
        std::atomic<uint64_t> order[ORDER_CAP]; // group | host part << 8
        std::atomic<uint64_t> released{0};
        std::atomic<int> n_finished{0};
        const size_t PIECE = std::getenv("HYAS_PIECE") ? (size_t)std::max(1, std::min(4096, std::atoi(std::getenv("HYAS_PIECE")))) : 2; // queries per piece
        // Releases the next host part of a group to the workers: push the successors of the launch that has come back
        // (if any), pop the next node of every query.  Called by the thread that saw the launch finish.
        const int nw = n_workers;
        auto begin_host_part = [&](Group &g, bool have_results) {
            for (size_t l = 0; l < DEPTH; ++l)
                g.out[l] = hs + out_offset(g, l);
            g.read_tag = g.part_no; // the launch before this host part carried the number of the one before it
            // equal shares of the group's live queries, in index order: a query stays with the same worker from
            // round to round (its heap, cell map and duplicate hash stay in that core's cache) and only moves to a
            // neighbour when queries in front of it finish
            const size_t n_pieces = (g.live.size() + PIECE - 1) / PIECE;
            g.part_no = g.part_no == ROUND_TAG_MASK ? 1u : g.part_no + 1; // never 0: that is what a cleared unit carries
            if (!g.cur)
                g.cur.reset(new Group::Cursor[nw]);
            g.upd_claim.store(0, std::memory_order_relaxed);
            while (filter_on && (int)g.upd.size() < nw)
                g.upd.emplace_back(new CellUpd[g.upd_cap + 1]);
            // this round's widths: the call's, or what the queries still alive leave room for (every group takes the
            // call to be as thinned out as itself)
            g.r_spec_k = spec_k, g.r_tree = tree_cfg, g.r_ahead = ahead_cfg, g.r_adapt = spec_adapt;
            if (widths_follow_live)
            {
                const int share = share_of(g.live.size() * (size_t)ng, ROUND_NODES);
                const bool fixed = share >= SHARE_FIXED;
                g.r_adapt = !fixed;
                g.r_spec_k = fixed ? std::min(16, 4 + (share + 2) / 5) : 6;
                g.r_tree = fixed && m.ad_mode == 0 ? share - g.r_spec_k : 0;
                g.r_ahead = m.ad_mode == 0 ? 1 : 0;
            }
            g.r_nodes = std::min<size_t>(KS, (size_t)g.r_spec_k + (size_t)(g.r_tree ? g.r_tree : g.r_ahead * ahead_width));
            g.have_results = have_results;
            if (trace_on)
            {
                if (g.rel_ns && g.last_ns.load())
                    g.sum_first += g.first_ns.load() - g.rel_ns, g.sum_last += g.last_ns.load() - g.rel_ns, g.parts += 1;
                g.rel_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
                g.first_ns.store(~0ull), g.last_ns.store(0);
            }
            if (n_pieces == 0) // (only a group none of whose queries passed the pose checks)
            {
                g.state.store(G_FINISHED, std::memory_order_release);
                n_finished.fetch_add(1, std::memory_order_acq_rel);
                return;
            }
            g.pieces_left.store((uint32_t)n_pieces, std::memory_order_relaxed);
            for (int w = 0; w < nw; ++w)
                g.cur[w].v.store((uint64_t)g.part_no << 40 | (uint64_t)(n_pieces * (size_t)(w + 1) / (size_t)nw) << 20 | (uint64_t)(n_pieces * (size_t)w / (size_t)nw),
                                 std::memory_order_release);
            // (one releasing thread at a time: the caller holds the poll lock, or the workers have not started yet)
            const uint64_t at = released.load(std::memory_order_relaxed);
            order[at % ORDER_CAP].store((uint64_t)(&g - G) | (uint64_t)g.part_no << 8, std::memory_order_relaxed);
            released.store(at + 1, std::memory_order_release);
        };
        // queries live[k0 .. k1) of a group's host part, done by worker w
        auto host_piece = [&](Group &g, int w, size_t k0, size_t k1, UpdSink *upd) {
            {
                const bool have_results = g.have_results;
                const int r_spec_k = g.r_spec_k, r_tree = g.r_tree, r_ahead = g.r_ahead;
                const bool r_adapt = g.r_adapt;
                const uint32_t read_tag = g.read_tag;
                uint64_t hits = 0, deep = 0, ahead = 0, n_ahead_sent = 0, n_open_sent = 0, c_consume = 0, c_advance = 0, c_next = 0;
                const size_t r_nodes = g.r_nodes;
                // Nearly every line a query touches is cold (a worker sees it once per round, between dozens of other
                // queries): what can be known ahead is requested ahead.  First the queries' own records ...
                for (size_t k = k0; k < k1; ++k)
                {
                    const char *qp = (const char *)&Q[g.live[k]];
                    __builtin_prefetch(qp), __builtin_prefetch(qp + 64), __builtin_prefetch(qp + 128), __builtin_prefetch(qp + 192);
                }
                // ... then, one query ahead of the one being worked on: the head of every answer block it may read, the
                // top of its open list and its node of this round
                auto warm = [&](size_t k) {
                    const Query &nq = Q[g.live[k]];
                    if (have_results)
                    {
                        const char *blk = g.out[0] + (size_t)slot0[g.live[k]] * out_units * 16;
                        for (int i = 0; i < nq.nspec; ++i)
                            __builtin_prefetch(blk + (size_t)i * out_units * 16);
                        if (nq.cur >= 0)
                            __builtin_prefetch(&nq.arena[nq.cur]);
                    }
                    if (!nq.open.empty())
                        __builtin_prefetch(&nq.open.hot[nq.open.nh - 1]), __builtin_prefetch(&nq.open.hot[std::max(0, nq.open.nh - 5)]);
                };
                if (k0 < k1)
                    warm(k0);
                for (size_t k = k0; k < k1; ++k)
                {
                    if (k + 1 < k1)
                        warm(k + 1);
                    Query &q = Q[g.live[k]];
                    const uint32_t qi = g.live[k];
                    bool want;
                    if (have_results)
                    {
                        // the nodes this round holds answers for: the open-list entries sent along (first launch) and,
                        // as they come into being, the successors that the deeper launches expanded
                        constexpr int ANS_CAP = 2 * HYAS_MAX_NODES * HYAS_MAX_DEPTH;
                        int32_t ans_id[ANS_CAP];
                        uint8_t ans_lvl[ANS_CAP], ans_at[ANS_CAP];
                        int n_ans = 0;
                        for (int i = 1; i < q.nspec; ++i)
                            if (q.spec[i] >= 0)
                                ans_id[n_ans] = q.spec[i], ans_lvl[n_ans] = 0, ans_at[n_ans] = (uint8_t)i, ++n_ans;
                        size_t lvl = 0;
                        int at = 0; // which of the query's nodes of launch `lvl` is being consumed
                        int deepest = 0;
                        for (;;)
                        {
                            const size_t sl = slot0[qi] + at;
                            int32_t best_child = -1;
                            const uint64_t c0 = tick();
                            // which of the nodes sent ahead of their existence are successors of this one
                            Driver::Wanted wanted;
                            int wanted_at[HYAS_MAX_NODES];
                            if (lvl == 0)
                                for (int i = 1; i < q.nspec; ++i)
                                    if (q.spec[i] < 0 && q.ahead_parent[i] == at)
                                        wanted.cand[wanted.n] = q.ahead_cand[i], wanted.id[wanted.n] = -1, wanted_at[wanted.n] = i, ++wanted.n;
                            // the node's answer, unit by unit as it arrives: the successors' count, the successors, and
                            // the shot's verdict where the search asks for it
                            const char *blk = g.out[lvl] + sl * out_units * 16;
                            uint32_t w[4], wb[4];
                            hyas_succ got[HYAS_ADAPTIVE_SLOTS];
                            if (!unit(blk + 16, read_tag, 0, w))
                                return;
                            const uint32_t cnt = w[0];
                            const int n_got = (cnt & 0xFFu) == 0xFFu ? 0 : (int)std::min<uint32_t>(cnt & 0xFFu, HYAS_ADAPTIVE_SLOTS);
                            for (int j = 0; j < n_got; ++j)
                            {
                                if (!unit(blk + 32 + 32 * j, read_tag, 9, w) || !unit(blk + 48 + 32 * j, read_tag, 0, wb))
                                    return;
                                hyas_succ &sj = got[j];
                                std::memcpy(&sj.x, &w[0], 12), std::memcpy(&sj.g, &wb[0], 8);
                                sj.idx = (int32_t)wb[2], sj.prim = (int8_t)(w[3] & 7u), sj.free = 1;
                                sj.pad = (uint16_t)(((w[3] >> 3) & 31u) << 8 | ((w[3] >> 8) & 1u));
                            }
                            uint8_t shot_ok = 0;
                            float shot_len = 0;
                            if (q.shoot)
                            {
                                if (!unit(blk, read_tag, 0, w, true))
                                    return;
                                shot_ok = (uint8_t)w[1];
                                std::memcpy(&shot_len, &w[0], 4);
                            }
                            D.consume(q, shot_ok, shot_len, cnt, got, lvl + 1 < DEPTH ? &best_child : nullptr, wanted.n ? &wanted : nullptr, upd);
                            if (best_child >= 0 && n_ans < ANS_CAP)
                                ans_id[n_ans] = best_child, ans_lvl[n_ans] = (uint8_t)(lvl + 1), ans_at[n_ans] = (uint8_t)at, ++n_ans;
                            for (int w = 0; w < wanted.n; ++w)
                            {
                                if (wanted.id[w] < 0 || n_ans >= ANS_CAP)
                                    continue;
                                // the node as the kernel produced it against the node as it was sent: the answer holds
                                // for exactly these bits
                                const SNode &nd = q.arena[wanted.id[w]];
                                const StepIn &sent = stage[qi * KS + wanted_at[w]];
                                if (std::memcmp(&nd.x, &sent.x, 4 * sizeof(float)) != 0 || (int32_t)nd.prim != sent.prim)
                                    continue;
                                q.spec[wanted_at[w]] = wanted.id[w];
                                ans_id[n_ans] = wanted.id[w], ans_lvl[n_ans] = 0, ans_at[n_ans] = (uint8_t)wanted_at[w], ++n_ans;
                            }
                            const uint64_t c1 = tick();
                            const int32_t consumed = q.cur;
                            want = !q.done && D.advance(q, upd);
                            c_consume += c1 - c0, c_advance += tick() - c1;
                            if (!want)
                                break;
                            if (ahead_cfg)
                            {
                                const int a = q.arena[consumed].cand, b = q.arena[q.cur].cand;
                                if (a >= 0 && a < 8)
                                {
                                    if (++q.seen[a] >= 60000)
                                    {
                                        q.seen[a] /= 2;
                                        for (int e = 0; e < 8; ++e)
                                            q.follow[a][e] /= 2;
                                    }
                                    if (b >= 0 && b < 8 && q.arena[q.cur].pred == consumed)
                                        ++q.follow[a][b];
                                }
                            }
                            // the node popped now: does the round hold its answer?
                            int j = -1;
                            for (int i = 0; i < n_ans; ++i)
                                if (ans_id[i] == q.cur)
                                    j = i;
                            if (j < 0)
                            {
                                if (trace_on)
                                {
                                    // why the round ends: the popped node is a successor pushed during this round, or
                                    // an older open-list entry beyond the ones sent along
                                    bool child = false;
                                    for (int i = 0; i < q.nspec; ++i)
                                        if (q.spec[i] >= 0 && q.arena[q.cur].pred == q.spec[i])
                                        {
                                            child = true;
                                            const int pp = q.arena[q.spec[i]].cand, cp = q.arena[q.cur].cand;
                                            probe_hist[(i == 0 ? 0 : 1) * 64 + (pp & 7) * 8 + (cp & 7)].fetch_add(1, std::memory_order_relaxed);
                                        }
                                    (child ? miss_child : miss_other).fetch_add(1, std::memory_order_relaxed);
                                }
                                break; // no: it goes to the GPU
                            }
                            lvl = ans_lvl[j], at = ans_at[j];
                            if (lvl == 0 && at < q.n_open)
                                deepest = std::max(deepest, at), ++hits;
                            else if (lvl == 0)
                            {
                                ++ahead;
                                ++q.ahead_used[q.ahead_kind[at]][q.ahead_made_by[at]];
                            }
                            else
                                ++deep;
                        }
                        // width of the next round: two beyond the deepest entry that was used, one less when none was
                        if (r_adapt)
                            q.spec_w = deepest ? std::min(r_spec_k - 1, deepest + 2) : std::max(spec_floor, q.spec_w - 1);
                    }
                    else
                        want = !q.done && D.advance(q, upd);
                    wants[qi] = want ? 1 : 0;
                    const uint64_t c2 = tick();
                    if (want)
                    {
                        q.spec[0] = q.cur;
                        const int extra = r_adapt ? std::min(q.spec_w, r_spec_k - 1) : r_spec_k - 1;
                        q.nspec = 1 + (extra > 0 ? D.next_best(q, q.spec + 1, extra) : 0);
                        q.n_open = q.nspec;
                        for (int i = 0; i < q.nspec; ++i)
                        {
                            const SNode &nd = q.arena[q.spec[i]];
                            // a node expanded ahead of time always gets its shot evaluated: whether the search will
                            // ask for it is only known when the node is popped
                            const uint32_t shoot = i == 0 ? (q.shoot ? 1u : 0u) : (P_analytic ? 1u : 0u);
                            stage[qi * KS + i] = StepIn{nd.x, nd.y, nd.t, nd.g, nd.prim, qi, shoot, q.dist_start_goal_bits};
                        }
                        // nodes that do not exist yet: for the popped node (and, HYAS_AHEAD_WIDTH > 1, the open-list
                        // entries behind it) the successor by the node's own motion-set entry, that one's successor ...
                        if (r_tree)
                        {
                            // breadth first from the popped node; a parent's own motion-set entry comes first
                            int8_t queue[HYAS_MAX_NODES];
                            int head = 0, tail = 0;
                            queue[tail++] = 0;
                            const int limit = q.n_open + r_tree;
                            while (head < tail && q.nspec < limit)
                            {
                                const int parent = queue[head++];
                                const StepIn from = stage[qi * KS + parent];
                                const int own = parent == 0 ? (int)q.arena[q.cur].cand : (int)q.ahead_cand[parent];
                                const float dist = normf(sub(f2(from.x, from.y), f2(q.goal.x, q.goal.y)));
                                for (int j = 0; j < m.n_succ && q.nspec < limit; ++j)
                                {
                                    // order: own entry, then the others as the motion set lists them
                                    const int entry = own >= 0 && own < m.n_succ ? (j == 0 ? own : (j <= own ? j - 1 : j)) : j;
                                    float sx, sy, st;
                                    int prim;
                                    if (!successor_pose(m, from.x, from.y, from.t, dist, entry, sx, sy, st, prim))
                                        continue;
                                    const float step = normf(sub(f2(sx, sy), f2(from.x, from.y)));
                                    const int i = q.nspec++;
                                    q.spec[i] = -1, q.ahead_parent[i] = (int8_t)parent, q.ahead_cand[i] = (int8_t)entry;
                                    q.ahead_kind[i] = 0, q.ahead_made_by[i] = (int8_t)(own >= 0 && own < 8 ? own : 0);
                                    stage[qi * KS + i] = StepIn{sx, sy, st, cost_so_far(m, from.g, step, prim, from.prim), prim, qi, P_analytic ? 1u : 0u,
                                                                q.dist_start_goal_bits};
                                    queue[tail++] = (int8_t)i;
                                }
                            }
                        }
                        else
                        for (int root = 0; root < std::min(ahead_width, q.n_open); ++root)
                        {
                            StepIn from = stage[qi * KS + root];
                            int parent = root, made_by = q.arena[q.spec[root]].cand;
                            const int kind = root ? 1 : 0; // the figures of the popped node and of the others differ
                            for (int a = 0; a < r_ahead && made_by >= 0 && made_by < 8 && made_by < m.n_succ; ++a)
                            {
                                // the entry this search has followed most often after a node made by `made_by`; the
                                // same entry again until there is something to go by
                                int entry = made_by;
                                if (q.seen[made_by] >= 16)
                                    for (int e = 0; e < m.n_succ && e < 8; ++e)
                                        if (q.follow[made_by][e] > 2 * q.follow[made_by][entry])
                                            entry = e;
                                // not worth a place in the launch when fewer than one in eight has been used; every
                                // sixteenth time it is sent all the same, so that the figures keep up with the search
                                if (q.ahead_sent[kind][made_by] >= 32 &&
                                    (uint32_t)q.ahead_used[kind][made_by] * 8u < (uint32_t)q.ahead_sent[kind][made_by] &&
                                    (++q.ahead_skipped[kind][made_by] & 15) != 0)
                                    break;
                                if (++q.ahead_sent[kind][made_by] >= 60000)
                                    q.ahead_sent[kind][made_by] /= 2, q.ahead_used[kind][made_by] /= 2;
                                const float dist = normf(sub(f2(from.x, from.y), f2(q.goal.x, q.goal.y)));
                                float sx, sy, st;
                                int prim;
                                if (!successor_pose(m, from.x, from.y, from.t, dist, entry, sx, sy, st, prim))
                                    break;
                                const float step = normf(sub(f2(sx, sy), f2(from.x, from.y)));
                                const StepIn next{sx, sy, st, cost_so_far(m, from.g, step, prim, from.prim), prim, qi, P_analytic ? 1u : 0u,
                                                  q.dist_start_goal_bits};
                                const int i = q.nspec++;
                                q.spec[i] = -1, q.ahead_parent[i] = (int8_t)parent, q.ahead_cand[i] = (int8_t)entry;
                                q.ahead_kind[i] = (int8_t)kind, q.ahead_made_by[i] = (int8_t)made_by;
                                stage[qi * KS + i] = next;
                                from = next, parent = i, made_by = entry;
                            }
                        }
                        n_open_sent += (uint64_t)q.n_open - 1, n_ahead_sent += (uint64_t)(q.nspec - q.n_open);
                    }
                    c_next += tick() - c2;
                }
                // The nodes of query k of the group's live list go into slots [k * r_nodes, (k + 1) * r_nodes) of the launch,
                // copied by the thread that has just written them; slots the query does not use, and all slots of a query that
                // has finished, carry STEP_DEAD.  (Fixed places: the workers share no counter for them.)
                const uint64_t c3 = tick();
                for (size_t k = k0; k < k1; ++k)
                {
                    const uint32_t qi = g.live[k];
                    const size_t at = k * r_nodes;
                    const size_t used = wants[qi] ? (size_t)Q[qi].nspec : 0;
                    slot0[qi] = (uint32_t)at;
                    StepIn *dst = r_in + (g.part_no & 1u) * nodes_max + g.base + at;
                    for (size_t i = 0; i < used; ++i)
                        dst[i] = stage[qi * KS + i];
                    for (size_t i = used; i < r_nodes; ++i)
                        dst[i] = StepIn{0.f, 0.f, 0.f, 0.f, 0, qi, STEP_DEAD, 0};
                }
                Tally &tl = tally[w];
                tl.hits += hits, tl.deep += deep, tl.ahead += ahead;
                if (trace_on)
                {
                    tl.ahead_sent += n_ahead_sent, tl.open_sent += n_open_sent;
                    tl.c_consume += c_consume, tl.c_advance += c_advance, tl.c_next += c_next, tl.c_copy += tick() - c3;
                }
            }
        };
        // what the worker that finishes a host part last does before it launches: queries that finished drop out
        auto end_host_part = [&](Group &g) -> bool {
            g.ns = g.live.size() * g.r_nodes; // every query that was alive at the release has its slots in the launch
            g.keep.clear();
            for (uint32_t qi : g.live)
                if (wants[qi])
                    g.keep.push_back(qi);
            g.live.swap(g.keep);
            if (g.live.empty())
                g.ns = 0;
            g.n_upd = std::min<size_t>(g.upd_claim.load(std::memory_order_relaxed), g.upd_cap);
            return g.ns != 0;
        };
        // One launch per round.  The kernel reads the popped nodes from, and writes its results to, the pinned
        // host block directly (the records are a few KB: no staging copies, no copy engines in the round trip).
        auto launch = [&](Group &g, int gi) -> int {
            hipStream_t st = c->round_stream[gi];
            hipEvent_t *ev = c->round_ev[gi];
            const size_t s0 = g.base, ns = g.ns;
            const double t0 = now();
            if (c->d_nodes)
                __builtin_ia32_sfence(); // the stores into device memory leave the write-combining buffers before the doorbell
            // Kernel time: the group's pair of events goes round every second launch at most.  The pair recorded two
            // launches ago is read first: this launch follows a host part that read answers of the launch before it,
            // so the one before that has finished.
            if (g.timed)
            {
                float ms = 0;
                if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess)
                    g.k_ms += ms, g.timed_launches += (int64_t)DEPTH;
                else
                    (void)hipGetLastError();
            }
            g.timed = ev_every > 0 && g.rounds % (2 * ev_every) == 0;
            g.rounds += 1;
            if (g.timed)
                (void)hipEventRecord(ev[0], st);
            const size_t par = g.part_no & 1u; // which set of node slots and cell updates this host part has filled
            for (size_t l = 0; l < DEPTH; ++l)
            {
                const StepIn *d_in = l == 0 ? k_in + par * nodes_max + s0 : (const StepIn *)(ds + o_child + (l - 1) * child_stride) + s0;
                StepIn *d_child = l + 1 < DEPTH ? (StepIn *)(ds + o_child + l * child_stride) + s0 : nullptr;
                // the first launch of a round brings the cell bounds up to date, every launch tests against them
                const CellFilter filter{filter_on ? c->d_bound : nullptr, (uint32_t)cells3d, k_upd + par * upd_max + g.upd_base, l == 0 ? (uint32_t)g.n_upd : 0u};
                const RoundOut out{(uint4 *)(hd + out_offset(g, l)), (uint32_t)out_units, g.part_no};
                launch_round(m, d_in, (uint32_t)ns, (const hyas_goal *)(ds + o_goal), d_twod_cur, c->d_seg, c->seg_cap, nullptr, nullptr, nullptr, nullptr,
                             c->stats_on ? c->d_stats : nullptr, c->stats_mode, st, d_child, P_analytic, filter, out);
            }
            if (g.timed)
                (void)hipEventRecord(ev[1], st);
            hipError_t e = hipGetLastError();
            g.launches += (int64_t)DEPTH;
            g.t_launch += now() - t0;
            return e == hipSuccess ? HYAS_OK : HYAS_ENODEV;
        };
        std::atomic<bool> release_busy{false};
        auto work = [&](int w) {
            if (w >= nw)
                return;
            if (hipSetDevice(c->device) != hipSuccess)
            {
                G[0].rc = HYAS_ENODEV;
                abort_all.store(true, std::memory_order_release);
                return;
            }
            cpu_set_t before;
            const bool pinned = w == 0 && c->pools->round_pinned && pthread_getaffinity_np(pthread_self(), sizeof before, &before) == 0;
            if (pinned)
                pin_self(0, nw);
            struct Unpin // the calling thread is worker 0: it gets its own mask back
            {
                bool on;
                cpu_set_t *set;
                ~Unpin()
                {
                    if (on)
                        (void)pthread_setaffinity_np(pthread_self(), sizeof *set, set);
                }
            } unpin{pinned, &before};
            uint64_t next = 0; // host parts this worker has done its share of
            unsigned idle = 0;
            t_probe = ConsumeProbe();
            struct KeepProbe // what consume() counted on this thread goes to the worker's tally
            {
                ConsumeProbe &to;
                ~KeepProbe() { to = t_probe; }
            } keep_probe{tally[w].probe};
            while (!abort_all.load(std::memory_order_acquire) && n_finished.load(std::memory_order_acquire) < ng)
            {
                if (released.load(std::memory_order_acquire) == next)
                {
                    if (++idle > 20000) // more threads than free CPUs: let the others run
                        std::this_thread::yield();
                    else
                        for (int i = 0; i < 4; ++i)
                            __builtin_ia32_pause();
                    continue;
                }
                idle = 0;
                const uint64_t entry = order[next % ORDER_CAP].load(std::memory_order_relaxed);
                const int gi = (int)(entry & 0xFFu);
                const uint32_t part = (uint32_t)(entry >> 8);
                ++next;
                Group &g = G[gi];
                // The pieces this worker gets hold of; what they change in the cell state goes into one list, handed in
                // when no piece is left to take, and only then are the pieces reported as done: the worker whose report
                // completes the host part finds everybody's nodes and cell changes in place.
                UpdSink sink{filter_on ? g.upd[w].get() : nullptr, 0, filter_on ? (uint32_t)g.upd_cap : 0u};
                uint32_t mine = 0;
                uint64_t t_in = 0;
                for (int k = 0; k < nw; ++k) // the home range first, then what is left of the others'
                {
                    Group::Cursor &cu = g.cur[(w + k) % nw];
                    for (;;)
                    {
                        uint64_t v = cu.v.load(std::memory_order_acquire);
                        const uint32_t nxt = (uint32_t)(v & 0xFFFFFu), end = (uint32_t)(v >> 20 & 0xFFFFFu);
                        if ((uint32_t)(v >> 40) != part || nxt >= end)
                            break;
                        if (!cu.v.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel))
                            continue;
                        if (trace_on && !mine)
                            t_in = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
                        host_piece(g, w, (size_t)nxt * PIECE, std::min(g.live.size(), ((size_t)nxt + 1) * PIECE), filter_on ? &sink : nullptr);
                        ++mine;
                    }
                }
                if (!mine)
                    continue;
                if (sink.n)
                {
                    const uint32_t u0 = g.upd_claim.fetch_add(sink.n, std::memory_order_relaxed);
                    const size_t room = u0 < g.upd_cap ? g.upd_cap - u0 : 0, take_n = std::min<size_t>(sink.n, room);
                    CellUpd *dst = r_upd + (g.part_no & 1u) * upd_max + g.upd_base + u0;
                    for (size_t i = 0; i < take_n; ++i)
                        dst[i] = sink.p[i];
                }
                if (c->d_nodes)
                    __builtin_ia32_sfence(); // stores into device memory leave this core's write-combining buffers
                if (trace_on)
                {
                    const uint64_t t_out = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
                    g.host_ns.fetch_add(t_out - t_in, std::memory_order_relaxed);
                    uint64_t v = g.first_ns.load();
                    while (t_in < v && !g.first_ns.compare_exchange_weak(v, t_in))
                        ;
                    v = g.last_ns.load();
                    while (t_out > v && !g.last_ns.compare_exchange_weak(v, t_out))
                        ;
                }
                if (g.pieces_left.fetch_sub(mine, std::memory_order_acq_rel) != mine)
                    continue;
                if (!end_host_part(g))
                {
                    g.state.store(G_FINISHED, std::memory_order_release);
                    n_finished.fetch_add(1, std::memory_order_acq_rel);
                }
                else if ((g.rc = launch(g, gi)) != HYAS_OK)
                    abort_all.store(true, std::memory_order_release);
                else
                {
                    // The answers are read as they arrive: the group's next host part is released right away, behind
                    // the host parts of the launches that left before this one.
                    while (release_busy.exchange(true, std::memory_order_acquire))
                        __builtin_ia32_pause();
                    begin_host_part(g, true);
                    release_busy.store(false, std::memory_order_release);
                }
            }
        };
        const double wall_setup = cur.t_all;
        Pool &round_pool = c->pools->round_pool(nw);
        if (filter_on)
        {
            // every cell of every query of the chunk unseen: +inf
            HYAS_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_bound, 0x7F800000, cn * cells3d, c->stream));
            HYAS_HIP(c, hipStreamSynchronize(c->stream));
        }
        std::memset(hs + o_succ, 0, (size_t)ng * DEPTH * res_stride(gslots)); // no unit carries a tag yet
        for (int gi = 0; gi < ng; ++gi)
            begin_host_part(G[gi], false);
        round_pool.run_parts(work);
        // the last launches may still be running (nodes whose answers the searches did not wait for)
        for (int gi = 0; gi < ng; ++gi)
            if (hipStreamSynchronize(c->round_stream[gi]) != hipSuccess && G[gi].rc == HYAS_OK)
                G[gi].rc = HYAS_ENODEV;
        for (int gi = 0; gi < ng; ++gi)
            if (G[gi].timed)
            {
                float ms = 0;
                if (hipEventElapsedTime(&ms, c->round_ev[gi][0], c->round_ev[gi][1]) == hipSuccess)
                    G[gi].k_ms += ms, G[gi].timed_launches += (int64_t)DEPTH;
                else
                    (void)hipGetLastError();
            }
        const double wall_rounds = since(wall0);
        Tally sum;
        for (const Tally &tl : tally)
        {
            sum.hits += tl.hits, sum.deep += tl.deep, sum.ahead += tl.ahead, sum.ahead_sent += tl.ahead_sent, sum.open_sent += tl.open_sent;
            sum.c_consume += tl.c_consume, sum.c_advance += tl.c_advance, sum.c_next += tl.c_next, sum.c_copy += tl.c_copy;
            sum.probe.dup += tl.probe.dup, sum.probe.cell += tl.probe.cell, sum.probe.push += tl.probe.push, sum.probe.chain += tl.probe.chain;
            sum.probe.heads += tl.probe.heads, sum.probe.succ += tl.probe.succ, sum.probe.pushed += tl.probe.pushed;
            sum.probe.drop_closed += tl.probe.drop_closed, sum.probe.drop_worse += tl.probe.drop_worse, sum.probe.drop_tie += tl.probe.drop_tie;
            sum.probe.drop_dup += tl.probe.drop_dup, sum.probe.waited += tl.probe.waited, sum.probe.waits += tl.probe.waits;
            sum.probe.waited_shot += tl.probe.waited_shot, sum.probe.waits_shot += tl.probe.waits_shot;
        }
        double t_host = 0, t_launch = 0;
        for (int gi = 0; gi < ng; ++gi)
        {
            if (G[gi].rc != HYAS_OK)
                return drain(c->fail(G[gi].rc, "a HIP call failed in a search round"));
            // kernel time of all launches, from the ones that carried timing events
            if (G[gi].timed_launches > 0)
                c->k_ms[0] += G[gi].k_ms * (double)G[gi].launches / (double)G[gi].timed_launches;
            c->k_launches[0] += G[gi].launches;
            t_host += (double)G[gi].host_ns.load() * 1e-9, t_launch += G[gi].t_launch;
        }

        if (trace_on)
            fprintf(stderr, "hyas rounds ended by: a successor of a node of the same round %llu, an older open-list entry %llu; answered ahead: "
                            "open-list entries %llu of %llu sent, nodes sent ahead of their existence %llu of %llu (at most %d per query), successors expanded "
                            "by the deeper launches %llu (%d launches per round)\n",
                    (unsigned long long)miss_child.load(), (unsigned long long)miss_other.load(), (unsigned long long)sum.hits,
                    (unsigned long long)sum.open_sent, (unsigned long long)sum.ahead, (unsigned long long)sum.ahead_sent, ahead_cfg,
                    (unsigned long long)sum.deep, depth_cfg);
        if (trace_on)
            fprintf(stderr, "hyas host threads, timestamp-counter cycles in all: taking successors %.3e, popping %.3e, choosing and staging the next nodes %.3e, "
                            "copying them into the launch %.3e\n",
                    (double)sum.c_consume, (double)sum.c_advance, (double)sum.c_next, (double)sum.c_copy);
        if (trace_on)
            for (int a = 0; a < 2; ++a)
                for (int pp = 0; pp < 8; ++pp)
                {
                    uint64_t row = 0;
                    for (int cp = 0; cp < 8; ++cp)
                        row += probe_hist[a * 64 + pp * 8 + cp].load();
                    if (!row)
                        continue;
                    fprintf(stderr, "hyas round-ending successor of %s, parent's motion-set entry %d: by own motion-set entry", a ? "another node of the round" : "the round's first node", pp);
                    for (int cp = 0; cp < 8; ++cp)
                        fprintf(stderr, " %llu", (unsigned long long)probe_hist[a * 64 + pp * 8 + cp].load());
                    fprintf(stderr, "\n");
                }
        if (trace_on)
            fprintf(stderr, "hyas taking successors: %.3e successors, duplicate check %.3e cycles (%.3e occupied buckets, %.3e open-list entries compared), "
                            "cell lookup %.3e cycles, %.3e pushes %.3e cycles\n",
                    (double)sum.probe.succ, (double)sum.probe.dup, (double)sum.probe.heads, (double)sum.probe.chain, (double)sum.probe.cell,
                    (double)sum.probe.pushed, (double)sum.probe.push);
        if (trace_on)
            fprintf(stderr, "hyas answers waited for: %.3e times, %.3e cycles in all; of which the shot's answer %.3e times, %.3e cycles\n", (double)sum.probe.waits,
                    (double)sum.probe.waited, (double)sum.probe.waits_shot, (double)sum.probe.waited_shot);
        if (trace_on)
            fprintf(stderr, "hyas successors dropped: cell closed %.3e, cell open with a cost not worse %.3e, same cell beyond the tie breaker %.3e, "
                            "duplicate of an open-list entry %.3e\n",
                    (double)sum.probe.drop_closed, (double)sum.probe.drop_worse, (double)sum.probe.drop_tie, (double)sum.probe.drop_dup);
        if (trace_on)
            for (int gi = 0; gi < ng; ++gi)
                fprintf(stderr, "hyas group %d host parts, from release: first worker begins after %.1f us, last worker ends after %.1f us\n", gi,
                        G[gi].sum_first * 1e-3 / std::max<uint64_t>(G[gi].parts, 1), G[gi].sum_last * 1e-3 / std::max<uint64_t>(G[gi].parts, 1));
        if (trace_on)
            for (int gi = 0; gi < ng; ++gi)
                fprintf(stderr, "hyas group %d: %lld launches, kernel %.1f us, submit-to-seen minus kernel %.1f us, seen-to-next-submit %.1f us, host parts %.3f thread-s, launch %.3f s\n",
                        gi, (long long)G[gi].launches, G[gi].k_ms * 1e3 / std::max<int64_t>(G[gi].timed_launches, 1), G[gi].t_turn * 1e6 / std::max<int64_t>(G[gi].timed_launches / (int64_t)DEPTH, 1),
                        G[gi].t_idle * 1e6 / std::max<int64_t>(G[gi].launches, 1), (double)G[gi].host_ns.load() * 1e-9, G[gi].t_launch);
        if (trace_on)
            fprintf(stderr,
                    "hyas rounds (%d groups, %d worker threads, %d nodes per query): host parts %.3f thread-s of %.3f, launches %.3f thread-s; "
                    "%llu expansions answered ahead of time; wall: set-up %.3f s (pose checks + 2D floods %.3f s), rounds %.3f s\n",
                    ng, nw, spec_k, t_host, wall_rounds * nw, t_launch, (unsigned long long)sum.hits, wall_setup, wall_flood, wall_rounds);
        // paths back to the start (TracePath, :813-828), GetPath's RemoveDuplicates for the endings that have it in front
        // of ShortCut (:121, :138), then ShortCut itself (:122-125, :139-142, :165-168) for all of them in one launch
        {
            std::vector<uint32_t> fin;
            for (size_t i = 0; i < cn; ++i)
                if (Q[i].end_node >= 0 || Q[i].shot_from >= 0)
                    fin.push_back((uint32_t)i);
            pool.run(fin.size(), [&](size_t a, size_t b) {
                for (size_t k = a; k < b; ++k)
                {
                    Query &q = Q[fin[k]];
                    if (q.end_node >= 0)
                    {
                        D.trace(q, q.end_node, q.path);
                        Driver::remove_duplicates(q.path);
                    }
                    else
                        D.trace(q, q.shot_from, q.path);
                }
            });
            if (c->P.short_cut && !fin.empty())
            {
                std::vector<uint32_t> offs(fin.size() + 1, 0);
                for (size_t k = 0; k < fin.size(); ++k)
                    offs[k + 1] = offs[k] + (uint32_t)(Q[fin[k]].path.size() / 3);
                std::vector<float> all((size_t)offs.back() * 3);
                for (size_t k = 0; k < fin.size(); ++k)
                    std::copy(Q[fin[k]].path.begin(), Q[fin[k]].path.end(), all.begin() + 3 * (size_t)offs[k]);
                std::vector<uint32_t> idx((size_t)offs.back() + fin.size()), cnt(fin.size());
                if ((rc = shortcut_paths(c, all.data(), offs.data(), fin.size(), idx.data(), cnt.data())) != HYAS_OK)
                    return rc;
                pool.run(fin.size(), [&](size_t a, size_t b) {
                    std::vector<float> np;
                    for (size_t k = a; k < b; ++k)
                    {
                        Query &q = Q[fin[k]];
                        const uint32_t *ix = &idx[(size_t)offs[k] + k];
                        np.clear();
                        for (uint32_t j = cnt[k]; j-- > 0;) // pushed from the last node backwards; std::reverse at :1177
                            np.insert(np.end(), q.path.begin() + 3 * (size_t)ix[j], q.path.begin() + 3 * (size_t)ix[j] + 3);
                        q.path = np;
                    }
                });
            }
        }
        // queries closed by an analytic shot: fetch the curve samples of all of them in two launches
        // (sample counts, then the samples) (:160-172: TracePath + the sampled curve + the goal)
        {
            std::vector<uint32_t> shot_q;
            for (size_t i = 0; i < cn; ++i)
                if (Q[i].shot_from >= 0)
                    shot_q.push_back((uint32_t)i);
            const size_t nsq = shot_q.size();
            if (nsq)
            {
                // the round staging is idle now: nodes reuse the StepIn slots, counts and offsets the result slices
                uint32_t *h_cnt = (uint32_t *)(hs + o_succ), *h_off = h_cnt + nsq;
                uint32_t *d_cnt = (uint32_t *)(ds + o_succ), *d_off = d_cnt + nsq;
                for (size_t k = 0; k < nsq; ++k)
                {
                    const Query &q = Q[shot_q[k]];
                    const SNode &nd = q.arena[q.shot_from];
                    h_in[k] = StepIn{nd.x, nd.y, nd.t, nd.g, nd.prim, shot_q[k], 1, 0};
                }
                HYAS_HIP(c, hipMemcpyAsync(ds + o_in, h_in, nsq * sizeof(StepIn), hipMemcpyHostToDevice, c->stream));
                launch_shot_path(m, (const StepIn *)(ds + o_in), (uint32_t)nsq, (const hyas_goal *)(ds + o_goal), c->d_seg, c->seg_cap, nullptr,
                                 nullptr, d_cnt, c->stream);
                HYAS_HIP(c, hipMemcpyAsync(h_cnt, d_cnt, nsq * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                HYAS_HIP(c, hipStreamSynchronize(c->stream));
                size_t total = 0;
                for (size_t k = 0; k < nsq; ++k)
                    h_off[k] = (uint32_t)total, total += h_cnt[k];
                if ((rc = ensure_buffer(c, (void **)&c->d_pts, &c->d_pts_bytes, std::max<size_t>(total, 1) * 3 * sizeof(float))) != HYAS_OK)
                    return rc;
                float *d_pts = c->d_pts;
                shot_buf.resize(total * 3);
                hipError_t e = hipMemcpyAsync(d_off, h_off, nsq * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess)
                {
                    launch_shot_path(m, (const StepIn *)(ds + o_in), (uint32_t)nsq, (const hyas_goal *)(ds + o_goal), c->d_seg, c->seg_cap,
                                     d_off, d_pts, nullptr, c->stream);
                    e = hipGetLastError();
                }
                if (e == hipSuccess && total)
                    e = hipMemcpyAsync(shot_buf.data(), d_pts, total * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess)
                    return hip_fail(c, e, "shot path download");
                const bool piecewise = c->P.piecewise_cubic_bezier_interpolation != 0;
                // piecewise interpolation (:178-189, :218-229): per query the anchor poses and the sample count
                std::vector<std::vector<float>> anchors(piecewise ? nsq : 0);
                std::vector<uint32_t> pw_n(piecewise ? nsq : 0);
                pool.run(nsq, [&](size_t a, size_t b) {
                    for (size_t k = a; k < b; ++k)
                    {
                        Query &q = Q[shot_q[k]];
                        const size_t analytical_expansion_index = q.path.size() / 3; // :169, :212
                        q.path.insert(q.path.end(), shot_buf.begin() + 3 * (size_t)h_off[k], shot_buf.begin() + 3 * ((size_t)h_off[k] + h_cnt[k]));
                        q.path.push_back(q.goal.x), q.path.push_back(q.goal.y), q.path.push_back(q.goal.t);
                        if (!piecewise)
                        {
                            D.finish(q, HYAS_STATUS_SHOT, q.shot_from, 2);
                            continue;
                        }
                        if (c->P.analytical_expansion_every_point)
                            Driver::remove_duplicates(q.path); // :172 runs before the conversion, :230 after the return
                        pw_n[k] = D.piecewise_anchors(q, analytical_expansion_index, anchors[k]);
                    }
                });
                if (piecewise)
                {
                    // every curve of every query in one launch: curve j of query k runs between its anchors j and j + 1
                    std::vector<float> anc;
                    std::vector<uint32_t> first, off, cnt, q_off(nsq + 1, 0);
                    size_t total_pts = 0;
                    for (size_t k = 0; k < nsq; ++k)
                    {
                        const size_t na = anchors[k].size() / 3, base_anchor = anc.size() / 3;
                        anc.insert(anc.end(), anchors[k].begin(), anchors[k].end());
                        for (size_t j = 0; j + 1 < na; ++j)
                            first.push_back((uint32_t)(base_anchor + j)), off.push_back((uint32_t)total_pts), cnt.push_back(pw_n[k]),
                                total_pts += (size_t)pw_n[k] + 1;
                        q_off[k + 1] = (uint32_t)total_pts;
                    }
                    std::vector<float> pts(total_pts * 3);
                    if ((rc = bezier_points(c, anc.data(), anc.size() / 3, first.data(), off.data(), cnt.data(), first.size(), total_pts, pts.data())) != HYAS_OK)
                        return rc;
                    pool.run(nsq, [&](size_t a, size_t b) {
                        for (size_t k = a; k < b; ++k)
                        {
                            Query &q = Q[shot_q[k]];
                            // piecewise path, then the analytic path once more (:181, :221)
                            std::vector<float> pw(pts.begin() + 3 * (size_t)q_off[k], pts.begin() + 3 * (size_t)q_off[k + 1]);
                            pw.insert(pw.end(), shot_buf.begin() + 3 * (size_t)h_off[k], shot_buf.begin() + 3 * ((size_t)h_off[k] + h_cnt[k]));
                            pw.push_back(q.goal.x), pw.push_back(q.goal.y), pw.push_back(q.goal.t);
                            q.path.swap(pw);
                            D.finish(q, HYAS_STATUS_SHOT, q.shot_from, 2); // RemoveDuplicates of :184 / :224 and of :994
                        }
                    });
                }
            }
        }

        for (size_t i = 0; i < cn; ++i)
            if (Q[i].end_node >= 0)
                D.finish(Q[i], Q[i].end_status, Q[i].end_node, 1);
        for (size_t i = 0; i < cn; ++i)
        {
            out[base + i] = Q[i].res;
            c->paths[base + i].swap(Q[i].path);
            c->expanded[base + i].swap(Q[i].expanded);
        }
        if (trace_on)
            fprintf(stderr, "hyas chunk of %zu queries: set-up %.3f s%s, rounds and finish %.3f s, %.3f s after the rounds\n", cn, wall_setup,
                    base ? " (beside the previous chunk's rounds)" : "", since(wall0), since(wall0) - wall_rounds);
        if (!has_next)
            break;
        if (pipelined)
            prep_thread.join();
        else
        {
            Q.clear();
            nxt.rc = prepare(nxt, false);
        }
        if (nxt.rc != HYAS_OK)
            return nxt.rc;
        cur = std::move(nxt);
    }
    return HYAS_OK;
}

} // namespace hyas
