// Test scaffolding, never shipped: the product's device geometry (hyas_dev.h)
// and list pruning (hyas_prune.h) compiled for the host, so the CPU test suite
// can compare the exact code the kernels run against the oracle on adversarial
// poses without a GPU.  The product library has no such path.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../path_planner_amd/csrc/hyas_dev.h"
#include "../../path_planner_amd/csrc/hyas_prune.h"
#include "../../path_planner_amd/csrc/hyas_rs.h"
#include "../../path_planner_amd/csrc/hyas_flood.h"
#include "../../path_planner_amd/csrc/hyas_bezier.h"
#include "../../path_planner_amd/csrc/hyas_adaptive.h"
#include "../../path_planner_amd/csrc/hyas_open_list.h"
#include <queue>
#include <set>

using namespace hyas;

// quads8: nq x 4 vertices (x, y); lists are rebuilt here from the quads with the
// reference's range rule (src/collisiondetection.cpp:319-354); verdict = on-grid
// && pose test && (pred ? swept-segment test : true).  prune = 0 walks the full
// per-cell lists with culling disabled.
extern "C" int hg_collision(const float *quads8, int nq, const uint8_t *charmap, int w, int h, float veh_len, float veh_wid, float range,
                            const float *xyt, const float *pred, int n, uint8_t *out, int prune, uint32_t *entries /* [3] */)
{
    std::vector<Quad> Q(nq);
    for (int i = 0; i < nq; i++)
    {
        Quad &q = Q[i];
        for (int k = 0; k < 4; k++)
            q.vx[k] = quads8[i * 8 + 2 * k], q.vy[k] = quads8[i * 8 + 2 * k + 1];
        q.xmin = q.xmax = q.vx[0], q.ymin = q.ymax = q.vy[0];
        for (int k = 1; k < 4; k++)
        {
            q.xmin = std::fmin(q.xmin, q.vx[k]), q.xmax = std::fmax(q.xmax, q.vx[k]);
            q.ymin = std::fmin(q.ymin, q.vy[k]), q.ymax = std::fmax(q.ymax, q.vy[k]);
        }
        q.axis_rect = (q.vx[0] == q.xmin && q.vy[0] == q.ymin && q.vx[1] == q.xmax && q.vy[1] == q.ymin && q.vx[2] == q.xmax &&
                       q.vy[2] == q.ymax && q.vx[3] == q.xmin && q.vy[3] == q.ymax && q.xmin < q.xmax && q.ymin < q.ymax);
    }
    std::vector<uint32_t> off(w * h + 1);
    std::vector<uint16_t> idx;
    for (int key = 0; key < w * h; key++)
    {
        off[key] = (uint32_t)idx.size();
        const float x = (float)(key % w), y = (float)(key / w);
        for (int q = 0; q < nq; q++)
            if (dist_quad_point(Q[q], f2(x, y)) < range)
                idx.push_back((uint16_t)q);
    }
    off[w * h] = (uint32_t)idx.size();
    DevMap m{};
    m.charmap = charmap, m.w = w, m.h = h, m.res = 1, m.ox = 0, m.oy = 0, m.res_f = 1, m.ox_f = 0, m.oy_f = 0;
    m.res_pow2 = 1, m.inv_res = 1, m.inv_res_f = 1;
    m.quads = Q.data(), m.n_quads = nq, m.cell_off = off.data(), m.cell_idx = idx.data();
    m.veh_len = veh_len, m.veh_wid = veh_wid, m.half_wid = 0.5 * (double)veh_wid;
    m.cull_margin = prune ? (float)(0.01 + 2e-3 * std::sqrt((double)w * w + (double)h * h)) : 1e9f;
    std::vector<uint32_t> po, so;
    std::vector<uint16_t> pi, si;
    const float lmax = 0.3f;
    prune_lists(m, Q.data(), off.data(), idx.data(), lmax, po, pi, so, si);
    m.pose_off = po.data(), m.pose_idx = pi.data(), m.seg_off = so.data(), m.seg_idx = si.data();
    m.seg_lmax = prune ? lmax : 0.f;
    if (!prune)
        m.pose_off = off.data(), m.pose_idx = idx.data();
    if (entries)
        entries[0] = (uint32_t)idx.size(), entries[1] = (uint32_t)pi.size(), entries[2] = (uint32_t)si.size();
    const GlobalQuads gq{Q.data()};
    for (int i = 0; i < n; i++)
    {
        const float x = xyt[3 * i], y = xyt[3 * i + 1], t = xyt[3 * i + 2];
        bool ok = on_grid(m, x, y) && pose_free(m, gq, x, y, t);
        if (ok && pred)
            ok = segment_free(m, gq, x, y, pred[2 * i], pred[2 * i + 1]);
        out[i] = ok ? 1 : 0;
    }
    return 0;
}

// The side-by-side form of the Reeds-Shepp candidates (rs::candidate_staged, what the shot kernel runs) against the
// plain form (rs::candidate, what the table kernel and the oracle's restatement follow): number of (goal, candidate)
// pairs whose validity, compared length or path record differ in any bit.  xyphi: n normalised goals (x, y, phi).
extern "C" long hg_rs_staged_mismatches(const double *xyphi, int n, long *valid_out)
{
    long bad = 0, valid = 0;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < RS_NUM_CANDIDATES; c++)
        {
            rs::Path a = rs::empty_path(), b = rs::empty_path();
            double la = 0, lb = 0;
            const bool oa = rs::candidate(c, xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2], a, la);
            const bool ob = rs::candidate_staged(c, xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2], b, lb);
            valid += oa;
            if (oa != ob)
            {
                ++bad;
                continue;
            }
            if (!oa)
                continue;
            bool same = a.type == b.type && std::memcmp(&la, &lb, 8) == 0 && std::memcmp(&a.total, &b.total, 8) == 0;
            for (int k = 0; k < 5; k++)
                same = same && std::memcmp(&a.len[k], &b.len[k], 8) == 0;
            bad += !same;
        }
    if (valid_out)
        *valid_out = valid;
    return bad;
}

// rs::solve_hoisted (what the table kernel runs: sine and cosine of +-phi taken once for all candidates) against
// rs::solve_serial (the reference's walk, src/ReedsShepp.cpp:494-503): goals whose winning path differs in any bit.
extern "C" long hg_rs_solve_uniform_mismatches(const double *xyphi, int n)
{
    long bad = 0;
    for (int i = 0; i < n; i++)
    {
        const rs::Path a = rs::solve_serial(xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2]);
        const rs::Path b = rs::solve_hoisted(xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2]);
        bool same = a.type == b.type && std::memcmp(&a.total, &b.total, 8) == 0;
        for (int k = 0; k < 5; k++)
            same = same && std::memcmp(&a.len[k], &b.len[k], 8) == 0;
        bad += !same;
    }
    return bad;
}

// rs::shortest_length (what the heuristic-table kernel runs: the candidates regrouped by symmetry, shared polar forms,
// the unrolled remainder) against the total of rs::solve_serial: goals where a bit differs.  Also counts the goals
// whose mod2pi_few_turns differs from mod2pi over a sweep of arguments derived from them.
extern "C" long hg_rs_shortest_length_mismatches(const double *xyphi, int n)
{
    long bad = 0;
    for (int i = 0; i < n; i++)
    {
        const double a = rs::solve_serial(xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2]).total;
        static const double quarter_atans[8] = HY_QUARTER_ATANS;
        const double b = rs::shortest_length(xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2], quarter_atans);
        bad += std::memcmp(&a, &b, 8) != 0;
    }
    return bad;
}
extern "C" long hg_mod2pi_mismatches(const double *x, int n)
{
    long bad = 0;
    for (int i = 0; i < n; i++)
    {
        bool beyond = false;
        const double a = rs::mod2pi(x[i]), b = rs::mod2pi_few_turns(x[i], beyond);
        const bool outside = !(std::fabs(x[i]) < 8 * HY_PI); // then the caller takes the general form
        bad += beyond != outside || (!beyond && std::memcmp(&a, &b, 8) != 0);
    }
    return bad;
}

// hy_sincos against hy_sin and hy_cos: number of arguments where a bit differs.
extern "C" long hg_sincos_mismatches(const double *x, int n)
{
    long bad = 0;
    for (int i = 0; i < n; i++)
    {
        double s, c;
        hy_sincos(x[i], &s, &c);
        const double s1 = hy_sin(x[i]), c1 = hy_cos(x[i]);
        bad += std::memcmp(&s, &s1, 8) != 0 || std::memcmp(&c, &c1, 8) != 0;
    }
    return bad;
}

// rs::interpolate_prefix (per-segment prefix states + one sine/cosine pair per point, what the shot kernel runs)
// against rs::interpolate (the reference's walk, src/ReedsShepp.cpp:576-625): points whose state differs in any bit.
// For each of the n goals the shortest curve from the origin pose (0, 0, phi0) is sampled at `per` arc lengths.
extern "C" long hg_rs_interp_mismatches(const double *xyphi, int n, int per, double rho)
{
    long bad = 0;
    for (int i = 0; i < n; i++)
    {
        const double q0[3] = {0.25 * i, -0.5 * i, xyphi[3 * i + 2] * 0.7};
        const rs::Path path = rs::solve_serial(xyphi[3 * i], xyphi[3 * i + 1], xyphi[3 * i + 2]);
        if (!(path.total < 1e300))
            continue;
        rs::Prefix P;
        rs::build_prefix(path, q0[2], P);
        for (int k = 0; k <= per; k++)
        {
            // arc lengths across the whole curve, the exact segment ends among them
            double seg = path.total * k / per;
            if (k % 7 == 3)
            {
                seg = 0;
                for (int j = 0; j <= (k / 7) % 5; j++)
                    seg += std::fabs(path.len[j]);
            }
            double a[3], b[3];
            rs::interpolate(q0, path, seg, rho, a);
            rs::interpolate_prefix(q0, path, P, seg, rho, b);
            bad += std::memcmp(a, b, 24) != 0;
        }
    }
    return bad;
}

// The flood kernel's open queue (fl::push / fl::pop over flat storage) against std::priority_queue with the
// reference's `>=` comparator (include/hybrid_a_star.h:45-53): a random mix of pushes and pops with many equal
// priorities; returns the number of pops whose cell differs.
struct VecHeap
{
    std::vector<fl::Ent> v;
    fl::Ent get(int i) const { return v[(size_t)i]; }
    void set(int i, const fl::Ent &e)
    {
        if ((size_t)i >= v.size())
            v.resize((size_t)i + 1);
        v[(size_t)i] = e;
    }
};
struct RefCmp
{
    bool operator()(const std::pair<int, double> &l, const std::pair<int, double> &r) const { return l.second >= r.second; }
};
// The search's open list (sorted front array + heap) against an ordered set under the same (f, seq) order: a mix of
// pushes, pops and look-aheads (which move entries from the heap into the array); returns the number of pops whose node
// differs plus the number of steps after which size or emptiness differ.
extern "C" long hg_open_list_mismatches(const float *f, const uint8_t *op, int n_ops)
{
    OpenList o;
    struct Less
    {
        bool operator()(const HeapItem &a, const HeapItem &b) const { return a.f < b.f || (a.f == b.f && a.seq < b.seq); }
    };
    std::set<HeapItem, Less> ref;
    long bad = 0;
    uint32_t seq = 0;
    for (int i = 0; i < n_ops; i++)
    {
        if (op[i] == 1) // pop
        {
            if (ref.empty())
            {
                bad += o.empty() ? 0 : 1;
                continue;
            }
            if (o.empty() || o.top().node != ref.begin()->node)
                ++bad;
            if (!o.empty())
                o.pop();
            ref.erase(ref.begin());
        }
        else if (op[i] == 2) // look ahead: the array must then hold the list's best entries, best last
        {
            o.refill(11);
            auto it = ref.begin();
            for (int k = o.nh - 1; k >= 0; --k, ++it)
                if (it == ref.end() || it->node != o.hot[k].node)
                {
                    ++bad;
                    break;
                }
        }
        else
        {
            const HeapItem e{f[i], seq, (int32_t)seq, i};
            ++seq;
            o.push(e);
            ref.insert(e);
        }
        bad += (o.size() != ref.size()) + (o.empty() != ref.empty());
    }
    return bad;
}

extern "C" long hg_heap_mismatches(const float *prio, const uint8_t *is_pop, int n_ops)
{
    VecHeap h;
    int size = 0;
    std::priority_queue<std::pair<int, double>, std::vector<std::pair<int, double>>, RefCmp> pq;
    long bad = 0;
    for (int i = 0; i < n_ops; i++)
    {
        if (is_pop[i])
        {
            if (pq.empty())
                continue;
            const int want = pq.top().first;
            pq.pop();
            const fl::Ent got = fl::pop(h, size);
            --size;
            bad += (int)got.id != want;
        }
        else
        {
            fl::Ent e;
            e.prio = prio[i], e.id = (uint32_t)i, e.x = 0, e.y = 0;
            fl::push(h, size, e);
            ++size;
            pq.emplace(i, (double)prio[i]);
        }
    }
    while (!pq.empty())
    {
        const int want = pq.top().first;
        pq.pop();
        const fl::Ent got = fl::pop(h, size);
        --size;
        bad += (int)got.id != want;
    }
    return bad;
}

// The flood kernel's algorithm run serially on the host: per cell {cost, parent, state}, coordinates kept apart,
// the edge from the parent added at pop time from the stored coordinates of both cells.
// out: w*h + w + 1 floats, -1 where the flood never closed the cell.
extern "C" void hg_flood(int w, int h, double res, double ox, double oy, float gx, float gy, float *out)
{
    DevMap m{};
    m.w = w, m.h = h, m.res = res, m.ox = ox, m.oy = oy, m.res_f = (float)res, m.ox_f = (float)ox, m.oy_f = (float)oy;
    int e = 0;
    const bool p2 = std::frexp(res, &e) == 0.5 && (double)m.res_f == res;
    m.res_pow2 = p2, m.inv_res = p2 ? 1.0 / res : 0, m.inv_res_f = p2 ? (float)(1.0 / res) : 0;
    const uint32_t N = (uint32_t)w * h + w + 1;
    m.twod_size = N;
    std::vector<float> g(N, 0.f), X(N, 0.f), Y(N, 0.f);
    std::vector<uint32_t> ps(N, 0);
    VecHeap heap;
    int size = 0;
    const int gid = idx_2d(m, gx, gy);
    if (gid >= 0 && (uint32_t)gid < N)
    {
        ps[gid] = 1, g[gid] = 0, X[gid] = gx, Y[gid] = gy;
        fl::Ent en;
        en.prio = 0, en.id = (uint32_t)gid, en.x = gx, en.y = gy;
        heap.set(0, en);
        size = 1;
    }
    const float step = m.res_f;
    while (size > 0)
    {
        const fl::Ent top = fl::pop(heap, size);
        --size;
        const uint32_t id = top.id;
        const float cx = top.x, cy = top.y;
        float cg = g[id];
        if ((ps[id] & 3) == 2)
        {
            const uint32_t par = ps[id] >> 2;
            cg = cg + normf(sub(f2(cx, cy), f2(X[par], Y[par])));
        }
        g[id] = cg, ps[id] |= 3;
        for (int o = 0; o < 9; o++)
        {
            if (o == 4)
                continue;
            const int i = o / 3, j = o % 3;
            const float xs = cx + (i == 0 ? -step : i == 1 ? 0.f : step), ys = cy + (j == 0 ? -step : j == 1 ? 0.f : step);
            if (!on_grid(m, xs, ys))
                continue;
            const int n_ = idx_2d(m, xs, ys);
            if (n_ < 0 || (uint32_t)n_ >= N)
                continue;
            const int st = ps[n_] & 3;
            if (st == 3)
                continue;
            if (st != 0)
            {
                if (g[n_] > cg)
                    g[n_] = cg, ps[n_] = (id << 2) | 2u;
                continue;
            }
            X[n_] = xs, Y[n_] = ys, g[n_] = cg, ps[n_] = (id << 2) | 2u;
            fl::Ent en;
            en.prio = cg, en.id = (uint32_t)n_, en.x = xs, en.y = ys;
            fl::push(heap, size, en);
            ++size;
        }
    }
    for (uint32_t c = 0; c < N; c++)
        out[c] = (ps[c] & 3) == 3 ? g[c] : -1.f;
}

// The cubic Bezier header run on the host, in the order the kernels use it.
// Table entry (x, y, theta): weights of the 101 polyline vertices, then the running length.
extern "C" float hg_bezier_table_cost(float x, float y, float theta)
{
    float px[4], py[4], w[4];
    bz::lib_control(x, y, theta, px, py);
    bz::lib_weights(bz::lib_param(0), w);
    float ax = bz::lib_axis(w, px), ay = bz::lib_axis(w, py), length = 0.0f;
    for (unsigned v = 1; v <= HYAS_BZ_INTERVALS; ++v)
    {
        bz::lib_weights(bz::lib_param(v), w);
        const float bx = bz::lib_axis(w, px), by = bz::lib_axis(w, py), dx = bx - ax, dy = by - ay;
        length += hy_sqrtf(dx * dx + dy * dy);
        ax = bx, ay = by;
    }
    return length;
}
// Shot curve start -> goal: returns its length; out4 = {x, y, heading, curvature} at parameter t.
extern "C" float hg_bezier_eval(const float *start, const float *goal, float t, float *out4)
{
    const bz::Curve c = bz::make(start, goal);
    float vx[HYAS_BZ_INTERVALS + 1], vy[HYAS_BZ_INTERVALS + 1], len = 0;
    for (unsigned i = 0; i <= HYAS_BZ_INTERVALS; ++i)
        bz::value(c, bz::powers(bz::length_param(i)), vx[i], vy[i]);
    for (unsigned i = 0; i < HYAS_BZ_INTERVALS; ++i)
    {
        const float dx = vx[i + 1] - vx[i], dy = vy[i + 1] - vy[i];
        len += hy_sqrtf(dx * dx + dy * dy);
    }
    const bz::Powers pw = bz::powers(t);
    bz::value(c, pw, out4[0], out4[1]);
    bz::angle_curvature(c, pw, out4[2], out4[3]);
    return len;
}

// ---- adaptive step size / steering, the shared header run on the host --------------------------------------------
extern "C" float hg_ray_distance(const float *quad8, float px, float py, float angle)
{
    Quad q{};
    for (int i = 0; i < 4; ++i)
        q.vx[i] = quad8[2 * i], q.vy[i] = quad8[2 * i + 1];
    return ad::ray_distance(q, f2(px, py), angle);
}
extern "C" int hg_sweep_angles(float *out, int cap)
{
    const std::vector<float> a = ad::sweep_angles();
    for (int i = 0; i < (int)a.size() && i < cap; ++i)
        out[i] = a[i];
    return (int)a.size();
}
// run-length pass + cut to the circle for one cell; out3: {label, start, width} triples
extern "C" int hg_ranges_of_sweep(const float *angle, const float *dist, int n, float min_distance, float detection_range, int cut, float *out3, int cap)
{
    std::vector<ad::Range> v;
    ad::ranges_of_sweep(angle, dist, (size_t)n, min_distance, detection_range, v);
    if (cut)
        ad::cut_to_circle(v);
    for (int i = 0; i < (int)v.size() && i < cap; ++i)
        out3[3 * i] = v[i].dist, out3[3 * i + 1] = v[i].start, out3[3 * i + 2] = v[i].range;
    return (int)v.size();
}
// The pair list of n nodes over a map given by its per-key ranges (CSR) and densities.
// flags: bit 0 add_one_more, 1 only in free ranges, 2 towards free range, 3 fixed number; steer/n_steer: the forward
// steering set.  pairs2: n x 13 x {step, steer}; counts[i] = pairs of node i, -1 on overflow.
extern "C" void hg_step_steer_pairs(int w, int h, double res, double ox, double oy, const uint32_t *ad_off, const float *ad_rng, const float *ad_density,
                                    float detection_range, float constant_density, int mode, int flags, const float *steer, int n_steer,
                                    const float *nodes_xyt, const float *goals_xyt, const float *dist_start_goal, int n, float *pairs2, int *counts,
                                    int *shared_out_checked)
{
    *shared_out_checked = 0;
    DevMap m{};
    m.w = w, m.h = h, m.res = res, m.ox = ox, m.oy = oy, m.res_f = (float)res, m.ox_f = (float)ox, m.oy_f = (float)oy;
    int e = 0;
    const bool p2 = std::frexp(res, &e) == 0.5 && (double)m.res_f == res;
    m.res_pow2 = p2, m.inv_res = p2 ? 1.0 / res : 0, m.inv_res_f = p2 ? (float)(1.0 / res) : 0;
    m.ad_off = ad_off, m.ad_rng = ad_rng, m.ad_density = ad_density, m.ad_range = detection_range, m.ad_constant_density = constant_density;
    m.ad_mode = mode, m.ad_one_more = flags & 1, m.ad_one_more_free_only = (flags >> 1) & 1, m.ad_towards_free = (flags >> 2) & 1,
    m.ad_fixed_number = (flags >> 3) & 1;
    m.n_succ = n_steer;
    for (int k = 0; k < n_steer; ++k)
        m.succ_steer[k] = steer[k], m.succ_reverse[k] = 0;
    for (int i = 0; i < n; ++i)
    {
        float st[HYAS_AD_MAX_PAIRS], sa[HYAS_AD_MAX_PAIRS];
        ad::Pairs p;
        p.step = st, p.steer = sa;
        ad::step_steer_pairs(m, nodes_xyt[3 * i], nodes_xyt[3 * i + 1], nodes_xyt[3 * i + 2], goals_xyt[3 * i], goals_xyt[3 * i + 1], goals_xyt[3 * i + 2],
                             dist_start_goal[i], p);
        counts[i] = p.overflow ? -1 : p.n;
        for (int k = 0; k < p.n; ++k)
            pairs2[(i * HYAS_AD_MAX_PAIRS + k) * 2] = st[k], pairs2[(i * HYAS_AD_MAX_PAIRS + k) * 2 + 1] = sa[k];
        // the form the kernel's lanes share out (range by range, then the duplicate rule) must give the same list
        const float x = nodes_xyt[3 * i], y = nodes_xyt[3 * i + 1], t = nodes_xyt[3 * i + 2];
        const ad::Cell c = ad::cell_of(m, x, y);
        if (mode == 1 && ad::staged_selection_applies(m) && c.n <= 16 && !(c.density > m.ad_constant_density))
        {
            ad::RangePre pre[16];
            float es[16 * HYAS_AD_EMIT_MAX], ea[16 * HYAS_AD_EMIT_MAX];
            int cnt[16];
            for (uint32_t r = 0; r < c.n; ++r)
                pre[r] = ad::range_pre(c.rng[r]);
            const float gx = goals_xyt[3 * i], gy = goals_xyt[3 * i + 1], gt = goals_xyt[3 * i + 2];
            const float toward = ad::toward_goal_steering(m, x, y, t, gx, gy, gt, dist_start_goal[i]);
            const float dist = normf(sub(f2(x, y), f2(gx, gy)));
            for (uint32_t r = 0; r < c.n; ++r)
                cnt[r] = ad::emit_range(m, c, pre, r, t, dist, toward, m.res_f, es + r * HYAS_AD_EMIT_MAX, ea + r * HYAS_AD_EMIT_MAX);
            float st2[HYAS_AD_MAX_PAIRS], sa2[HYAS_AD_MAX_PAIRS];
            ad::Pairs q;
            q.step = st2, q.steer = sa2;
            ad::merge_emitted(c, cnt, es, ea, q);
            if (q.n != p.n || q.overflow != p.overflow || std::memcmp(st, st2, sizeof(float) * p.n) != 0 || std::memcmp(sa, sa2, sizeof(float) * p.n) != 0)
                counts[i] = -1000; // no caller expects this value
            else
                ++*shared_out_checked;
        }
    }
}

extern "C" int hg_cut_to_circle(const float *in3, int n, float *out3)
{
    std::vector<ad::Range> v(n);
    for (int i = 0; i < n; ++i)
        v[i] = ad::Range{in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]};
    ad::cut_to_circle(v);
    for (size_t i = 0; i < v.size(); ++i)
        out3[3 * i] = v[i].dist, out3[3 * i + 1] = v[i].start, out3[3 * i + 2] = v[i].range;
    return (int)v.size();
}

// The huge-page pool behind the searches' host state: blocks of many sizes from several threads' worth of requests are
// distinct, 64-byte aligned, at least as large as asked and keep what is written into them; a freed block comes back for
// the next request of its size; trim() releases nothing while a block is in use and every region once none is.
// Returns the number of violations.
extern "C" long hg_hugepool_violations(const uint32_t *sizes, int n)
{
    HugePool &pool = HugePool::instance();
    long bad = 0;
    std::vector<std::pair<unsigned char *, size_t>> blocks;
    for (int i = 0; i < n; ++i)
    {
        const size_t bytes = sizes[i];
        size_t rounded = 0;
        const int cls = HugePool::class_of(bytes, rounded);
        if (bytes <= HugePool::BIGGEST && (rounded < bytes || cls < 0 || cls >= HugePool::N_CLASSES || rounded > bytes + bytes / 4 + 64))
            ++bad;
        unsigned char *p = (unsigned char *)pool.allocate(bytes);
        if (((uintptr_t)p & 63u) != 0)
            ++bad;
        std::memset(p, i & 0xFF, bytes);
        blocks.emplace_back(p, bytes);
    }
    // no two blocks overlap, and every block still holds its own byte
    std::vector<std::pair<uintptr_t, uintptr_t>> spans;
    for (int i = 0; i < n; ++i)
    {
        spans.emplace_back((uintptr_t)blocks[i].first, (uintptr_t)blocks[i].first + blocks[i].second);
        for (size_t k = 0; k < blocks[i].second; k += 509)
            if (blocks[i].first[k] != (unsigned char)(i & 0xFF))
                ++bad;
        if (blocks[i].second && blocks[i].first[blocks[i].second - 1] != (unsigned char)(i & 0xFF))
            ++bad;
    }
    std::sort(spans.begin(), spans.end());
    for (size_t i = 1; i < spans.size(); ++i)
        if (spans[i].first < spans[i - 1].second)
            ++bad;
    if (n > 0 && pool.trim() != 0) // blocks in use: nothing may be released
        ++bad;
    // a freed block is the next one handed out for its size
    if (n > 0)
    {
        unsigned char *p0 = blocks[0].first;
        const size_t b0 = blocks[0].second;
        pool.deallocate(p0, b0);
        unsigned char *again = (unsigned char *)pool.allocate(b0);
        if (b0 <= HugePool::BIGGEST && again != p0)
            ++bad;
        blocks[0].first = again;
    }
    for (auto &b : blocks)
        pool.deallocate(b.first, b.second);
    const size_t mapped = pool.mapped_bytes();
    if (pool.trim() != mapped + (mapped / HugePool::REGION) * HugePool::HUGE_PAGE || pool.mapped_bytes() != 0)
        ++bad;
    // and the pool works again afterwards, through the container alias too
    HVec<int> v;
    for (int i = 0; i < 100000; ++i)
        v.push_back(i);
    for (int i = 0; i < 100000; i += 997)
        if (v[i] != i)
            ++bad;
    return bad;
}
