// hyas_dev.h -- device-side data layout and the float geometry of the
// collision path (host+device, so the map-ingest code on the host evaluates
// the very same expressions).
//
// Arithmetic follows the reference expression by expression (operand order,
// float vs double promotion); citations give the reference file:line.  All
// elementary functions come from include/hyas_math.h; build with
// -ffp-contract=off.
#pragma once
#include <stdint.h>

#include "../../include/hyas.h"
#include "../../include/hyas_math.h"

#define HD __host__ __device__ __forceinline__
// out-of-line on the device: shared by every call site instead of being copied into each
#ifdef HYAS_INLINE_GEOM
#define HDC HD
#else
#define HDC __host__ __device__ inline __attribute__((noinline))
#endif

namespace hyas
{

struct F2
{
    float x, y;
};
HD F2 f2(float x, float y)
{
    F2 r;
    r.x = x;
    r.y = y;
    return r;
}
HD F2 sub(F2 a, F2 b) { return f2(a.x - b.x, a.y - b.y); }
HD F2 add(F2 a, F2 b) { return f2(a.x + b.x, a.y + b.y); }
HD float dotf(F2 a, F2 b) { return a.x * b.x + a.y * b.y; }         // Eigen dot()
HD float normf(F2 a) { return hy_sqrtf(a.x * a.x + a.y * a.y); }     // Eigen norm()
HD float crossf(F2 a, F2 b) { return a.x * b.y - b.x * a.y; }        // src/utility.cpp:273-279
HD float fmaxf_(float a, float b) { return a < b ? b : a; }          // std::max
HD float fminf_(float a, float b) { return b < a ? b : a; }          // std::min

// One obstacle quadrilateral: vertices 0..3 (vertex 4 == vertex 0 in the
// reference's closed polygon) and its axis-aligned bounds.
struct alignas(16) Quad
{
    float vx[4], vy[4];
    float xmin, ymin, xmax, ymax;
    // 1: the vertices are exactly (xmin,ymin) (xmax,ymin) (xmax,ymax) (xmin,ymax)
    // in this order -- what CreatePolygon(origin, w, h) and vertical merges produce
    uint32_t axis_rect;
    uint32_t pad[3];
};
HD F2 qv(const Quad &q, int i) { return f2(q.vx[i & 3], q.vy[i & 3]); }

// Everything a kernel needs to know about the map and the planner settings.
// one packed footprint list: byte 0 = len, then (dx, dy) as int8 pairs
struct alignas(16) FpPack
{
    uint32_t w[4];
};

struct DevMap
{
    // costmap_2d state (double, as Costmap2D keeps it)
    const uint8_t *charmap;
    uint32_t w, h;
    double res, ox, oy;
    // CollisionDetection members (float)
    float res_f, ox_f, oy_f;
    // resolution a power of two: dividing by it equals multiplying by its
    // (exact) inverse, bit for bit, and the division can be skipped
    int32_t res_pow2;
    double inv_res;
    float inv_res_f;
    // obstacles and per-grid-key lists (key = GetNode3DIndexOnGridMap)
    const Quad *quads;
    uint32_t n_quads;
    const uint32_t *cell_off; // w*h + 1 entries
    const uint16_t *cell_idx;
    // The same lists with every entry removed that provably cannot change a
    // verdict (see prune_lists() on the host): pose_* for the vehicle-polygon
    // test keyed by the cell of corner 0, seg_* for the swept-segment test of
    // segments no longer than seg_lmax per axis.  Bit 15 of a seg entry: only
    // the ray-casting inside test can still fire for this quad.
    const uint32_t *pose_off;
    const uint16_t *pose_idx;
    const uint32_t *seg_off;
    const uint16_t *seg_idx;
    float seg_lmax; // 0: pruned lists unavailable, walk the full ones
    // vehicle
    float veh_len, veh_wid;
    double half_wid;     // 0.5 * vehicle_width, src/collisiondetection.cpp:626
    float cull_margin;   // see seg_hits_quad()
    // heuristic table (dense by reference key; NaN = key never emplaced)
    const float *table;
    uint64_t table_size;
    float table_miss;
    int32_t pr, headings;
    float delta_heading; // (float)(2*M_PI/(float)headings)
    // footprint-list mode
    const int32_t *fp;   // entries x {len, pos[64][2]}
    // the same data in the form the batched footprint kernel gathers from:
    //  occ_tiles: one bit per cell (1: cost >= 250), 8x8 cells per 64-bit word, word (Y>>3)*tiles_x + (X>>3),
    //             bit ((Y&7)<<3)|(X&7): 1/8 of the byte grid, so a 4096x4096 map stays in one L2; bits of cells
    //             beyond the grid's last row / column are 1 (off the grid collides)
    //  fp_pack:   16 bytes per list {len, (dx, dy) as int8 x 7}; null when a list is longer or an offset wider
    //  fp_win:    every list as a bitmap over the fp_bb x fp_bb cell box it was rasterised in (bit i * fp_bb + j =
    //             cell (j, i) of the box), fp_win_words 32-bit words per list; the two bits after the bitmap say
    //             whether the box sits fp_bb / 2 or fp_bb / 2 + 1 cells left of / below the pose's cell; null when
    //             the box is wider than 9 cells (a box of up to 9 cells spans at most 2 x 2 bit tiles)
    //  fp_m8:     (boxes of up to 7 cells) every list as an 8 x 8 bitmap, byte i = row i of the box, bit j = cell (j, i);
    //             bits 62 / 63: the box sits one cell further left / down (as in fp_win); occ_pad: the tiles with a
    //             border of all-ones tiles, one to the left and above, two to the right and below:
    //             tile (tx, ty) at (ty + 1) * (tiles_x + 3) + tx + 1 for -1 <= tx <= tiles_x + 1
    const uint64_t *occ_tiles;
    const uint64_t *occ_pad;
    const uint64_t *fp_m8;
    uint32_t tiles_x, tiles_y;
    const uint32_t *fp_win;
    int32_t fp_bb, fp_win_words;
    const FpPack *fp_pack;
    uint32_t fp_entries;
    int32_t fp_max_len;
    // motion set
    int32_t n_succ;
    float succ_steer[HYAS_MAX_SUCC];
    int32_t succ_reverse[HYAS_MAX_SUCC];
    float pen_turn, pen_rev, pen_cod;
    float shot_rho; // params.min_turning_radius
    float max_steer; // params.max_steering_angle: ShortCut keeps a node whose steering towards the anchor is not below it
    int32_t curve_type;
    uint32_t twod_size; // floats per 2D cost map: w*h + w + 1
    // Dubins shot: sample positions x_k of `x += 0.2 * res` in float (src/hybrid_a_star.cpp:396-424)
    const float *seg_db;
    uint32_t seg_db_cap;
    // cubic Bezier shot: the same for `x += 0.1 * res` (src/hybrid_a_star.cpp:477-499)
    const float *seg_bz;
    uint32_t seg_bz_cap;
    // adaptive step size / steering (hyas_adaptive.h): 0 off, 1 adaptive_steering_angle_and_step_size, 2 adaptive_step_size
    int32_t ad_mode;
    int32_t n_slots;         // successor slots per node in the step's output: n_succ, or HYAS_ADAPTIVE_SLOTS in mode 1
    const uint32_t *ad_off;  // w*h + 1 entries: ranges of grid key k are ad_rng[3 * ad_off[k] .. 3 * ad_off[k + 1])
    const float *ad_rng;     // {distance, start, range} triples
    const float *ad_density; // normalised obstacle density per grid key
    float ad_range;          // obstacle_detection_range: the distance that marks a free range
    float ad_constant_density;
    int32_t ad_one_more, ad_one_more_free_only, ad_towards_free, ad_fixed_number;
};

// ---------------------------------------------------------------- Utility::
// src/utility.cpp:1180-1185
HD float deg2rad(float deg) { return (float)((double)(deg / 360 * 2) * HY_PI); }

// src/utility.cpp:1249-1268
HD float rad_0_2pi(float rad)
{
    const double two_pi = (double)2.f * HY_PI;
    if ((double)rad >= 0 && (double)rad < two_pi)
        return rad;
    float out = (float)hy_fmod((double)rad, two_pi);
    if (out < 0)
        out = (float)((double)out + two_pi);
    if (hy_fabs((double)out - 2 * HY_PI) < 1e-3)
        out = 0;
    return out;
}

// src/utility.cpp:1210-1231
HD float rad_norm(float rad)
{
    if ((double)rad >= -HY_PI && (double)rad <= HY_PI)
        return rad;
    const double two_pi = (double)2.f * HY_PI;
    float out = (float)hy_fmod((double)rad, two_pi);
    if (out < 0)
        out = (float)((double)out + two_pi);
    if ((double)out > HY_PI)
        out = (float)((double)out - (double)2.0f * HY_PI);
    if ((double)hy_fabsf(out) < 1e-6)
        out = 0;
    return out;
}

// src/utility.cpp:209-249: is q on segment [a, b]?
HD bool on_segment(F2 a, F2 b, F2 q)
{
    const bool in_y = q.y <= fmaxf_(a.y, b.y) && q.y >= fminf_(a.y, b.y);
    const bool in_x = q.x <= fmaxf_(a.x, b.x) && q.x >= fminf_(a.x, b.x);
    if ((double)hy_fabsf(q.x - a.x) < 1e-3 && (double)hy_fabsf(q.x - b.x) < 1e-3 && in_y)
        return true;
    if (in_x && (double)hy_fabsf(q.y - a.y) < 1e-3 && (double)hy_fabsf(q.y - b.y) < 1e-3)
        return true;
    if (in_x && in_y)
    {
        F2 aq = sub(q, a), ab = sub(b, a);
        return (double)hy_fabsf(dotf(aq, ab) / normf(aq) / normf(ab) - 1) < 1e-6;
    }
    return false;
}

// src/utility.cpp:251-271
HD bool point_on_line(F2 p, F2 start, F2 unit)
{
    F2 d = sub(p, start);
    if ((double)hy_fabsf(d.x) < 1e-3 && (double)hy_fabsf(d.y) < 1e-3)
        return true;
    float angle = hy_acosf(dotf(d, unit) / normf(d) / normf(unit));
    return (double)hy_fabsf(angle) < 1e-3 || hy_fabs((double)angle - HY_PI) < 1e-3;
}

// src/utility.cpp:280-306
HD bool seg_intersect(F2 p1, F2 p2, F2 p3, F2 p4)
{
    if ((p1.x == p2.x && p2.x == p3.x && p3.x == p4.x) || (p1.y == p2.y && p2.y == p3.y && p3.y == p4.y))
        return false;
    if (fmaxf_(p3.x, p4.x) < fminf_(p1.x, p2.x) || fmaxf_(p3.y, p4.y) < fminf_(p1.y, p2.y) ||
        fmaxf_(p1.x, p2.x) < fminf_(p3.x, p4.x) || fmaxf_(p1.y, p2.y) < fminf_(p3.y, p4.y))
        return false;
    F2 d43 = sub(p4, p3), d12 = sub(p1, p2);
    return crossf(sub(p1, p3), d43) * crossf(sub(p2, p3), d43) <= 0 &&
           crossf(sub(p3, p2), d12) * crossf(sub(p4, p2), d12) <= 0;
}

// src/utility.cpp:308-327 (caller has already established intersection)
HD F2 intersection_point(F2 p1, F2 p2, F2 p3, F2 p4)
{
    F2 d1 = sub(p2, p1), d2 = sub(p4, p3);
    float t = crossf(sub(p3, p1), d2) / crossf(d1, d2);
    return add(p1, f2(t * d1.x, t * d1.y));
}

// src/utility.cpp:328-374, returns 1 for above/on, 0 for below
HD int above_segment(F2 p1, F2 p2, F2 p3)
{
    if (on_segment(p1, p2, p3))
        return 1;
    F2 d = sub(p2, p1);
    if (d.x == 0)
        return p3.x > p1.x ? 0 : 1;
    if (d.y == 0)
        return p3.y > p1.y ? 1 : 0;
    float k = d.y / d.x;
    float b = p1.y - k * p1.x;
    return (k * p3.x + b - p3.y < 0) ? 1 : 0;
}

// src/utility.cpp:375-422: ray casting towards (10000, p.y)
// (out-of-line routines take the vertices one by one: a 64-byte Quad by value would travel through the stack)
HDC bool inside_verts(F2 q0, F2 q1, F2 q2, F2 q3, F2 p)
{
    const F2 v[4] = {q0, q1, q2, q3};
    F2 far_pt = f2(10000, p.y);
    int crossings = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        F2 a = v[i], b = v[(i + 1) & 3];
        if (seg_intersect(p, far_pt, a, b))
        {
            F2 ip = intersection_point(p, far_pt, a, b);
            if ((double)normf(sub(a, ip)) < 1e-3)
                crossings += (above_segment(p, far_pt, b) == 0);
            else if ((double)normf(sub(b, ip)) < 1e-3)
                crossings += (above_segment(p, far_pt, a) == 0);
            else
                crossings += 1;
        }
    }
    return (crossings & 1) != 0;
}
HD bool inside_quad(const Quad &q, F2 p) { return inside_verts(qv(q, 0), qv(q, 1), qv(q, 2), qv(q, 3), p); }

// src/utility.cpp:1591-1596: Eigen normalized()
HD F2 unit_vector(F2 a, F2 b)
{
    F2 d = sub(b, a);
    float z = d.x * d.x + d.y * d.y;
    if (z > 0)
    {
        float n = hy_sqrtf(z);
        return f2(d.x / n, d.y / n);
    }
    return d;
}

// src/utility.cpp:578-592
HD float dist_point_line(F2 p, F2 start, F2 unit)
{
    if (point_on_line(p, start, unit))
        return 0;
    F2 d = sub(p, start);
    float proj = hy_fabsf(dotf(d, unit));
    float n = normf(d);
    return hy_sqrtf(n * n - proj * proj);
}

// src/utility.cpp:541-576
HDC float dist_seg_point(F2 a, F2 b, F2 p)
{
    if (on_segment(a, b, p))
        return 0;
    F2 unit = unit_vector(a, b);
    float da = normf(sub(a, p)), db = normf(sub(b, p));
    F2 far_pt = da > db ? a : b;
    F2 ab = sub(b, a);
    float len = normf(ab);
    float proj = hy_fabsf(dotf(sub(p, far_pt), ab)) / len;
    if (proj > len)
        return fminf_(da, db);
    return dist_point_line(p, a, unit);
}

// src/utility.cpp:594-619
HDC float dist_quad_point(const Quad &q, F2 p)
{
    for (int i = 0; i < 4; ++i)
        if (on_segment(qv(q, i), qv(q, i + 1), p))
            return 0;
    if (inside_quad(q, p))
        return -1;
    float best = 100000;
    for (int i = 0; i < 4; ++i)
    {
        float d = dist_seg_point(qv(q, i), qv(q, i + 1), p);
        if (d < best)
            best = d;
    }
    return best;
}

// src/utility.cpp:967-997
HDC float dist_quad_segment(const Quad &q, F2 s, F2 e)
{
    for (int i = 0; i < 4; ++i)
        if (seg_intersect(qv(q, i), qv(q, i + 1), s, e))
            return -1;
    float out = 100000;
    float d = dist_quad_point(q, s);
    if (out > d)
        out = d;
    d = dist_quad_point(q, e);
    if (out > d)
        out = d;
    for (int i = 0; i < 5; ++i)
    {
        d = dist_seg_point(s, e, qv(q, i));
        if (out > d)
            out = d;
    }
    return out;
}

// ------------------------------------------------------- costmap / indices
// costmap_2d::Costmap2D::worldToMap (ROS dependency, restated: reject below the
// origin, truncate (w - origin)/resolution, accept iff inside the size).
HD bool world_to_map(const DevMap &m, float wx, float wy, uint32_t &mx, uint32_t &my)
{
    double x = (double)wx, y = (double)wy;
    if (x < m.ox || y < m.oy)
        return false;
    if (m.res_pow2)
    {
        mx = (uint32_t)(int)((x - m.ox) * m.inv_res);
        my = (uint32_t)(int)((y - m.oy) * m.inv_res);
    }
    else
    {
        mx = (uint32_t)(int)((x - m.ox) / m.res);
        my = (uint32_t)(int)((y - m.oy) / m.res);
    }
    return mx < m.w && my < m.h;
}
HD bool on_grid(const DevMap &m, float x, float y)
{
    uint32_t a, b;
    return world_to_map(m, x, y, a, b);
}

// floor(round(v*1000)/1000), the index rounding of include/node3d.h:67
HD float idx_round(float v) { return hy_floorf(hy_roundf(v * 1000) / 1000); }

// (v - origin) / resolution in float
HD float cell_coord(const DevMap &m, float v, float origin)
{
    return m.res_pow2 ? (v - origin) * m.inv_res_f : (v - origin) / m.res_f;
}
// src/collisiondetection.cpp:362-369
HD uint32_t grid_key(const DevMap &m, float x, float y)
{
    return (uint32_t)(idx_round(cell_coord(m, y, m.oy_f)) * m.w + idx_round(cell_coord(m, x, m.ox_f)));
}
// include/node2d.h:59-66
HD int idx_2d(const DevMap &m, float x, float y)
{
    return (int)(idx_round(cell_coord(m, y, m.oy_f)) * (int)m.w + idx_round(cell_coord(m, x, m.ox_f)));
}
// include/node3d.h:65-72 with delta = (float)((float)2*M_PI/headings)
HD int idx_3d(const DevMap &m, float x, float y, float t)
{
    const float delta = (float)((double)(float)2 * HY_PI / m.headings);
    return (int)(idx_round(t / delta) * (int)m.w * (int)m.h + idx_round(cell_coord(m, y, m.oy_f)) * (int)m.w +
                 idx_round(cell_coord(m, x, m.ox_f)));
}

// -------------------------------------------------- collision, polygon mode
// src/utility.cpp:476-510: corners of the vehicle rectangle at (x, y, t)
HD void vehicle_corners(const DevMap &m, float x, float y, float t, F2 c[4])
{
    const float w = m.veh_len, h = m.veh_wid; // CreatePolygon(center, vehicle_length, vehicle_width, t)
    float ct, st;
    hy_sincosf(t, &st, &ct);
    const float m01 = -st;
    const float px[4] = {-w / 2, +w / 2, +w / 2, -w / 2};
    const float py[4] = {-h / 2, -h / 2, +h / 2, +h / 2};
    for (int i = 0; i < 4; ++i)
        c[i] = f2(x + (ct * px[i] + m01 * py[i]), y + (st * px[i] + ct * py[i]));
}

// Where the kernels read obstacle quads from: global memory (default) or a
// workgroup-local copy.  Fetch returns the quad by value so the out-of-line
// geometry routines below work on registers, not on pointers.
// stop() lets a caller whose overall answer is already decided (another lane
// found a collision) cut the list walks short; hit() announces a collision.
// Cursor hands out the entries idx[k] of one obstacle list for k = begin, begin + 1, ...
struct GlobalQuads
{
    const Quad *p;
    HD Quad operator()(uint32_t i) const { return p[i]; }
    HD bool stop() const { return false; }
    HD void hit() const {}
    struct Cursor
    {
        const uint16_t *idx;
        HD Cursor(const uint16_t *i, uint32_t) : idx(i) {}
        HD uint16_t at(uint32_t k) { return idx[k]; }
    };
};

// src/utility.cpp:956-966 + :528-540 for a 4-corner vehicle against one quad.
// The bounding-box test is exact: when the boxes are strictly apart every
// edge pair fails IsIntersect's own box test.
HDC bool vehicle_hits_verts(F2 c0, F2 c1, F2 c2, F2 c3, F2 q0, F2 q1, F2 q2, F2 q3)
{
    const F2 c[4] = {c0, c1, c2, c3};
    const F2 v[4] = {q0, q1, q2, q3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const F2 a = c[i], b = c[(i + 1) & 3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (seg_intersect(v[j], v[(j + 1) & 3], a, b))
                return true;
    }
    return false;
}
HD bool vehicle_hits_quad(F2 c0, F2 c1, F2 c2, F2 c3, const Quad &q)
{
    return vehicle_hits_verts(c0, c1, c2, c3, qv(q, 0), qv(q, 1), qv(q, 2), qv(q, 3));
}

// The part of the pose test that reads one byte: a pose whose centre lies off the grid or in a cell of cost >= 250
// is in collision whatever its heading (src/collisiondetection.cpp:189-203).  pose_free() below comes to the same
// verdict for such a pose; callers that only need to know whether ANY pose of a set collides ask this first.
HD bool centre_blocked(const DevMap &m, float x, float y)
{
    uint32_t sx = 0, sy = 0;
    if (!world_to_map(m, x, y, sx, sy))
        return true;
    return m.charmap[sx + sy * m.w] >= 250;
}

// src/collisiondetection.cpp:189-214 + :573-606.  true = free.
// (sub, nsub): this caller walks entries sub, sub + nsub, ... of the obstacle list only; the pose is free iff all nsub
// callers that share it say so (the verdict is an AND over the list's polygons).
template <class QuadSrc>
HD bool pose_free(const DevMap &m, const QuadSrc &quads, float x, float y, float t, uint32_t sub = 0, uint32_t nsub = 1)
{
    // Everything that decides which memory is read comes first, then the cost cell and the list bounds are
    // requested together (two trips to memory in a row instead of three); the verdict logic follows in the
    // reference's order.
    uint32_t sx = 0, sy = 0;
    const bool centre_on = world_to_map(m, x, y, sx, sy);
    F2 c[4];
    vehicle_corners(m, x, y, t, c);
    float bxmin = c[0].x, bxmax = c[0].x, bymin = c[0].y, bymax = c[0].y;
    bool corners_on = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        corners_on = corners_on && on_grid(m, c[i].x, c[i].y);
        bxmin = fminf_(bxmin, c[i].x), bxmax = fmaxf_(bxmax, c[i].x);
        bymin = fminf_(bymin, c[i].y), bymax = fmaxf_(bymax, c[i].y);
    }
    const uint32_t key = grid_key(m, c[0].x, c[0].y);
    const bool listed = centre_on && corners_on && key < m.w * m.h;
    const uint32_t *off = m.seg_lmax > 0 ? m.pose_off : m.cell_off;
    const uint16_t *idx = m.seg_lmax > 0 ? m.pose_idx : m.cell_idx;
    const uint8_t cost = centre_on ? m.charmap[sx + sy * m.w] : (uint8_t)255;
    const uint32_t begin = listed ? off[key] : 0u, e = listed ? off[key + 1] : 0u;
    if (!centre_on || cost >= 250 || !corners_on)
        return false;
    if (!listed)
        return true; // key absent from in_range_obstacle_map_: nothing to test
    typename QuadSrc::Cursor cur(idx, begin);
#pragma unroll 1
    for (uint32_t k = begin + sub; k < e; k += nsub)
    {
        if (quads.stop())
            return true; // the caller no longer needs this verdict
        const Quad q = quads(cur.at(k));
        if (q.xmax < bxmin || q.ymax < bymin || bxmax < q.xmin || bymax < q.ymin)
            continue;
        if (vehicle_hits_quad(c[0], c[1], c[2], c[3], q))
        {
            quads.hit();
            return false;
        }
    }
    return true;
}

// inside_quad() for a point already known to have q.ymin <= p.y <= q.ymax and
// p.x <= q.xmax.  For an axis-aligned rectangle and a point whose y is more
// than 2e-3 away from both horizontal edges the ray to (10000, p.y) meets the
// two horizontal edges nowhere (their box test fails) and each vertical edge
// right of p exactly once, away from the vertices (the vertex rules of
// src/utility.cpp:387-402 act within 1e-3): the count is 2 left of the
// rectangle and 1 inside it.  Within 1e-2 of a vertical edge the exact routine
// decides.  Returns 0 outside, 1 inside, -1 undecided.
HD int inside_quad_fast(const Quad &q, F2 p)
{
    if (q.axis_rect && p.y > q.ymin + 2e-3f && p.y < q.ymax - 2e-3f)
    {
        if (p.x < q.xmin - 1e-2f)
            return 0;
        if (p.x > q.xmin + 1e-2f && p.x < q.xmax - 1e-2f)
            return 1;
    }
    return -1;
}

// Does GetDistanceFromPolygonToSegment(q, s, e) come out <= half_wid
// (src/collisiondetection.cpp:626)?  Exact, but evaluated as a predicate.
//
// The reference takes the minimum over: -1 if an edge crosses the segment;
// for each segment end p: 0 if p is on an edge, -1 if the ray-casting
// IsInsidePolygon says inside, else the distances from p to the four edges; and
// the distances from the (five, first repeated) vertices to the segment.  The
// minimum is <= half_wid iff ANY candidate is, so candidates are tested one by
// one and each is skipped when a box test proves it cannot fire:
//  * a point-to-edge distance is a true Euclidean distance up to (a) float
//    rounding, (b) the OnSegment / IsPointOnLine snaps, which turn distances
//    below 1e-3 m or below 1e-3 rad x (distance to the edge start) into 0, and
//    (c) cancellation in sqrt(n^2 - proj^2), about 1e-3 / distance.  All are
//    covered by cull_margin (host: 0.01 + 2e-3 x map diagonal), so a point
//    outside the edge's box grown by half_wid + cull_margin is farther than
//    half_wid whatever the reference computes (NaN candidates never count);
//  * IsInsidePolygon (src/utility.cpp:375-422) can say "inside" for a point far
//    to the LEFT of the quad when its y lies in the quad's y-range (vertex
//    rules), so it is evaluated for every such end point, near or far.
// Assumes 0 <= half_wid (hyas_create rejects a negative vehicle width).
HD bool near_box(F2 p, F2 a, F2 b, float r)
{
    return !(p.x < fminf_(a.x, b.x) - r || p.x > fmaxf_(a.x, b.x) + r || p.y < fminf_(a.y, b.y) - r || p.y > fmaxf_(a.y, b.y) + r);
}
// the edge part: the quad's box is within r of the segment's box
HDC bool seg_hits_verts_near(float r, double half_wid, F2 s, F2 e, F2 q0, F2 q1, F2 q2, F2 q3)
{
    const F2 v[4] = {q0, q1, q2, q3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const F2 a = v[i], b = v[(i + 1) & 3];
        if (seg_intersect(a, b, s, e))
            return true;
        if (near_box(s, a, b, r) && (double)dist_seg_point(a, b, s) <= half_wid)
            return true;
        if (near_box(e, a, b, r) && (double)dist_seg_point(a, b, e) <= half_wid)
            return true;
        if (near_box(a, s, e, r) && (double)dist_seg_point(s, e, a) <= half_wid)
            return true;
    }
    return false;
}
// the ray-casting part
HD bool seg_ray_hits_quad(F2 s, F2 e, const Quad &q)
{
    if (s.y >= q.ymin && s.y <= q.ymax && s.x <= q.xmax)
    {
        int f = inside_quad_fast(q, s);
        if (f > 0 || (f < 0 && inside_quad(q, s)))
            return true;
    }
    if (e.y >= q.ymin && e.y <= q.ymax && e.x <= q.xmax)
    {
        int f = inside_quad_fast(q, e);
        if (f > 0 || (f < 0 && inside_quad(q, e)))
            return true;
    }
    return false;
}
// LONG: the segment may be metres long (adaptive step sizes, ShortCut), so that its box holds many quads it passes at
// a distance: a quad whose four vertices lie on one side of the segment's line, farther than r from it, has no edge
// that crosses the segment, no edge within r of a segment end (the ends are on the line) and no vertex within r of the
// segment: none of the candidates above can fire (r = half_wid + cull_margin covers the reference's rounding and snaps
// as for the box tests; the products below are exact to a few 1e-5 m at map scale).
template <bool LONG = false>
HD bool seg_hits_quad(float r, double half_wid, F2 s, F2 e, const Quad &q)
{
    if (seg_ray_hits_quad(s, e, q))
        return true;
    const float sxmin = fminf_(s.x, e.x) - r, sxmax = fmaxf_(s.x, e.x) + r;
    const float symin = fminf_(s.y, e.y) - r, symax = fmaxf_(s.y, e.y) + r;
    if (q.xmax < sxmin || sxmax < q.xmin || q.ymax < symin || symax < q.ymin)
        return false;
    if (LONG)
    {
        const float nx = s.y - e.y, ny = e.x - s.x; // normal of the line, as long as the segment
        const float far = r * hy_sqrtf(nx * nx + ny * ny) * 1.0001f;
        float lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const F2 v = qv(q, i);
            const float d = nx * (v.x - s.x) + ny * (v.y - s.y);
            lo = i ? fminf_(lo, d) : d, hi = i ? fmaxf_(hi, d) : d;
        }
        if (lo > far || hi < -far)
            return false;
    }
    return seg_hits_verts_near(r, half_wid, s, e, qv(q, 0), qv(q, 1), qv(q, 2), qv(q, 3));
}

// src/collisiondetection.cpp:608-656.  true = free.  (sub, nsub) as for pose_free().
template <class QuadSrc>
HD bool segment_free(const DevMap &m, const QuadSrc &quads, float sx, float sy, float ex, float ey, uint32_t sub = 0, uint32_t nsub = 1)
{
    if (!on_grid(m, sx, sy) || !on_grid(m, ex, ey))
        return false;
    const F2 s = f2(sx, sy), e = f2(ex, ey);
    const uint32_t ks = grid_key(m, sx, sy), ke = grid_key(m, ex, ey);
    const uint32_t cells = m.w * m.h;
    const float r = (float)m.half_wid + m.cull_margin;
    const bool pruned = hy_fabsf(sx - ex) <= m.seg_lmax && hy_fabsf(sy - ey) <= m.seg_lmax;
    const uint32_t *off = pruned ? m.seg_off : m.cell_off;
    const uint16_t *idx = pruned ? m.seg_idx : m.cell_idx;
    // the bounds of both lists are requested together; same key: same list, same verdicts
    const bool walk0 = ks < cells, walk1 = ke < cells && ke != ks;
    const uint32_t b0 = walk0 ? off[ks] : 0u, e0 = walk0 ? off[ks + 1] : 0u;
    const uint32_t b1 = walk1 ? off[ke] : 0u, e1 = walk1 ? off[ke + 1] : 0u;
    // A long segment reads the full lists of two cells that lie metres apart at most: mostly the same quads.  The
    // verdict is an OR over quads, so the second walk passes over the ones the first has tested (quad numbers below
    // 192, the ones kept in workgroup memory, are remembered in three words).
    uint64_t seen0 = 0, seen1 = 0, seen2 = 0;
    const bool remember = !pruned && walk0 && walk1;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass)
    {
        const uint32_t begin = pass ? b1 : b0, end = pass ? e1 : e0;
        if (begin >= end)
            continue;
        typename QuadSrc::Cursor cur(idx, begin);
#pragma unroll 1
        for (uint32_t k = begin + sub; k < end; k += nsub)
        {
            if (quads.stop())
                return true; // the caller no longer needs this verdict
            const uint16_t ent = cur.at(k);
            if (remember && (ent & 0x7fffu) < 192u)
            {
                const uint32_t id = ent & 0x7fffu;
                const uint64_t bit = 1ull << (id & 63u);
                uint64_t &word = id < 64u ? seen0 : (id < 128u ? seen1 : seen2);
                if (pass && (word & bit))
                    continue;
                word |= bit;
            }
            const Quad q = quads(ent & 0x7fffu);
            // bit 15 (pruned lists only): only the ray-casting test can fire
            if (pruned ? ((ent & 0x8000u) ? seg_ray_hits_quad(s, e, q) : seg_hits_quad<false>(r, m.half_wid, s, e, q))
                       : seg_hits_quad<true>(r, m.half_wid, s, e, q))
            {
                quads.hit();
                return false;
            }
        }
    }
    return true;
}

// src/collisiondetection.cpp:93-127
template <class QuadSrc>
HD bool traversable(const DevMap &m, const QuadSrc &quads, float x, float y, float t, bool has_pred, float px, float py)
{
    if (!on_grid(m, x, y))
        return false;
    if (!pose_free(m, quads, x, y, t))
        return false;
    return has_pred ? segment_free(m, quads, x, y, px, py) : true;
}

// ------------------------------------------------ collision, footprint mode
// Pose test over the per-heading footprint cell list (table layout of
// include/constants.h:62-80, built by include/lookup.h:30-284): truncate the
// pose to its cell, sub-cell bin and heading bin, then every listed cell must
// be on the grid and below the lethal threshold (>= 250 collides, as in
// src/collisiondetection.cpp:200).
HD bool footprint_free(const DevMap &m, float x, float y, float t)
{
    const int pr = m.pr;
    if (!(x == x) || !(y == y) || !(t == t))
        return false; // NaN: the reference's int conversions give INT_MIN, i.e. off the grid
    int X = (int)x, Y = (int)y;
    int iX = (int)((x - (long)x) * pr);
    iX = iX > 0 ? iX : 0;
    int iY = (int)((y - (long)y) * pr);
    iY = iY > 0 ? iY : 0;
    int iT = (int)(t / m.delta_heading);
    if (iT < 0 || iT >= m.headings || iX >= pr || iY >= pr)
        return false;
    const int32_t *e = m.fp + (size_t)((iY * pr + iX) * m.headings + iT) * 129;
    int n = e[0] < 64 ? e[0] : 64;
    for (int i = 0; i < n; ++i)
    {
        int cX = X + e[1 + 2 * i], cY = Y + e[2 + 2 * i];
        if (cX < 0 || (uint32_t)cX >= m.w || cY < 0 || (uint32_t)cY >= m.h)
            return false;
        if (m.charmap[(size_t)cY * m.w + cX] >= 250)
            return false;
    }
    return true;
}

// ------------------------------------------------------- heuristic lookups
// src/lookup_table.cpp:123-146
HD int table_key(const DevMap &m, float x, float y, float theta)
{
    float angle = ((double)hy_fabsf(theta) < 1e-4) ? 0 : theta;
    int x_index = (int)(x * m.pr);
    int y_index = (int)(y * m.pr);
    return (int)(angle / m.delta_heading) * (int)m.w * (int)m.h * m.pr * m.pr + (int)(y_index * m.pr) * (int)m.w + x_index;
}
// src/lookup_table.cpp:56-96, :113-122
HD float table_cost(const DevMap &m, float x, float y, float t)
{
    int key = table_key(m, x, y, rad_0_2pi(t));
    if (key >= 0 && (uint64_t)key < m.table_size)
    {
        float v = m.table[key];
        if (v == v)
            return v;
    }
    return m.table_miss;
}
// src/hybrid_a_star.cpp:316-366; twod = this goal's 2D cost map (-1 = absent) or null
HD float heuristic(const DevMap &m, float x, float y, float t, float gx, float gy, float gt, const float *twod)
{
    float curve = table_cost(m, hy_fabsf(gx - x), hy_fabsf(gy - y), gt - t);
    float two_d = 0;
    if (twod)
    {
        int id = idx_2d(m, x, y);
        if (id >= 0 && id < (int)m.twod_size)
        {
            float v = twod[id];
            if (v >= 0)
                two_d = v;
        }
    }
    return fmaxf_(curve, two_d);
}

// Utility::FindSteeringAngle(closest, direction), src/utility.cpp:2111-2117, over GetAngle (:1288-1296): the signed angle
// from closest's heading to the direction in which `direction` lies
HD float find_steering_angle(float cx, float cy, float ct, float dx, float dy)
{
    const float angle_between = rad_norm(rad_0_2pi(hy_atan2f(dy - cy, dx - cx)));
    return -rad_norm(rad_norm(ct) - angle_between);
}

// ------------------------------------------------------------- successors
// src/hybrid_a_star.cpp:659-756 for motion-set entry k of node (x,y,t).
// Returns false when the step is too small to create a successor (:681).
// One {step size, steering angle} pair of CreateSuccessor(pred, vector), src/hybrid_a_star.cpp:667-741.
HD bool successor_from_pair(float x, float y, float t, float first, float steer, float &sx, float &sy, float &st, int &prim)
{
    const bool reverse_flag = first < 0;
    if ((double)hy_fabsf(first) <= 1e-3)
        return false;
    const float radius = first / hy_fabsf(steer);
    float dx, dy;
    if (steer < 0)
    {
        dx = radius * hy_sinf(hy_fabsf(steer));
        dy = -(radius * (1 - hy_cosf(steer)));
        prim = reverse_flag ? 4 : 1;
    }
    else if ((double)steer > 1e-3)
    {
        dx = radius * hy_sinf(hy_fabsf(steer));
        dy = (radius * (1 - hy_cosf(steer)));
        prim = reverse_flag ? 5 : 2;
    }
    else
    {
        dx = first;
        dy = 0;
        prim = reverse_flag ? 3 : 0;
    }
    float ct, sn;
    hy_sincosf(t, &sn, &ct);
    sx = x + dx * ct - dy * sn;
    sy = y + dx * sn + dy * ct;
    st = rad_0_2pi(t + steer);
    return true;
}
// step of the fixed motion set (:606-614) or of the reverse pair (:634-642)
HD float motion_step(bool rev, float dist_to_goal)
{
    if (!rev)
    {
        float step = 0.2f;
        return (step > dist_to_goal) ? dist_to_goal : step;
    }
    float step = -0.2f;
    return (hy_fabsf(step) > dist_to_goal) ? -dist_to_goal : step;
}
HD bool successor_pose(const DevMap &m, float x, float y, float t, float dist_to_goal, int k, float &sx, float &sy, float &st,
                       int &prim)
{
    return successor_from_pair(x, y, t, motion_step(m.succ_reverse[k] != 0, dist_to_goal), m.succ_steer[k], sx, sy, st, prim);
}

// src/hybrid_a_star.cpp:761-811
HD float cost_so_far(const DevMap &m, float g, float step, int prim, int pre_prim)
{
    if (prim < 3)
    {
        if (pre_prim != prim)
            g += (pre_prim > 2) ? step * m.pen_turn * m.pen_cod : step * m.pen_turn;
        else
            g += step;
    }
    else
    {
        if (pre_prim != prim)
            g += (pre_prim < 3) ? step * m.pen_turn * m.pen_rev * m.pen_cod : step * m.pen_turn * m.pen_rev;
        else
            g += step * m.pen_rev;
    }
    return g;
}

} // namespace hyas
