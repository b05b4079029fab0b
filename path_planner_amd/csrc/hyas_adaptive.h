// Adaptive step size and steering angle of the successor generation
// (HybridAStar::CreateSuccessor, src/hybrid_a_star.cpp:540-600, with CollisionDetection::FindStepSizeAndSteeringAngle,
// SelectStepSizeAndSteeringAngle, FindStepSize and their helpers, src/collisiondetection.cpp:688-938, :1263-1484, and the
// angle-range algebra of src/utility.cpp:771-827, :1298-1330, :1465-1492, :1737-1849), shared by host and device code.
//
// What depends on the map alone is built once per costmap and read here through DevMap:
//   ad_off / ad_rng  per grid key the {distance, start, range} triples of distance_angle_range_map_ that survive the
//                    common-range filter of FindFreeAngleRangeAndObstacleAngleRange (it does not depend on the node's
//                    heading: the "steering range" it intersects with is the full circle less 0.0001 degrees);
//   ad_density       per grid key the normalised obstacle density (0 for a key the reference's map does not hold).
// What depends on the node runs per node: step_steer_pairs() returns the {step size, steering angle} list the reference
// hands to CreateSuccessor(pred, vector), in its order and with its duplicate rule.
#pragma once
#include "hyas_dev.h"
#include <vector>

namespace hyas
{
namespace ad
{
struct Range // std::pair<float, Utility::AngleRange>
{
    float dist, start, range;
};
HD float range_end(float start, float range) { return start + range; } // GetAngleRangeEnd, src/utility.cpp:1789-1792
HD float rad2deg(float rad) { return (float)((double)(rad / 2) / HY_PI * 360); } // :1186-1191

// GetAngleDistance(a1, a2), src/utility.cpp:1836-1849
HD float angle_distance(float angle1, float angle2)
{
    float out = angle2 - angle1;
    if ((double)out >= HY_PI || (double)out <= -HY_PI)
        out = rad_norm(out);
    return out;
}
// GetAngleDistance(angle, range), :1831-1834
HD float angle_distance_to_range(float angle, float start, float range)
{
    return fminf_(hy_fabsf(angle_distance(angle, start)), hy_fabsf(angle_distance(angle, range_end(start, range))));
}
// IsEqual(a1, a2), :1812-1823
HD bool angle_equal(float a1, float a2) { return hy_fabsf(angle_distance(a1, a2)) < deg2rad(0.01f); }
// IsAngleRangeInclude(range, angle), :1298-1330
HD bool range_includes(float start, float range, float angle)
{
    const float end = range_end(start, range);
    if (angle_equal(angle, start) || angle_equal(angle, end))
        return true;
    const float a2 = rad_0_2pi(angle);
    if (a2 >= rad_0_2pi(start) && a2 <= rad_0_2pi(end))
        return true;
    const float an = rad_norm(angle);
    return an >= rad_norm(start) && an <= rad_norm(end);
}
// IsAngleRangeInclude(range 1, range 2), :798-827
HD bool range_includes_range(float s1, float r1, float s2, float r2)
{
    if (r2 == 0)
        return range_includes(s1, r1, s2);
    if (angle_equal(s1, range_end(s2, r2)) && angle_equal(range_end(s1, r1), s2))
        return false;
    return range_includes(s1, r1, s2) && range_includes(s1, r1, range_end(s2, r2)) && r1 > r2;
}
// IsOverlap, :771-796
HD bool ranges_overlap(float s1, float r1, float s2, float r2)
{
    const float e1 = range_end(s1, r1), e2 = range_end(s2, r2);
    if (range_includes_range(s1, r1, s2, r2) || range_includes_range(s2, r2, s1, r1))
        return false;
    const bool a = range_includes(s1, r1, s2), b = range_includes(s1, r1, e2), c = range_includes(s2, r2, s1), d = range_includes(s2, r2, e1);
    return (a && !b) || (!a && b) || (c && !d) || (!c && d);
}
// FindAngleRange(a1, a2)[0], :1465-1492
HD void range_between(float a1, float a2, float &start, float &range)
{
    const float t1 = rad_0_2pi(a1), t2 = rad_0_2pi(a2);
    start = t1;
    if (angle_equal(t1, t2))
    {
        range = 0;
        return;
    }
    range = t2 - t1;
    if ((double)range > HY_PI)
        range = (float)HY_PI;
}
// FindCommonAngleRange, :1737-1782; {-1, -1} when the ranges have nothing in common
HD void common_range(float s1, float r1, float s2, float r2, float &start, float &range)
{
    if (range_includes_range(s1, r1, s2, r2))
    {
        start = s2, range = r2;
        return;
    }
    if (range_includes_range(s2, r2, s1, r1))
    {
        start = s1, range = r1;
        return;
    }
    if (angle_equal(s1, s2) && angle_equal(range_end(s1, r1), range_end(s2, r2))) // IsEqual(ar1, ar2), :1824-1829
    {
        start = s1, range = r1;
        return;
    }
    if (ranges_overlap(s1, r1, s2, r2))
    {
        const float e1 = range_end(s1, r1), e2 = range_end(s2, r2);
        if (range_includes(s1, r1, s2) && range_includes(s2, r2, e1))
        {
            range_between(s2, e1, start, range);
            return;
        }
        if (range_includes(s1, r1, e2) && range_includes(s2, r2, s1))
        {
            range_between(s1, e2, start, range);
            return;
        }
    }
    start = -1, range = -1;
}

// Utility::GetDistanceFromPolygonToPointAtAngle (src/utility.cpp:1072-1109) for one obstacle rectangle: distance from p
// to the rectangle along the ray of 10000 map units in direction `angle`; 0 on the outline, -1 when the ray misses.
// The two intersection tests take their arguments in the reference's two orders (edge first, then ray first:
// ComputationalGeometry::Segment::IsIntersect, src/computational_geometry.cpp:20-45, is Utility::IsIntersect on the ray).
HD float ray_distance(const Quad &q, F2 p, float angle)
{
    float sn, cs;
    hy_sincosf(angle, &sn, &cs);
    const F2 end = add(p, f2(10000.f * cs, 10000.f * sn));
    for (int i = 0; i < 4; ++i)
        if (on_segment(qv(q, i), qv(q, i + 1), p))
            return 0;
    bool any = false;
    for (int i = 0; i < 4; ++i)
        any = any || seg_intersect(qv(q, i), qv(q, i + 1), p, end);
    float out = -1;
    if (!any)
        return out;
    for (int i = 0; i < 4; ++i)
        if (seg_intersect(p, end, qv(q, i), qv(q, i + 1)))
        {
            const float d = normf(sub(intersection_point(p, end, qv(q, i), qv(q, i + 1)), p));
            if (out > d || out < 0)
                out = d;
        }
    return out;
}

// ---- per node ----------------------------------------------------------------------------------------------------
// the node's cell: its ranges and density (GetNormalizedObstacleDensity, src/collisiondetection.cpp:1153-1177)
struct Cell
{
    const Range *rng;
    uint32_t n;
    float density;
};
HD Cell cell_of(const DevMap &m, float x, float y)
{
    Cell c;
    c.rng = nullptr, c.n = 0, c.density = 0;
    const uint32_t key = grid_key(m, x, y);
    if (key < m.w * m.h)
    {
        const uint32_t o = m.ad_off[key];
        c.rng = reinterpret_cast<const Range *>(m.ad_rng) + o;
        c.n = m.ad_off[key + 1] - o;
        c.density = m.ad_density[key];
    }
    return c;
}
// LimitSteeringAngle, src/collisiondetection.cpp:1364-1381: clamp, else round to a tenth of a degree
HD float limit_steering(float steering, float limit)
{
    if (steering > limit)
        return limit;
    if (steering < -limit)
        return -limit;
    return deg2rad(hy_roundf(rad2deg(steering) * 10) / 10);
}
// FindStepSize, src/collisiondetection.cpp:1449-1484
HD float find_step_size(const DevMap &m, const Cell &c, float t, float steering, float dist_to_goal, float fixed_step)
{
    float min_dist = m.ad_range;
    const float final_orientation = rad_norm(t + steering);
    for (uint32_t i = 0; i < c.n; ++i)
        if (range_includes(c.rng[i].start, c.rng[i].range, final_orientation))
        {
            min_dist = c.rng[i].dist;
            break;
        }
    const float weight = (float)(-0.8 * (double)c.density + 0.9); // GetStepSizeWeight, :1206-1221
    float step = weight * min_dist;
    if (step < fixed_step)
        step = fixed_step;
    if (step > dist_to_goal)
        step = dist_to_goal;
    return step;
}
// IsCloseToFreeAngleRange, src/collisiondetection.cpp:1293-1362.  Without a free range in the node's list the
// reference fills two shadowing locals and compares the untouched outer pair (10000 > 10000): angle1.
HD float closest_to_free_range(const DevMap &m, const Cell &c, float angle1, float angle2)
{
    float d1 = 10000, d2 = 10000;
    for (uint32_t i = 0; i < c.n; ++i)
        if (c.rng[i].dist == m.ad_range)
        {
            d1 = fminf_(angle_distance_to_range(angle1, c.rng[i].start, c.rng[i].range), d1);
            d2 = fminf_(angle_distance_to_range(angle2, c.rng[i].start, c.rng[i].range), d2);
        }
    return d1 > d2 ? angle2 : angle1;
}

// Most pairs of one node, the reverse pair not counted: with it a node fills its HYAS_ADAPTIVE_SLOTS.  (Two pairs differ
// by a degree of steering at least and most steering angles are kept within +-30 degrees; the in-tree maps give at most
// 20 pairs with the 5-degree fans of fixed_number_of_steering_angle_in_free_angle_range = 0, 7 with the defaults.)
#define HYAS_AD_MAX_PAIRS (HYAS_ADAPTIVE_SLOTS - 1)
struct Pairs
{
    float *step, *steer; // HYAS_AD_MAX_PAIRS entries each, the caller's (workgroup memory on the device)
    int n;
    bool overflow; // more than HYAS_AD_MAX_PAIRS distinct pairs: the list is cut, the caller must not use it
};
// emplace unless Utility::DuplicateCheck(out, {step, steer}, false) (src/utility.cpp:1928-1948: same steering angle
// to within one degree) finds it
HD void push_pair(Pairs &out, float step, float steer)
{
    const float one_deg = deg2rad(1);
    for (int i = 0; i < out.n; ++i)
        if (hy_fabsf(out.steer[i] - steer) < one_deg)
            return;
    if (out.n >= HYAS_AD_MAX_PAIRS)
    {
        out.overflow = true;
        return;
    }
    out.step[out.n] = step, out.steer[out.n] = steer, ++out.n;
}
// SelectStepSizeAndSteeringAngle on the node's ranges (src/collisiondetection.cpp:728-938; caller :1383-1401)
HD void select_pairs(const DevMap &m, const Cell &c, float x, float y, float t, float gx, float gy, float gt, float dist_start_goal,
                     float fixed_step, Pairs &out)
{
    const float dist_to_goal = normf(sub(f2(x, y), f2(gx, gy)));
    const float lim = deg2rad(30);
    float toward_goal = 0;
    if (m.ad_one_more)
    {
        const float weighting = dist_to_goal / dist_start_goal;
        const float angle_to_goal = rad_0_2pi(hy_atan2f(gy - y, gx - x)); // Utility::GetAngle, src/utility.cpp:1269-1296
        const float orientation_diff = -rad_norm(t - gt);
        toward_goal = -rad_norm(weighting * angle_to_goal + (1 - weighting) * orientation_diff);
    }
    // AddOneMoreStepSizeAndSteeringAngle, :1263-1291 (the step of FindStepSize never exceeds the goal distance)
    const auto one_more = [&]() {
        const float step = find_step_size(m, c, t, toward_goal, dist_to_goal, fixed_step);
        push_pair(out, dist_to_goal < step ? dist_to_goal : step, limit_steering(toward_goal, lim));
    };
    const auto steer_to = [&](float steering) { push_pair(out, find_step_size(m, c, t, steering, dist_to_goal, fixed_step), steering); };
    for (uint32_t i = 0; i < c.n; ++i)
    {
        const float dist = c.rng[i].dist, start = c.rng[i].start, range = c.rng[i].range;
        if (range == 0) // :751-762: the step is the range's distance
        {
            push_pair(out, dist, limit_steering(rad_norm(start - t), lim));
            continue;
        }
        if (dist == m.ad_range) // free range, :764-867
        {
            if (m.ad_one_more)
                one_more();
            const float middle = rad_norm((float)((double)start + 0.5 * (double)range - (double)t));
            if (m.ad_fixed_number)
                steer_to(limit_steering(middle, lim));
            else if (range < deg2rad(3))
                steer_to(limit_steering(middle, lim));
            else
            {
                const float step5 = deg2rad(5), end = range_end(start, range);
                for (float a = start; a <= end; a += step5)
                    steer_to(limit_steering(rad_norm(a - t), lim));
            }
        }
        else // obstacle range, :869-930
        {
            if (m.ad_one_more && !m.ad_one_more_free_only)
                one_more();
            if (m.ad_towards_free)
                steer_to(limit_steering(rad_norm(closest_to_free_range(m, c, start, range_end(start, range)) - t), lim));
            else
            {
                steer_to(limit_steering(rad_norm(start - t), lim));
                steer_to(rad_norm(start + range - t));
            }
        }
    }
}
// ---- the same selection, laid out for a group of lanes ---------------------------------------------------------------
// select_pairs() walks the node's ranges once per candidate and normalises every range's ends on every visit: one long
// dependent chain.  The pieces below let the lanes of a node work side by side: range_pre() once per range; emit_range()
// forms the candidates of ONE range (they depend on the node and the ranges only, not on what has been emitted so far);
// merge_emitted() then applies the reference's duplicate rule to them in range order, which is all that has to be
// serial.  Same functions on the same values as select_pairs(): the lists are identical, bit for bit.  Not for
// fixed_number_of_steering_angle_in_free_angle_range = 0 (a free range then emits a fan of any length).
struct RangePre // a range's ends in both normal forms
{
    float s0, e0, sn, en;
};
HD RangePre range_pre(const Range &r)
{
    const float end = range_end(r.start, r.range);
    RangePre p;
    p.s0 = rad_0_2pi(r.start), p.e0 = rad_0_2pi(end), p.sn = rad_norm(r.start), p.en = rad_norm(end);
    return p;
}
// range_includes(r.start, r.range, angle) with a2 = rad_0_2pi(angle), an = rad_norm(angle) formed by the caller
HD bool range_includes_pre(const Range &r, const RangePre &p, float angle, float a2, float an)
{
    if (angle_equal(angle, r.start) || angle_equal(angle, range_end(r.start, r.range)))
        return true;
    if (a2 >= p.s0 && a2 <= p.e0)
        return true;
    return an >= p.sn && an <= p.en;
}
HD float find_step_size_pre(const DevMap &m, const Cell &c, const RangePre *pre, float t, float steering, float dist_to_goal, float fixed_step)
{
    float min_dist = m.ad_range;
    const float final_orientation = rad_norm(t + steering);
    const float a2 = rad_0_2pi(final_orientation), an = rad_norm(final_orientation);
    for (uint32_t i = 0; i < c.n; ++i)
        if (range_includes_pre(c.rng[i], pre[i], final_orientation, a2, an))
        {
            min_dist = c.rng[i].dist;
            break;
        }
    const float weight = (float)(-0.8 * (double)c.density + 0.9);
    float step = weight * min_dist;
    if (step < fixed_step)
        step = fixed_step;
    if (step > dist_to_goal)
        step = dist_to_goal;
    return step;
}
#define HYAS_AD_EMIT_MAX 3 // candidates of one range at most: the extra successor towards the goal and two steering angles
// the steering angle of the extra successor (SelectStepSizeAndSteeringAngle, :736-749)
HD float toward_goal_steering(const DevMap &m, float x, float y, float t, float gx, float gy, float gt, float dist_start_goal)
{
    if (!m.ad_one_more)
        return 0;
    const float dist_to_goal = normf(sub(f2(x, y), f2(gx, gy)));
    const float weighting = dist_to_goal / dist_start_goal;
    const float angle_to_goal = rad_0_2pi(hy_atan2f(gy - y, gx - x));
    const float orientation_diff = -rad_norm(t - gt);
    return -rad_norm(weighting * angle_to_goal + (1 - weighting) * orientation_diff);
}
// candidates {step, steering angle} of range i in the order select_pairs() pushes them; returns how many
HD int emit_range(const DevMap &m, const Cell &c, const RangePre *pre, uint32_t i, float t, float dist_to_goal, float toward_goal, float fixed_step,
                  float *step, float *steer)
{
    const float lim = deg2rad(30);
    int n = 0;
    const auto one_more = [&]() {
        const float st = find_step_size_pre(m, c, pre, t, toward_goal, dist_to_goal, fixed_step);
        step[n] = dist_to_goal < st ? dist_to_goal : st, steer[n] = limit_steering(toward_goal, lim), ++n;
    };
    const auto steer_to = [&](float steering) {
        step[n] = find_step_size_pre(m, c, pre, t, steering, dist_to_goal, fixed_step), steer[n] = steering, ++n;
    };
    const float dist = c.rng[i].dist, start = c.rng[i].start, range = c.rng[i].range;
    if (range == 0)
    {
        step[n] = dist, steer[n] = limit_steering(rad_norm(start - t), lim), ++n;
        return n;
    }
    if (dist == m.ad_range)
    {
        if (m.ad_one_more)
            one_more();
        steer_to(limit_steering(rad_norm((float)((double)start + 0.5 * (double)range - (double)t)), lim));
    }
    else
    {
        if (m.ad_one_more && !m.ad_one_more_free_only)
            one_more();
        if (m.ad_towards_free)
            steer_to(limit_steering(rad_norm(closest_to_free_range(m, c, start, range_end(start, range)) - t), lim));
        else
        {
            steer_to(limit_steering(rad_norm(start - t), lim));
            steer_to(rad_norm(start + range - t));
        }
    }
    return n;
}
// can the staged form be used for this configuration?  (a narrow free range takes the middle either way; a wide one fans out)
HD bool staged_selection_applies(const DevMap &m) { return m.ad_fixed_number != 0; }
// counts[i] candidates of range i at step/steer[i * HYAS_AD_EMIT_MAX ...]: the reference's list
HD void merge_emitted(const Cell &c, const int *counts, const float *step, const float *steer, Pairs &out)
{
    out.n = 0, out.overflow = false;
    for (uint32_t i = 0; i < c.n; ++i)
        for (int j = 0; j < counts[i]; ++j)
            push_pair(out, step[i * HYAS_AD_EMIT_MAX + j], steer[i * HYAS_AD_EMIT_MAX + j]);
}

// The pair list of CreateSuccessor(pred) without the reverse pair (src/hybrid_a_star.cpp:547-621).
// m.ad_mode: 1 adaptive_steering_angle_and_step_size, 2 adaptive_step_size.
// c: the cell of (x, y) as cell_of gives it, or a copy of it with the ranges somewhere faster
HD void step_steer_pairs(const DevMap &m, const Cell &c, float x, float y, float t, float gx, float gy, float gt, float dist_start_goal, Pairs &out)
{
    out.n = 0, out.overflow = false;
    const float dist_to_goal = normf(sub(f2(x, y), f2(gx, gy)));
    if (m.ad_mode == 1)
    {
        if (c.density > m.ad_constant_density)
        {
            const float first = (m.res_f > dist_to_goal) ? dist_to_goal : m.res_f;
            for (int k = 0; k < m.n_succ; ++k)
                if (!m.succ_reverse[k])
                    out.step[out.n] = first, out.steer[out.n] = m.succ_steer[k], ++out.n;
        }
        else
            select_pairs(m, c, x, y, t, gx, gy, gt, dist_start_goal, m.res_f, out);
    }
    else
    {
        for (int k = 0; k < m.n_succ; ++k)
            if (!m.succ_reverse[k])
                out.step[out.n] = find_step_size(m, c, t, m.succ_steer[k], dist_to_goal, m.res_f), out.steer[out.n] = m.succ_steer[k], ++out.n;
    }
}

HD void step_steer_pairs(const DevMap &m, float x, float y, float t, float gx, float gy, float gt, float dist_start_goal, Pairs &out)
{
    step_steer_pairs(m, cell_of(m, x, y), x, y, t, gx, gy, gt, dist_start_goal, out);
}

// ---- per costmap (host) ------------------------------------------------------------------------------------------
// angles of CollisionDetection::SweepDistanceAndAngle without a steering range (src/collisiondetection.cpp:949-956):
// from 0 for as long as the float accumulator stays <= ConvertDegToRad(360), one degree at a time
inline std::vector<float> sweep_angles()
{
    std::vector<float> a;
    const float first = 0, last = first + deg2rad(360), one = deg2rad(1);
    for (float angle = first; angle <= last; angle = angle + one)
        a.push_back(angle);
    return a;
}
// The run-length pass of SetDistanceAngleRangeMap (src/collisiondetection.cpp:415-489) over one cell's sweep:
// free runs (distance == detection range) and obstacle runs alternate; a run that ends becomes {label, start, width},
// labelled with the detection range when it was free and with the cell's minimum obstacle distance otherwise; the run
// still open at the last angle is labelled with the minimum distance whatever its kind.
// angle(i), dist(i): the sweep; emit(label, start, width) receives the runs in order.  Host and device.
template <class AngleAt, class DistAt, class Emit>
HD void ranges_of_sweep_each(AngleAt angle, DistAt dist, uint32_t n, float min_distance, float detection_range, Emit emit)
{
    float range_start = -1, range_range = -1;
    bool free_flag = false, obstacle_flag = false;
    const float last = n ? angle(n - 1) : 0.f;
    for (uint32_t i = 0; i < n; ++i)
    {
        const float a = angle(i);
        if (dist(i) == detection_range)
        {
            if (obstacle_flag)
            {
                obstacle_flag = false, free_flag = true;
                range_range = a - range_start;
                if (range_start != -1 && range_range != -1)
                    emit(min_distance, range_start, range_range);
                range_start = a;
            }
            if (!free_flag)
                free_flag = true, obstacle_flag = false, range_start = a;
        }
        else
        {
            if (free_flag)
            {
                free_flag = false, obstacle_flag = true;
                range_range = a - range_start;
                if (range_start != -1 && range_range != -1)
                    emit(detection_range, range_start, range_range);
                range_start = a;
            }
            if (!obstacle_flag)
                free_flag = false, obstacle_flag = true, range_start = a;
        }
        if (a == last)
        {
            range_range = a - range_start;
            if (range_start != -1 && range_range != -1 && range_range != 0)
                emit(min_distance, range_start, range_range);
        }
    }
}
// FindFreeAngleRangeAndObstacleAngleRange (src/collisiondetection.cpp:688-725): a range cut to what it has in
// common with {0, ConvertDegToRad(359.9999)}; false: nothing in common, the range is dropped
HD bool cut_range_to_circle(Range &r)
{
    float s, w;
    common_range(r.start, r.range, 0.f, deg2rad(359.9999f), s, w);
    if (s == -1 || w == -1)
        return false;
    r.start = s, r.range = w;
    return true;
}
#if !defined(__HIP_DEVICE_COMPILE__)
inline void ranges_of_sweep(const float *angle, const float *dist, size_t n, float min_distance, float detection_range, std::vector<Range> &out)
{
    out.clear();
    ranges_of_sweep_each([&](uint32_t i) { return angle[i]; }, [&](uint32_t i) { return dist[i]; }, (uint32_t)n, min_distance, detection_range,
                         [&](float label, float start, float width) { out.push_back(Range{label, start, width}); });
}
inline void cut_to_circle(std::vector<Range> &v)
{
    size_t k = 0;
    for (size_t i = 0; i < v.size(); ++i)
    {
        Range r = v[i];
        if (cut_range_to_circle(r))
            v[k++] = r;
    }
    v.resize(k);
}
#endif
} // namespace ad
} // namespace hyas
