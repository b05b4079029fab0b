/*
 * oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Serial CPU restatement of the Hybrid A* expansion path of
 * JialiangHan/path_planner, written to be read side by side with the
 * reference.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file; the product library
 * (path_planner_amd/csrc) never does.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference root).  Arithmetic types (float vs double, operand order) are
 * kept exactly as the C++ usual arithmetic conversions give them in the
 * reference; elementary functions go through include/hyas_math.h (or glibc
 * when built with -DORC_USE_LIBM) -- see the note in that header.
 *
 * Parity pins: Utility:: known answers of test/test.cpp (tests/golden/
 * utility_kat.json); Reeds-Shepp known answers and the 60x30 table checksum
 * recorded from the in-tree src/ReedsShepp.cpp; the Dubins words against
 * oracle/_ref (src/dubins.cpp compiled as is).  Not pinned by any reference
 * fixture ("parity unpinned"): collision verdicts, successor sets, g/h,
 * expansion order (Boost binomial-heap tie order is replaced by
 * "lowest f, then earliest push"), costmap_2d::worldToMap (restated from its
 * documented semantics), OMPL's Reeds-Shepp distance (the in-tree
 * ReedsShepp.cpp stands in for it) and OMPL's Dubins distance (restated from
 * OMPL's published source, double arithmetic as there).
 */
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <queue>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/hyas.h"
#include "../include/hyas_math.h"

namespace orc
{

// ---------------------------------------------------------------- math switch
#ifdef ORC_USE_LIBM
static inline float m_sinf(float x) { return sinf(x); }
static inline float m_cosf(float x) { return cosf(x); }
static inline float m_acosf(float x) { return acosf(x); }
static inline float m_atan2f(float y, float x) { return atan2f(y, x); }
static inline double m_sin(double x) { return sin(x); }
static inline double m_cos(double x) { return cos(x); }
static inline double m_atan2(double y, double x) { return atan2(y, x); }
static inline double m_asin(double x) { return asin(x); }
static inline double m_acos(double x) { return acos(x); }
static inline double m_fmod(double x, double y) { return fmod(x, y); }
#else
static inline float m_sinf(float x) { return hy_sinf(x); }
static inline float m_cosf(float x) { return hy_cosf(x); }
static inline float m_acosf(float x) { return hy_acosf(x); }
static inline float m_atan2f(float y, float x) { return hy_atan2f(y, x); }
static inline double m_sin(double x) { return hy_sin(x); }
static inline double m_cos(double x) { return hy_cos(x); }
static inline double m_atan2(double y, double x) { return hy_atan2(y, x); }
static inline double m_asin(double x) { return hy_asin(x); }
static inline double m_acos(double x) { return hy_acos(x); }
static inline double m_fmod(double x, double y) { return hy_fmod(x, y); }
#endif
static inline float m_sqrtf(float x) { return hy_sqrtf(x); }
static inline float m_fabsf(float x) { return hy_fabsf(x); }
static inline float m_floorf(float x) { return hy_floorf(x); }
static inline float m_roundf(float x) { return hy_roundf(x); }

static const double kPi = 3.14159265358979323846; // M_PI

// ------------------------------------------------------------ Eigen stand-ins
// Eigen::Vector2f semantics used by the reference: component-wise +,-, scalar
// *, dot = x1*x2 + y1*y2, norm = sqrt(x*x + y*y), normalized() divides by the
// norm when squaredNorm > 0.
struct V2
{
    float x, y;
};
static inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
static inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
static inline bool operator==(V2 a, V2 b) { return a.x == b.x && a.y == b.y; }
static inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline float norm(V2 a) { return m_sqrtf(a.x * a.x + a.y * a.y); }
typedef std::vector<V2> Polygon; // 5 vertices, first repeated

// ---------------------------------------------------------------- Utility::
// src/utility.cpp:1180-1185
static float ConvertDegToRad(float deg)
{
    float rad = (float)((double)(deg / 360 * 2) * kPi);
    return rad;
}

// src/utility.cpp:1249-1268
static float RadToZeroTo2P(float rad)
{
    const double two_pi = (double)2.f * kPi;
    if ((double)rad >= 0 && (double)rad < two_pi)
        return rad;
    float out = (float)m_fmod((double)rad, two_pi);
    if (out < 0)
        out = (float)((double)out + two_pi);
    if (hy_fabs((double)out - 2 * kPi) < 1e-3)
        out = 0;
    return out;
}

// src/utility.cpp:1210-1231
static float RadNormalization(float rad)
{
    if ((double)rad >= -kPi && (double)rad <= kPi)
        return rad;
    const double two_pi = (double)2.f * kPi;
    float out = (float)m_fmod((double)rad, two_pi);
    if (out < 0)
        out = (float)((double)out + two_pi);
    if ((double)out > kPi)
        out = (float)((double)out - (double)2.0f * kPi);
    if ((double)m_fabsf(out) < 1e-6)
        out = 0;
    return out;
}

// src/utility.cpp:209-249.  Argument order: (p1, p3, p2), p2 is the query.
static bool OnSegment(V2 p1, V2 p3, V2 p2)
{
    if ((double)m_fabsf(p2.x - p1.x) < 1e-3 && (double)m_fabsf(p2.x - p3.x) < 1e-3 &&
        p2.y <= std::max(p1.y, p3.y) && p2.y >= std::min(p1.y, p3.y))
        return true;
    if (p2.x <= std::max(p1.x, p3.x) && p2.x >= std::min(p1.x, p3.x) &&
        (double)m_fabsf(p2.y - p1.y) < 1e-3 && (double)m_fabsf(p2.y - p3.y) < 1e-3)
        return true;
    if (p2.x <= std::max(p1.x, p3.x) && p2.x >= std::min(p1.x, p3.x) &&
        p2.y <= std::max(p1.y, p3.y) && p2.y >= std::min(p1.y, p3.y))
    {
        V2 p1p2 = p2 - p1;
        V2 p1p3 = p3 - p1;
        return (double)m_fabsf(dot(p1p2, p1p3) / norm(p1p2) / norm(p1p3) - 1) < 1e-6;
    }
    return false;
}

// src/utility.cpp:251-271
static bool IsPointOnLine(V2 point, V2 start, V2 unit_vector)
{
    V2 diff = point - start;
    if ((double)m_fabsf(diff.x) < 1e-3 && (double)m_fabsf(diff.y) < 1e-3)
        return true;
    float angle = m_acosf(dot(diff, unit_vector) / norm(diff) / norm(unit_vector));
    if ((double)m_fabsf(angle) < 1e-3 || hy_fabs((double)angle - kPi) < 1e-3)
        return true;
    return false;
}

// src/utility.cpp:273-279
static float CrossProduct(V2 p1, V2 p2) { return p1.x * p2.y - p2.x * p1.y; }

// src/utility.cpp:280-306
static int IsIntersect(V2 p1, V2 p2, V2 p3, V2 p4)
{
    if ((p1.x == p2.x && p2.x == p3.x && p3.x == p4.x) ||
        (p1.y == p2.y && p2.y == p3.y && p3.y == p4.y))
        return 0;
    if (std::max(p3.x, p4.x) < std::min(p1.x, p2.x) || std::max(p3.y, p4.y) < std::min(p1.y, p2.y) ||
        std::max(p1.x, p2.x) < std::min(p3.x, p4.x) || std::max(p1.y, p2.y) < std::min(p3.y, p4.y))
        return 0;
    if (CrossProduct(p1 - p3, p4 - p3) * CrossProduct(p2 - p3, p4 - p3) <= 0 &&
        CrossProduct(p3 - p2, p1 - p2) * CrossProduct(p4 - p2, p1 - p2) <= 0)
        return 1;
    return 0;
}

// src/utility.cpp:308-327
static V2 FindIntersectionPoint(V2 p1, V2 p2, V2 p3, V2 p4)
{
    if (0 == IsIntersect(p1, p2, p3, p4))
        return V2{10000, 10000};
    V2 dir_1 = p2 - p1, dir_2 = p4 - p3;
    float t = CrossProduct(p3 - p1, dir_2) / CrossProduct(dir_1, dir_2);
    return p1 + V2{t * dir_1.x, t * dir_1.y};
}

// src/utility.cpp:328-374
static int IsAboveSegment(V2 p1, V2 p2, V2 p3)
{
    const int above = 1, on_segment = 1, below = 0;
    if (OnSegment(p1, p2, p3))
        return on_segment;
    if ((p2 - p1).x == 0)
        return p3.x > p1.x ? below : above;
    if ((p2 - p1).y == 0)
        return p3.y > p1.y ? above : below;
    float k = (p2 - p1).y / (p2 - p1).x;
    float b = p1.y - k * p1.x;
    return (k * p3.x + b - p3.y < 0) ? above : below;
}

// src/utility.cpp:375-422
static int IsInsidePolygon(const Polygon &polygon, V2 point)
{
    V2 far_away_point{10000, point.y};
    int number_intersection = 0;
    for (size_t i = 0; i + 1 < polygon.size(); ++i)
    {
        if (IsIntersect(point, far_away_point, polygon[i], polygon[i + 1]) == 1)
        {
            V2 ip = FindIntersectionPoint(point, far_away_point, polygon[i], polygon[i + 1]);
            if ((double)norm(polygon[i] - ip) < 1e-3)
            {
                if (IsAboveSegment(point, far_away_point, polygon[i + 1]) == 0)
                    number_intersection += 1;
            }
            else if ((double)norm(polygon[i + 1] - ip) < 1e-3)
            {
                if (IsAboveSegment(point, far_away_point, polygon[i]) == 0)
                    number_intersection += 1;
            }
            else
                number_intersection += 1;
        }
    }
    return (number_intersection % 2) == 0 ? 0 : 1;
}

// src/utility.cpp:430-443
static bool IsOnPolygon(const Polygon &polygon, V2 point)
{
    for (size_t i = 0; i + 1 < polygon.size(); ++i)
        if (OnSegment(polygon[i], polygon[i + 1], point))
            return true;
    return false;
}

// src/utility.cpp:457-474
static Polygon CreatePolygon(V2 origin, float width = 1, float height = 1)
{
    Polygon p;
    p.push_back(origin);
    p.push_back(V2{width + origin.x, origin.y});
    p.push_back(V2{width + origin.x, origin.y + height});
    p.push_back(V2{origin.x, origin.y + height});
    p.push_back(origin);
    return p;
}

// src/utility.cpp:476-510.  Matrix2f * Vector2f = (m00*v0 + m01*v1, m10*v0 + m11*v1).
static Polygon CreatePolygon(V2 center, float width, float height, float heading)
{
    V2 pt[4] = {{-width / 2, -height / 2}, {+width / 2, -height / 2}, {+width / 2, +height / 2}, {-width / 2, +height / 2}};
    float m00 = m_cosf(heading), m01 = -m_sinf(heading), m10 = m_sinf(heading), m11 = m_cosf(heading);
    Polygon p;
    for (int i = 0; i < 4; ++i)
        p.push_back(center + V2{m00 * pt[i].x + m01 * pt[i].y, m10 * pt[i].x + m11 * pt[i].y});
    p.push_back(p[0]);
    return p;
}

// src/utility.cpp:521-526
static float GetDistanceFromPointToPoint(V2 p1, V2 p2) { return norm(p1 - p2); }

// src/utility.cpp:528-540
static int IsSegmentIntersectWithPolygon(const Polygon &polygon, V2 start, V2 end)
{
    for (size_t i = 0; i + 1 < polygon.size(); ++i)
        if (IsIntersect(polygon[i], polygon[i + 1], start, end))
            return 1;
    return 0;
}

// src/utility.cpp:1591-1596 (Eigen normalized())
static V2 GetUnitVector(V2 start, V2 end)
{
    V2 diff = end - start;
    float z = diff.x * diff.x + diff.y * diff.y;
    if (z > 0)
    {
        float n = m_sqrtf(z);
        return V2{diff.x / n, diff.y / n};
    }
    return diff;
}

// src/utility.cpp:578-592
static float GetDistanceFromPointToLine(V2 point, V2 start, V2 unit_vector)
{
    if (IsPointOnLine(point, start, unit_vector))
        return 0;
    float projection_length = m_fabsf(dot(point - start, unit_vector));
    return m_sqrtf(norm(point - start) * norm(point - start) - projection_length * projection_length);
}

// src/utility.cpp:541-576
static float GetDistanceFromSegmentToPoint(V2 start, V2 end, V2 point)
{
    if (OnSegment(start, end, point))
        return 0;
    V2 far_point, unit_vector = GetUnitVector(start, end);
    float distance_to_start = GetDistanceFromPointToPoint(start, point);
    float distance_to_goal = GetDistanceFromPointToPoint(end, point);
    far_point = distance_to_start > distance_to_goal ? start : end;
    float projection_length = m_fabsf(dot(point - far_point, end - start)) / norm(end - start);
    if (projection_length > norm(end - start))
        return std::min(distance_to_start, distance_to_goal);
    return GetDistanceFromPointToLine(point, start, unit_vector);
}

// src/utility.cpp:594-619
static float GetDistanceFromPolygonToPoint(const Polygon &polygon, V2 point)
{
    float min_distance = 100000;
    if (IsOnPolygon(polygon, point))
        return 0;
    if (IsInsidePolygon(polygon, point))
        return -1;
    for (size_t i = 0; i + 1 < polygon.size(); ++i)
    {
        float d = GetDistanceFromSegmentToPoint(polygon[i], polygon[i + 1], point);
        if (d < min_distance)
            min_distance = d;
    }
    return min_distance;
}

// src/utility.cpp:956-966
static bool IsPolygonIntersectWithPolygon(const Polygon &polygon1, const Polygon &polygon2)
{
    for (size_t i = 0; i + 1 < polygon1.size(); ++i)
        if (IsSegmentIntersectWithPolygon(polygon2, polygon1[i], polygon1[i + 1]))
            return true;
    return false;
}

// src/utility.cpp:967-997
static float GetDistanceFromPolygonToSegment(const Polygon &polygon, V2 start, V2 end)
{
    float out = 100000;
    if (IsSegmentIntersectWithPolygon(polygon, start, end))
        return -1;
    float d = GetDistanceFromPolygonToPoint(polygon, start);
    if (out > d)
        out = d;
    d = GetDistanceFromPolygonToPoint(polygon, end);
    if (out > d)
        out = d;
    for (const V2 &vertex : polygon)
    {
        d = GetDistanceFromSegmentToPoint(start, end, vertex);
        if (out > d)
            out = d;
    }
    return out;
}

// src/utility.cpp:999-1012
static bool IsPolygonInNeighbor(const Polygon &a, const Polygon &b)
{
    for (const V2 &v1 : a)
        for (const V2 &v2 : b)
            if (v1 == v2)
                return true;
    return false;
}

// src/utility.cpp:1014-1058
static Polygon CombinePolygon(const Polygon &p1, const Polygon &p2)
{
    Polygon out;
    if (!IsPolygonInNeighbor(p1, p2))
        return out;
    if (p2[0].y > p1[0].y)
        out = {p1[0], p1[1], p2[2], p2[3], p1[0]};
    else if (p2[0].x < p1[0].x)
        out = {p2[0], p1[1], p1[2], p2[3], p2[0]};
    else if (p2[0].y < p1[0].y)
        out = {p2[0], p2[1], p1[2], p1[3], p2[0]};
    else if (p2[0].x > p1[0].x)
        out = {p1[0], p2[1], p2[2], p1[3], p1[0]};
    return out;
}

// ------------------------------------------------------------------ angle ranges (adaptive step size / steering)
// Utility::AngleRange = std::pair<float, float>{start, range}, both in rad (include/utility.h:28)
typedef std::pair<float, float> AngleRange;
// src/utility.cpp:1186-1191
static float ConvertRadToDeg(float rad) { return (float)((double)(rad / 2) / kPi * 360); }
// src/utility.cpp:1789-1792
static float GetAngleRangeEnd(const AngleRange &ar) { return ar.first + ar.second; }
// src/utility.cpp:1836-1849
static float GetAngleDistance(float angle1, float angle2)
{
    float out = angle2 - angle1;
    if ((double)out >= kPi || (double)out <= -kPi)
        out = RadNormalization(out);
    return out;
}
// src/utility.cpp:1831-1834
static float GetAngleDistance(float angle, const AngleRange &ar)
{
    return std::min(m_fabsf(GetAngleDistance(angle, ar.first)), m_fabsf(GetAngleDistance(angle, GetAngleRangeEnd(ar))));
}
// src/utility.cpp:1812-1823
static bool IsEqual(float a1, float a2) { return m_fabsf(GetAngleDistance(a1, a2)) < ConvertDegToRad(0.01f); }
// src/utility.cpp:1824-1829
static bool IsEqual(const AngleRange &ar1, const AngleRange &ar2)
{
    return IsEqual(ar1.first, ar2.first) && IsEqual(GetAngleRangeEnd(ar1), GetAngleRangeEnd(ar2));
}
// src/utility.cpp:1298-1330
static bool IsAngleRangeInclude(const AngleRange &angle_range, float angle)
{
    float angle_range_end = GetAngleRangeEnd(angle_range);
    if (IsEqual(angle, angle_range.first))
        return true;
    if (IsEqual(angle, angle_range_end))
        return true;
    if ((RadToZeroTo2P(angle) >= RadToZeroTo2P(angle_range.first)) && (RadToZeroTo2P(angle) <= RadToZeroTo2P(angle_range_end)))
        return true;
    if ((RadNormalization(angle) >= RadNormalization(angle_range.first)) && (RadNormalization(angle) <= RadNormalization(angle_range_end)))
        return true;
    return false;
}
// src/utility.cpp:798-827
static bool IsAngleRangeInclude(const AngleRange &angle_range_1, const AngleRange &angle_range_2)
{
    if (angle_range_2.second == 0)
        return IsAngleRangeInclude(angle_range_1, angle_range_2.first);
    if (IsEqual(angle_range_1.first, GetAngleRangeEnd(angle_range_2)) && IsEqual(GetAngleRangeEnd(angle_range_1), angle_range_2.first))
        return false;
    float angle_range_2_start = angle_range_2.first;
    float angle_range_2_end = GetAngleRangeEnd(angle_range_2);
    if (IsAngleRangeInclude(angle_range_1, angle_range_2_start) && IsAngleRangeInclude(angle_range_1, angle_range_2_end) &&
        (angle_range_1.second > angle_range_2.second))
        return true;
    return false;
}
// src/utility.cpp:771-796
static bool IsOverlap(const AngleRange &angle_range_1, const AngleRange &angle_range_2)
{
    float angle_range_1_start = angle_range_1.first;
    float angle_range_1_end = GetAngleRangeEnd(angle_range_1);
    float angle_range_2_start = angle_range_2.first;
    float angle_range_2_end = GetAngleRangeEnd(angle_range_2);
    if (IsAngleRangeInclude(angle_range_1, angle_range_2) || IsAngleRangeInclude(angle_range_2, angle_range_1))
        return false;
    else if ((IsAngleRangeInclude(angle_range_1, angle_range_2_start) && !IsAngleRangeInclude(angle_range_1, angle_range_2_end)) ||
             (!IsAngleRangeInclude(angle_range_1, angle_range_2_start) && IsAngleRangeInclude(angle_range_1, angle_range_2_end)) ||
             (IsAngleRangeInclude(angle_range_2, angle_range_1_start) && !IsAngleRangeInclude(angle_range_2, angle_range_1_end)) ||
             (!IsAngleRangeInclude(angle_range_2, angle_range_1_start) && IsAngleRangeInclude(angle_range_2, angle_range_1_end)))
        return true;
    return false;
}
// src/utility.cpp:1465-1492
static std::vector<AngleRange> FindAngleRange(float a1, float a2)
{
    std::vector<AngleRange> out;
    float temp1 = RadToZeroTo2P(a1), temp2 = RadToZeroTo2P(a2), range;
    if (IsEqual(temp1, temp2))
    {
        out.emplace_back(AngleRange(temp1, 0));
        return out;
    }
    range = temp2 - temp1;
    out.emplace_back(AngleRange(temp1, range));
    if ((double)range > kPi)
    {
        out[0].second = (float)kPi;
        out.emplace_back(AngleRange(RadToZeroTo2P((float)((double)temp1 + kPi)), (float)((double)range - kPi)));
    }
    return out;
}
// src/utility.cpp:1737-1782
static AngleRange FindCommonAngleRange(const AngleRange &ar1, const AngleRange &ar2)
{
    AngleRange out;
    if (IsAngleRangeInclude(ar1, ar2))
        return ar2;
    if (IsAngleRangeInclude(ar2, ar1))
        return ar1;
    if (IsEqual(ar1, ar2))
        return ar1;
    if (IsOverlap(ar1, ar2))
    {
        float ar1_end = GetAngleRangeEnd(ar1);
        float ar2_end = GetAngleRangeEnd(ar2);
        if (IsAngleRangeInclude(ar1, ar2.first) && IsAngleRangeInclude(ar2, ar1_end))
            return FindAngleRange(ar2.first, ar1_end)[0];
        if (IsAngleRangeInclude(ar1, ar2_end) && IsAngleRangeInclude(ar2, ar1.first))
            return FindAngleRange(ar1.first, ar2_end)[0];
    }
    out.first = -1;
    out.second = -1;
    return out;
}
// src/utility.cpp:1928-1948
static bool DuplicateCheck(const std::vector<std::pair<float, float>> &vector, const std::pair<float, float> &element, bool flag)
{
    for (const auto &pair : vector)
        if (m_fabsf(pair.second - element.second) < ConvertDegToRad(1))
        {
            if (flag)
            {
                if ((double)m_fabsf(pair.first - element.first) < 1e-2)
                    return true;
            }
            else
                return true;
        }
    return false;
}
// ComputationalGeometry::Segment (src/computational_geometry.cpp:12-62)
struct CgSegment
{
    V2 start_, end_;
    CgSegment(V2 a, V2 b) : start_(a), end_(b) {}
    CgSegment(V2 start, float radius, float rad) : start_(start) // :12-18
    {
        V2 rotation{m_cosf(rad), m_sinf(rad)};
        end_ = V2{start_.x + radius * rotation.x, start_.y + radius * rotation.y};
    }
    bool IsIntersect(const CgSegment &o) const // :20-45
    {
        if ((start_.x == end_.x && end_.x == o.start_.x && o.start_.x == o.end_.x) ||
            (start_.y == end_.y && end_.y == o.start_.y && o.start_.y == o.end_.y))
            return false;
        if (std::max(o.start_.x, o.end_.x) < std::min(start_.x, end_.x) || std::max(o.start_.y, o.end_.y) < std::min(start_.y, end_.y) ||
            std::max(start_.x, end_.x) < std::min(o.start_.x, o.end_.x) || std::max(start_.y, end_.y) < std::min(o.start_.y, o.end_.y))
            return false;
        return CrossProduct(start_ - o.start_, o.end_ - o.start_) * CrossProduct(end_ - o.start_, o.end_ - o.start_) <= 0 &&
               CrossProduct(o.start_ - end_, start_ - end_) * CrossProduct(o.end_ - end_, start_ - end_) <= 0;
    }
    V2 FindIntersectionPoint(const CgSegment &o) const // :46-62
    {
        if (!IsIntersect(o))
            return V2{10000, 10000};
        V2 dir_1 = end_ - start_, dir_2 = o.end_ - o.start_;
        float t = CrossProduct(o.start_ - start_, dir_2) / CrossProduct(dir_1, dir_2);
        return V2{start_.x + t * dir_1.x, start_.y + t * dir_1.y};
    }
};
// src/utility.cpp:1072-1109
static float GetDistanceFromPolygonToPointAtAngle(const Polygon &polygon, V2 point, float angle)
{
    float out = -1;
    CgSegment segment(point, 10000, angle);
    if (IsOnPolygon(polygon, point))
        return 0;
    if (!IsSegmentIntersectWithPolygon(polygon, segment.start_, segment.end_))
        return out;
    for (size_t index = 0; index + 1 < polygon.size(); ++index)
    {
        CgSegment segment_on_polygon(polygon[index], polygon[index + 1]);
        if (segment.IsIntersect(segment_on_polygon))
        {
            float distance = norm(segment.FindIntersectionPoint(segment_on_polygon) - point);
            if (out > distance || out < 0)
                out = distance;
        }
    }
    return out;
}

// src/utility.cpp:2073-2089
static std::vector<float> FormSteeringAngleVec(float steering_angle, int number_of_successors)
{
    std::vector<float> out;
    out.push_back(0);
    if (number_of_successors <= 1)
        return out;
    for (int i = 1; i <= (number_of_successors - 1) / 2; i++)
    {
        out.push_back(i * ConvertDegToRad(steering_angle));
        out.push_back(-(i * ConvertDegToRad(steering_angle)));
    }
    return out;
}

// ------------------------------------------------------------------ Reeds-Shepp
// src/ReedsShepp.cpp:45-503.  Segment kinds: 0 none, 1 left, 2 straight, 3 right
// (include/ReedsShepp.h:50-56).
static const int kRsType[18][5] = {
    {1, 3, 1, 0, 0}, {3, 1, 3, 0, 0}, {1, 3, 1, 3, 0}, {3, 1, 3, 1, 0}, {1, 3, 2, 1, 0}, {3, 1, 2, 3, 0},
    {1, 2, 3, 1, 0}, {3, 2, 1, 3, 0}, {1, 3, 2, 3, 0}, {3, 1, 2, 1, 0}, {3, 2, 3, 1, 0}, {1, 2, 1, 3, 0},
    {1, 2, 3, 0, 0}, {3, 2, 1, 0, 0}, {1, 2, 1, 0, 0}, {3, 2, 3, 0, 0}, {1, 3, 2, 1, 3}, {3, 1, 2, 3, 1}};

struct RsPath
{
    int type = 0;
    double len[5] = {std::numeric_limits<double>::max(), 0, 0, 0, 0};
    double total = std::numeric_limits<double>::max();
    RsPath() {}
    RsPath(int ty, double t, double u, double v, double w = 0., double x = 0.) : type(ty)
    {
        len[0] = t, len[1] = u, len[2] = v, len[3] = w, len[4] = x;
        total = hy_fabs(t) + hy_fabs(u) + hy_fabs(v) + hy_fabs(w) + hy_fabs(x);
    }
};

namespace rs
{
static const double pi = kPi, twopi = 2. * kPi;
static const double ZERO = 10 * std::numeric_limits<double>::epsilon();
static inline double fabs_(double x) { return hy_fabs(x); }

static inline double mod2pi(double x)
{
    double v = m_fmod(x, twopi);
    if (v < -pi)
        v += twopi;
    else if (v > pi)
        v -= twopi;
    return v;
}
static inline void polar(double x, double y, double &r, double &theta)
{
    r = hy_sqrt(x * x + y * y);
    theta = m_atan2(y, x);
}
static inline void tauOmega(double u, double v, double xi, double eta, double phi, double &tau, double &omega)
{
    double delta = mod2pi(u - v), A = m_sin(u) - m_sin(delta), B = m_cos(u) - m_cos(delta) - 1.;
    double t1 = m_atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2. * (m_cos(delta) - m_cos(v) - m_cos(u)) + 3;
    tau = (t2 < 0) ? mod2pi(t1 + pi) : mod2pi(t1);
    omega = mod2pi(tau - u + v - phi);
}
static inline bool LpSpLp(double x, double y, double phi, double &t, double &u, double &v)
{
    polar(x - m_sin(phi), y - 1. + m_cos(phi), u, t);
    if (t >= -ZERO)
    {
        v = mod2pi(phi - t);
        if (v >= -ZERO)
            return true;
    }
    return false;
}
static inline bool LpSpRp(double x, double y, double phi, double &t, double &u, double &v)
{
    double t1, u1;
    polar(x + m_sin(phi), y - 1. - m_cos(phi), u1, t1);
    u1 = u1 * u1;
    if (u1 >= 4.)
    {
        u = hy_sqrt(u1 - 4.);
        double theta = m_atan2(2., u);
        t = mod2pi(t1 + theta);
        v = mod2pi(t - phi);
        return t >= -ZERO && v >= -ZERO;
    }
    return false;
}
static inline bool LpRmL(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x - m_sin(phi), eta = y - 1. + m_cos(phi), u1, theta;
    polar(xi, eta, u1, theta);
    if (u1 <= 4.)
    {
        u = -2. * m_asin(.25 * u1);
        t = mod2pi(theta + .5 * u + pi);
        v = mod2pi(phi - t + u);
        return t >= -ZERO && u <= ZERO;
    }
    return false;
}
static inline bool LpRupLumRm(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x + m_sin(phi), eta = y - 1. - m_cos(phi), rho = .25 * (2. + hy_sqrt(xi * xi + eta * eta));
    if (rho <= 1.)
    {
        u = m_acos(rho);
        tauOmega(u, -u, xi, eta, phi, t, v);
        return t >= -ZERO && v <= ZERO;
    }
    return false;
}
static inline bool LpRumLumRp(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x + m_sin(phi), eta = y - 1. - m_cos(phi), rho = (20. - xi * xi - eta * eta) / 16.;
    if (rho >= 0 && rho <= 1)
    {
        u = -m_acos(rho);
        if (u >= -.5 * pi)
        {
            tauOmega(u, u, xi, eta, phi, t, v);
            return t >= -ZERO && v >= -ZERO;
        }
    }
    return false;
}
static inline bool LpRmSmLm(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x - m_sin(phi), eta = y - 1. + m_cos(phi), rho, theta;
    polar(xi, eta, rho, theta);
    if (rho >= 2.)
    {
        double r = hy_sqrt(rho * rho - 4.);
        u = 2. - r;
        t = mod2pi(theta + m_atan2(r, -2.));
        v = mod2pi(phi - .5 * pi - t);
        return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
}
static inline bool LpRmSmRm(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x + m_sin(phi), eta = y - 1. - m_cos(phi), rho, theta;
    polar(-eta, xi, rho, theta);
    if (rho >= 2.)
    {
        t = theta;
        u = 2. - rho;
        v = mod2pi(t + .5 * pi - phi);
        return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
}
static inline bool LpRmSLmRp(double x, double y, double phi, double &t, double &u, double &v)
{
    double xi = x + m_sin(phi), eta = y - 1. - m_cos(phi), rho, theta;
    polar(xi, eta, rho, theta);
    if (rho >= 2.)
    {
        u = 4. - hy_sqrt(rho * rho - 4.);
        if (u <= ZERO)
        {
            t = mod2pi(m_atan2((4 - u) * xi - 2 * eta, -2 * xi + (u - 4) * eta));
            v = mod2pi(t - phi);
            return t >= -ZERO && v >= -ZERO;
        }
    }
    return false;
}

typedef bool (*Word)(double, double, double, double &, double &, double &);

// One row of the candidate list: base formula, the (x, y, phi) symmetry
// (timeflip negates x and phi, reflect negates y and phi), and how (t,u,v) map
// to the stored segment lengths.
static void CSC(double x, double y, double phi, RsPath &path)
{
    double t, u, v, Lmin = path.total, L;
    if (LpSpLp(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(14, t, u, v), Lmin = L;
    if (LpSpLp(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(14, -t, -u, -v), Lmin = L;
    if (LpSpLp(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(15, t, u, v), Lmin = L;
    if (LpSpLp(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(15, -t, -u, -v), Lmin = L;
    if (LpSpRp(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(12, t, u, v), Lmin = L;
    if (LpSpRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(12, -t, -u, -v), Lmin = L;
    if (LpSpRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(13, t, u, v), Lmin = L;
    if (LpSpRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(13, -t, -u, -v);
}
static void CCC(double x, double y, double phi, RsPath &path)
{
    double t, u, v, Lmin = path.total, L;
    if (LpRmL(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(0, t, u, v), Lmin = L;
    if (LpRmL(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(0, -t, -u, -v), Lmin = L;
    if (LpRmL(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(1, t, u, v), Lmin = L;
    if (LpRmL(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(1, -t, -u, -v), Lmin = L;
    double xb = x * m_cos(phi) + y * m_sin(phi), yb = x * m_sin(phi) - y * m_cos(phi);
    if (LpRmL(xb, yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(0, v, u, t), Lmin = L;
    if (LpRmL(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(0, -v, -u, -t), Lmin = L;
    if (LpRmL(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(1, v, u, t), Lmin = L;
    if (LpRmL(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(1, -v, -u, -t);
}
static void CCCC(double x, double y, double phi, RsPath &path)
{
    double t, u, v, Lmin = path.total, L;
    if (LpRupLumRm(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(2, t, u, -u, v), Lmin = L;
    if (LpRupLumRm(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(2, -t, -u, u, -v), Lmin = L;
    if (LpRupLumRm(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(3, t, u, -u, v), Lmin = L;
    if (LpRupLumRm(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(3, -t, -u, u, -v), Lmin = L;
    if (LpRumLumRp(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(2, t, u, u, v), Lmin = L;
    if (LpRumLumRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(2, -t, -u, -u, -v), Lmin = L;
    if (LpRumLumRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(3, t, u, u, v), Lmin = L;
    if (LpRumLumRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + 2. * fabs_(u) + fabs_(v)))
        path = RsPath(3, -t, -u, -u, -v);
}
static void CCSC(double x, double y, double phi, RsPath &path)
{
    double t, u, v, Lmin = path.total - .5 * pi, L;
    if (LpRmSmLm(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(4, t, -.5 * pi, u, v), Lmin = L;
    if (LpRmSmLm(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(4, -t, .5 * pi, -u, -v), Lmin = L;
    if (LpRmSmLm(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(5, t, -.5 * pi, u, v), Lmin = L;
    if (LpRmSmLm(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(5, -t, .5 * pi, -u, -v), Lmin = L;
    if (LpRmSmRm(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(8, t, -.5 * pi, u, v), Lmin = L;
    if (LpRmSmRm(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(8, -t, .5 * pi, -u, -v), Lmin = L;
    if (LpRmSmRm(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(9, t, -.5 * pi, u, v), Lmin = L;
    if (LpRmSmRm(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(9, -t, .5 * pi, -u, -v), Lmin = L;
    double xb = x * m_cos(phi) + y * m_sin(phi), yb = x * m_sin(phi) - y * m_cos(phi);
    if (LpRmSmLm(xb, yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(6, v, u, -.5 * pi, t), Lmin = L;
    if (LpRmSmLm(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(6, -v, -u, .5 * pi, -t), Lmin = L;
    if (LpRmSmLm(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(7, v, u, -.5 * pi, t), Lmin = L;
    if (LpRmSmLm(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(7, -v, -u, .5 * pi, -t), Lmin = L;
    if (LpRmSmRm(xb, yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(10, v, u, -.5 * pi, t), Lmin = L;
    if (LpRmSmRm(-xb, yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(10, -v, -u, .5 * pi, -t), Lmin = L;
    if (LpRmSmRm(xb, -yb, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(11, v, u, -.5 * pi, t), Lmin = L;
    if (LpRmSmRm(-xb, -yb, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(11, -v, -u, .5 * pi, -t);
}
static void CCSCC(double x, double y, double phi, RsPath &path)
{
    double t, u, v, Lmin = path.total - pi, L;
    if (LpRmSLmRp(x, y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(16, t, -.5 * pi, u, -.5 * pi, v), Lmin = L;
    if (LpRmSLmRp(-x, y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(16, -t, .5 * pi, -u, .5 * pi, -v), Lmin = L;
    if (LpRmSLmRp(x, -y, -phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(17, t, -.5 * pi, u, -.5 * pi, v), Lmin = L;
    if (LpRmSLmRp(-x, -y, phi, t, u, v) && Lmin > (L = fabs_(t) + fabs_(u) + fabs_(v)))
        path = RsPath(17, -t, .5 * pi, -u, .5 * pi, -v);
}
} // namespace rs

// src/ReedsShepp.cpp:494-503
static RsPath reedsSheppNormalised(double x, double y, double phi)
{
    RsPath path;
    rs::CSC(x, y, phi, path);
    rs::CCC(x, y, phi, path);
    rs::CCCC(x, y, phi, path);
    rs::CCSC(x, y, phi, path);
    rs::CCSCC(x, y, phi, path);
    return path;
}

// src/ReedsShepp.cpp:544-550
static RsPath reedsShepp(const double q0[3], const double q1[3], double rho)
{
    double dx = q1[0] - q0[0], dy = q1[1] - q0[1], dth = q1[2] - q0[2];
    double c = m_cos(q0[2]), s = m_sin(q0[2]);
    double x = c * dx + s * dy, y = -s * dx + c * dy;
    return reedsSheppNormalised(x / rho, y / rho, dth);
}

// src/ReedsShepp.cpp:576-625
static void rsInterpolate(const double q0[3], const RsPath &path, double seg, double rho, double s[3])
{
    if (seg < 0.0)
        seg = 0.0;
    if (seg > path.total)
        seg = path.total;
    double phi, v;
    s[0] = s[1] = 0.0;
    s[2] = q0[2];
    for (unsigned int i = 0; i < 5 && seg > 0; ++i)
    {
        if (path.len[i] < 0)
        {
            v = std::max(-seg, path.len[i]);
            seg += v;
        }
        else
        {
            v = std::min(seg, path.len[i]);
            seg -= v;
        }
        phi = s[2];
        switch (kRsType[path.type][i])
        {
        case 1:
            s[0] += (m_sin(phi + v) - m_sin(phi));
            s[1] += (-m_cos(phi + v) + m_cos(phi));
            s[2] = phi + v;
            break;
        case 3:
            s[0] += (-m_sin(phi - v) + m_sin(phi));
            s[1] += (m_cos(phi - v) - m_cos(phi));
            s[2] = phi - v;
            break;
        case 2:
            s[0] += (v * m_cos(phi));
            s[1] += (v * m_sin(phi));
            break;
        default:
            break;
        }
    }
    s[0] = s[0] * rho + q0[0];
    s[1] = s[1] * rho + q0[1];
}

// src/ReedsShepp.cpp:561-574
static void rsSample(const double q0[3], const double q1[3], double rho, double step_size, double &length,
                     std::vector<std::array<double, 3>> &points)
{
    RsPath path = reedsShepp(q0, q1, rho);
    length = rho * path.total;
    for (double seg = 0.0; seg <= length; seg += step_size)
    {
        std::array<double, 3> q{};
        rsInterpolate(q0, path, seg / rho, rho, q.data());
        points.push_back(q);
    }
}

// ------------------------------------------------------------ OMPL's Dubins distance
// src/lookup_table.cpp:152, :174 call ompl::base::DubinsStateSpace::distance, which is not in the reference tree (OMPL is
// a system dependency, version unpinned: CMakeLists.txt:40).  Restated from OMPL's published source,
// src/ompl/base/spaces/src/DubinsStateSpace.cpp (releases 1.4 - 1.6): one function per word as there, double arithmetic,
// OMPL's tolerances.  PARITY UNPINNED: no file of the reference holds an output of this function.
namespace ompl_dubins
{
const double twopi = 2. * kPi;
const double DUBINS_EPS = 1e-6;
const double DUBINS_ZERO = -1e-7;
struct DubinsPath
{
    double length_[3];
    DubinsPath(double t = 0., double p = std::numeric_limits<double>::max(), double q = 0.) { length_[0] = t, length_[1] = p, length_[2] = q; }
    double length() const { return length_[0] + length_[1] + length_[2]; }
};
static inline double mod2pi(double x)
{
    if (x < 0 && x > DUBINS_ZERO)
        return 0;
    double xm = x - twopi * std::floor(x / twopi);
    if (twopi - xm < .5 * DUBINS_EPS)
        xm = 0.;
    return xm;
}
static DubinsPath dubinsLSL(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sa - sb));
    if (tmp >= DUBINS_ZERO)
    {
        double theta = m_atan2(cb - ca, d + sa - sb);
        double t = mod2pi(-alpha + theta);
        double p = std::sqrt(std::max(tmp, 0.));
        double q = mod2pi(beta - theta);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubinsRSR(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sb - sa));
    if (tmp >= DUBINS_ZERO)
    {
        double theta = m_atan2(ca - cb, d - sa + sb);
        double t = mod2pi(alpha - theta);
        double p = std::sqrt(std::max(tmp, 0.));
        double q = mod2pi(-beta + theta);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubinsRSL(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = d * d - 2. + 2. * (ca * cb + sa * sb - d * (sa + sb));
    if (tmp >= DUBINS_ZERO)
    {
        double p = std::sqrt(std::max(tmp, 0.));
        double theta = m_atan2(ca + cb, d - sa - sb) - m_atan2(2., p);
        double t = mod2pi(alpha - theta);
        double q = mod2pi(beta - theta);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubinsLSR(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = -2. + d * d + 2. * (ca * cb + sa * sb + d * (sa + sb));
    if (tmp >= DUBINS_ZERO)
    {
        double p = std::sqrt(std::max(tmp, 0.));
        double theta = m_atan2(-ca - cb, d + sa + sb) - m_atan2(-2., p);
        double t = mod2pi(-alpha + theta);
        double q = mod2pi(-beta + theta);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubinsRLR(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb + d * (sa - sb)));
    if (std::fabs(tmp) < 1.)
    {
        double p = twopi - m_acos(tmp);
        double theta = m_atan2(ca - cb, d - sa + sb);
        double t = mod2pi(alpha - theta + .5 * p);
        double q = mod2pi(alpha - beta - t + p);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubinsLRL(double d, double alpha, double beta)
{
    double ca = m_cos(alpha), sa = m_sin(alpha), cb = m_cos(beta), sb = m_sin(beta);
    double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb - d * (sa - sb)));
    if (std::fabs(tmp) < 1.)
    {
        double p = twopi - m_acos(tmp);
        double theta = m_atan2(-ca + cb, d + sa - sb);
        double t = mod2pi(-alpha + theta + .5 * p);
        double q = mod2pi(beta - alpha - t + p);
        return DubinsPath(t, p, q);
    }
    return DubinsPath();
}
static DubinsPath dubins(double d, double alpha, double beta)
{
    if (d < DUBINS_EPS && std::fabs(alpha - beta) < DUBINS_EPS)
        return DubinsPath(0, d, 0);
    DubinsPath path(dubinsLSL(d, alpha, beta)), tmp(dubinsRSR(d, alpha, beta));
    double len, minLength = path.length();
    if ((len = tmp.length()) < minLength)
    {
        minLength = len;
        path = tmp;
    }
    tmp = dubinsRSL(d, alpha, beta);
    if ((len = tmp.length()) < minLength)
    {
        minLength = len;
        path = tmp;
    }
    tmp = dubinsLSR(d, alpha, beta);
    if ((len = tmp.length()) < minLength)
    {
        minLength = len;
        path = tmp;
    }
    tmp = dubinsRLR(d, alpha, beta);
    if ((len = tmp.length()) < minLength)
    {
        minLength = len;
        path = tmp;
    }
    tmp = dubinsLRL(d, alpha, beta);
    if ((len = tmp.length()) < minLength)
        path = tmp;
    return path;
}
// DubinsStateSpace::distance for the default (not symmetric) space of turning radius rho
static double distance(double x1, double y1, double th1, double x2, double y2, double th2, double rho)
{
    double dx = x2 - x1, dy = y2 - y1, d = std::sqrt(dx * dx + dy * dy) / rho, th = m_atan2(dy, dx);
    double alpha = mod2pi(th1 - th), beta = mod2pi(th2 - th);
    return rho * dubins(d, alpha, beta).length();
}
} // namespace ompl_dubins

// ----------------------------------------------------------------------- Dubins
// src/dubins.cpp:81-314, float arithmetic as the reference's <math.h> float
// overloads give it; mixed double sub-expressions are kept (atan2(-2.0, p),
// 2*M_PI - acos(..), 6. - d*d ...).
namespace db
{
struct Path
{
    float qi[3];
    float param[3];
    float rho;
    int type;
};
static const int DIRDATA[6][3] = {{0, 1, 0}, {0, 1, 2}, {2, 1, 0}, {2, 1, 2}, {2, 0, 2}, {0, 2, 0}};

static float fmodr(float x, float y) { return x - y * m_floorf(x / y); }
static float mod2pi(float theta) { return fmodr(theta, (float)(2 * kPi)); }

struct In
{
    float sa, sb, ca, cb, c_ab;
    In(float alpha, float beta)
        : sa(m_sinf(alpha)), sb(m_sinf(beta)), ca(m_cosf(alpha)), cb(m_cosf(beta)), c_ab(m_cosf(alpha - beta))
    {
    }
};
static bool LSL(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float tmp0 = d + i.sa - i.sb;
    float p_squared = 2 + (d * d) - (2 * i.c_ab) + (2 * d * (i.sa - i.sb));
    if (p_squared < 0)
        return false;
    float tmp1 = m_atan2f((i.cb - i.ca), tmp0);
    o[0] = mod2pi(-alpha + tmp1);
    o[1] = m_sqrtf(p_squared);
    o[2] = mod2pi(beta - tmp1);
    return true;
}
static bool RSR(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float tmp0 = d - i.sa + i.sb;
    float p_squared = 2 + (d * d) - (2 * i.c_ab) + (2 * d * (i.sb - i.sa));
    if (p_squared < 0)
        return false;
    float tmp1 = m_atan2f((i.ca - i.cb), tmp0);
    o[0] = mod2pi(alpha - tmp1);
    o[1] = m_sqrtf(p_squared);
    o[2] = mod2pi(-beta + tmp1);
    return true;
}
static bool LSR(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float p_squared = -2 + (d * d) + (2 * i.c_ab) + (2 * d * (i.sa + i.sb));
    if (p_squared < 0)
        return false;
    float p = m_sqrtf(p_squared);
    float tmp2 = (float)((double)m_atan2f((-i.ca - i.cb), (d + i.sa + i.sb)) - m_atan2(-2.0, (double)p));
    o[0] = mod2pi(-alpha + tmp2);
    o[1] = p;
    o[2] = mod2pi(-mod2pi(beta) + tmp2);
    return true;
}
static bool RSL(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float p_squared = (d * d) - 2 + (2 * i.c_ab) - (2 * d * (i.sa + i.sb));
    if (p_squared < 0)
        return false;
    float p = m_sqrtf(p_squared);
    float tmp2 = (float)((double)m_atan2f((i.ca + i.cb), (d - i.sa - i.sb)) - m_atan2(2.0, (double)p));
    o[0] = mod2pi(alpha - tmp2);
    o[1] = p;
    o[2] = mod2pi(beta - tmp2);
    return true;
}
static bool RLR(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float tmp_rlr = (float)((6. - (double)(d * d) + (double)(2 * i.c_ab) + (double)(2 * d * (i.sa - i.sb))) / 8.);
    if (hy_fabs((double)tmp_rlr) > 1)
        return false;
    float p = mod2pi((float)(2 * kPi - (double)m_acosf(tmp_rlr)));
    float t = mod2pi(alpha - m_atan2f(i.ca - i.cb, d - i.sa + i.sb) + mod2pi((float)((double)p / 2.)));
    float q = mod2pi(alpha - beta - t + mod2pi(p));
    o[0] = t, o[1] = p, o[2] = q;
    return true;
}
static bool LRL(float alpha, float beta, float d, float *o)
{
    In i(alpha, beta);
    float tmp_lrl = (float)((6. - (double)(d * d) + (double)(2 * i.c_ab) + (double)(2 * d * (-i.sa + i.sb))) / 8.);
    if (hy_fabs((double)tmp_lrl) > 1)
        return false;
    float p = mod2pi((float)(2 * kPi - (double)m_acosf(tmp_lrl)));
    float t = mod2pi((float)((double)(-alpha - m_atan2f(i.ca - i.cb, d + i.sa - i.sb)) + (double)p / 2.));
    float q = mod2pi(mod2pi(beta) - alpha - t + mod2pi(p));
    o[0] = t, o[1] = p, o[2] = q;
    return true;
}
typedef bool (*Word)(float, float, float, float *);
static const Word words[6] = {LSL, LSR, RSL, RSR, RLR, LRL};

// src/dubins.cpp:91-140; returns 0 on success
static int init(const float q0[3], const float q1[3], float rho, Path *path)
{
    float dx = q1[0] - q0[0];
    float dy = q1[1] - q0[1];
    float D = m_sqrtf(dx * dx + dy * dy);
    float d = D / rho;
    if (rho <= 0.)
        return 3;
    float theta = mod2pi(m_atan2f(dy, dx));
    float alpha = mod2pi(q0[2] - theta);
    float beta = mod2pi(q1[2] - theta);
    for (int i = 0; i < 3; i++)
        path->qi[i] = q0[i];
    path->rho = rho;
    float best_cost = INFINITY;
    int best_word = -1;
    for (int i = 0; i < 6; i++)
    {
        float params[3];
        if (words[i](alpha, beta, d, params))
        {
            float cost = params[0] + params[1] + params[2];
            if (cost < best_cost)
            {
                best_word = i;
                best_cost = cost;
                path->param[0] = params[0];
                path->param[1] = params[1];
                path->param[2] = params[2];
                path->type = i;
            }
        }
    }
    if (best_word == -1)
        return 4;
    path->type = best_word;
    return 0;
}
static float length(const Path *path)
{
    float length = 0.;
    length += path->param[0];
    length += path->param[1];
    length += path->param[2];
    length = length * path->rho;
    return length;
}
static void segment(float t, const float qi[3], float qt[3], int type)
{
    if (type == 0)
    {
        qt[0] = qi[0] + m_sinf(qi[2] + t) - m_sinf(qi[2]);
        qt[1] = qi[1] - m_cosf(qi[2] + t) + m_cosf(qi[2]);
        qt[2] = qi[2] + t;
    }
    else if (type == 2)
    {
        qt[0] = qi[0] - m_sinf(qi[2] - t) + m_sinf(qi[2]);
        qt[1] = qi[1] + m_cosf(qi[2] - t) - m_cosf(qi[2]);
        qt[2] = qi[2] - t;
    }
    else
    {
        qt[0] = qi[0] + m_cosf(qi[2]) * t;
        qt[1] = qi[1] + m_sinf(qi[2]) * t;
        qt[2] = qi[2];
    }
}
static int sample(const Path *path, float t, float q[3])
{
    if (t < 0 || t >= length(path))
        return 2;
    float tprime = t / path->rho;
    float qi[3] = {0, 0, path->qi[2]};
    const int *types = DIRDATA[path->type];
    float p1 = path->param[0];
    float p2 = path->param[1];
    float q1[3], q2[3];
    segment(p1, qi, q1, types[0]);
    segment(p2, q1, q2, types[1]);
    if (tprime < p1)
        segment(tprime, qi, q, types[0]);
    else if (tprime < (p1 + p2))
        segment(tprime - p1, q1, q, types[1]);
    else
        segment(tprime - p1 - p2, q2, q, types[2]);
    q[0] = q[0] * path->rho + path->qi[0];
    q[1] = q[1] * path->rho + path->qi[1];
    q[2] = mod2pi(q[2]);
    return 0;
}
} // namespace db

// ------------------------------------------------------------ costmap_2d facts
// costmap_2d::Costmap2D is a ROS dependency absent from the reference tree;
// restated: worldToMap rejects wx<origin or wy<origin, truncates
// (w-origin)/resolution to int, accepts iff mx<size_x && my<size_y.
// ====================================================================== cubic Bezier (curve_type 2)
// Two different pieces of the reference are involved.  The heuristic table (src/lookup_table.cpp:226-277) uses the
// third-party header include/bezier.h (Bezier::Bezier<3>::length, :744-768, Bernstein form with pow, :100-103,
// :681-690); the analytic shot (src/hybrid_a_star.cpp:470-521) uses CubicBezier::CubicBezier (src/cubic_bezier.cpp),
// Eigen matrix products of the geometry matrix, the basis matrix and a power vector.
// Assumptions that no reference output pins (parity unpinned): std::pow / powf return the correctly rounded value for
// the exponents 0..3 used here; Eigen is a 3.3 / 3.4 release, whose small fixed-size products take each coefficient as
// (row . column).sum() with the unrolled scalar reduction, which halves the range: (p0 + p1) + (p2 + p3)
// (Eigen/src/Core/ProductEvaluators.h product_evaluator::coeff, Eigen/src/Core/Redux.h redux_novec_unroller).
namespace bez
{
// std::pow(double, k), k = 0..3, and its float overload
static inline double pow_d(double u, unsigned k) { return k == 0 ? 1.0 : k == 1 ? u : k == 2 ? u * u : (u * u) * u; }
static inline float pow_f(float t, unsigned k) { return (float)pow_d((double)t, k); }

// include/bezier.h: Bezier<3> over Vec2 (float) control points
struct Lib3
{
    float px[4], py[4];
    // PolynomialPair::valueAt (:100-103): pow(1.0f - t, size_t) is the promoting (double) overload, pow(t, float) powf
    static float poly(unsigned n, float t) { return (float)(pow_d((double)(1.0f - t), 3 - n) * (double)pow_f(t, n)); }
    float valueAt(float t, int axis) const // :681-690
    {
        static const size_t binom[4] = {1, 3, 3, 1};
        float sum = 0;
        for (unsigned n = 0; n < 4; n++)
            sum += binom[n] * poly(n, t) * (axis == 0 ? px[n] : py[n]);
        return sum;
    }
    float length(size_t intervals = 100) const // :744-768
    {
        float length = 0.0f;
        float t = 0.0f;
        const float dt = 1.0f / (float)intervals;
        float p1x = valueAt(t, 0), p1y = valueAt(t, 1);
        for (size_t i = 0; i < intervals; i++)
        {
            float p2x = valueAt(t + dt, 0), p2y = valueAt(t + dt, 1);
            float x = p2x - p1x, y = p2y - p1y;
            length += m_sqrtf(x * x + y * y);
            p1x = p2x, p1y = p2y;
            t += dt;
        }
        return length;
    }
};
// src/lookup_table.cpp:250-262: the curve from (0, 0, start angle 0) to (x, y, theta) and its length
static float tableCost(float x, float y, float theta)
{
    const float start_angle = 0;
    float dx = m_cosf(start_angle), dy = m_sinf(start_angle);
    const float gx = x - 0.f, gy = y - 0.f;
    float t = m_sqrtf(gx * gx + gy * gy) / 3; // (goal - start).length() / 3
    Lib3 c;
    c.px[0] = 0, c.py[0] = 0;
    c.px[1] = 0 + dx * t, c.py[1] = 0 + dy * t;
    dx = m_cosf(theta), dy = m_sinf(theta);
    c.px[2] = x - dx * t, c.py[2] = y - dy * t;
    c.px[3] = x, c.py[3] = y;
    return c.length();
}

// src/cubic_bezier.cpp + include/cubic_bezier.h: Eigen::Matrix<float, 3, 4> geometry, 4 x 4 basis
struct Cubic
{
    float G[3][4], GB[3][4];
    Cubic(const float start[3], const float goal[3]) // :31-44, CalculateControlPoints :70-136 (use_random_ false)
    {
        static const float B[4][4] = {{1, -3, 3, -1}, {0, 3, -6, 3}, {0, 0, 3, -3}, {0, 0, 0, 1}};
        float dir[3] = {m_cosf(start[2]), m_sinf(start[2]), 0};
        const float tx = goal[0] - start[0], ty = goal[1] - start[1];
        const float t = m_sqrtf(tx * tx + ty * ty) / 3;
        float c1[3], c2[3];
        for (int i = 0; i < 3; i++)
            c1[i] = start[i] + dir[i] * t;
        dir[0] = m_cosf(goal[2]), dir[1] = m_sinf(goal[2]);
        for (int i = 0; i < 3; i++)
            c2[i] = goal[i] - dir[i] * t;
        for (int i = 0; i < 3; i++)
            G[i][0] = start[i], G[i][1] = c1[i], G[i][2] = c2[i], G[i][3] = goal[i];
        for (int i = 0; i < 3; i++) // geometrical_constraint_matrix_ * basis_matrix_
            for (int j = 0; j < 4; j++)
                GB[i][j] = (G[i][0] * B[0][j] + G[i][1] * B[1][j]) + (G[i][2] * B[2][j] + G[i][3] * B[3][j]);
    }
    void times(const float c[4], float out[3]) const
    {
        for (int i = 0; i < 3; i++)
            out[i] = (GB[i][0] * c[0] + GB[i][1] * c[1]) + (GB[i][2] * c[2] + GB[i][3] * c[3]);
    }
    // :16-51; `i - 1 < 0` never holds for the unsigned i, so the i = 0 terms are 0 * pow(t, huge) = 0 for t <= 1
    static void coef(float t, float c[4])
    {
        for (unsigned i = 0; i < 4; ++i)
            c[i] = (float)pow_d((double)t, i);
    }
    static void coef1(float t, float c[4])
    {
        c[0] = 0;
        for (unsigned i = 1; i < 4; ++i)
            c[i] = (float)((double)i * pow_d((double)t, i - 1));
    }
    static void coef2(float t, float c[4])
    {
        c[0] = 0, c[1] = 0;
        for (unsigned i = 2; i < 4; ++i)
            c[i] = (float)((double)(i * (i - 1)) * pow_d((double)t, i - 2));
    }
    void GetValueAt(float t, float out[3]) const
    {
        float c[4];
        coef(t, c);
        times(c, out);
    }
    float GetAngleAt(float t) const // :158-165
    {
        float c[4], d[3];
        coef1(t, c);
        times(c, d);
        return m_atan2f(d[1], d[0]);
    }
    float GetLength() const // :166-178
    {
        float length = 0;
        for (unsigned i = 0; i < 100; ++i)
        {
            float a[3], b[3];
            GetValueAt((float)((i + 1) / 100.0), a);
            GetValueAt((float)(i / 100.0), b);
            const float x = a[0] - b[0], y = a[1] - b[1];
            length += m_sqrtf(x * x + y * y);
        }
        return length;
    }
    float GetCurvatureAt(float t) const // :224-241
    {
        float c[4], f[3], s[3];
        coef1(t, c);
        times(c, f);
        coef2(t, c);
        times(c, s);
        const float fn = m_sqrtf(f[0] * f[0] + f[1] * f[1]);
        if (fn != 0)
            return (float)((double)m_fabsf(f[0] * s[1] - s[0] * f[1]) / pow_d((double)fn, 3));
        return 100000;
    }
};
} // namespace bez

struct Costmap
{
    uint32_t w = 0, h = 0;
    double res = 1, ox = 0, oy = 0;
    std::vector<uint8_t> cost;
    bool worldToMap(double wx, double wy, unsigned int &mx, unsigned int &my) const
    {
        if (wx < ox || wy < oy)
            return false;
        mx = (int)((wx - ox) / res);
        my = (int)((wy - oy) / res);
        return mx < w && my < h;
    }
};

struct Node3D
{
    float x = 0, y = 0, t = 0, g = 999, h = 0;
    int idx = -1;
    bool o = false, c = false, reverse = false;
    int prim = 0;
    int pred = -1; // index into the query's node arena (shared_ptr in the reference)
};

// include/node3d.h:65-72
static int setIdx3D(const Node3D &n, int width, int height, float delta_heading_in_rad, float resolution,
                    float origin_x, float origin_y)
{
    return (int)(m_floorf(m_roundf(n.t / delta_heading_in_rad * 1000) / 1000) * width * height +
                 m_floorf(m_roundf(((n.y - origin_y) / resolution) * 1000) / 1000) * width +
                 m_floorf(m_roundf(((n.x - origin_x) / resolution) * 1000) / 1000));
}
// include/node2d.h:59-66
static int setIdx2D(float x, float y, int width, float resolution, float origin_x, float origin_y)
{
    return (int)(m_floorf(m_roundf(((y - origin_y) / resolution) * 1000) / 1000) * width +
                 m_floorf(m_roundf(((x - origin_x) / resolution) * 1000) / 1000));
}

// Footprint table entry, include/constants.h:62-80
struct FpConfig
{
    int32_t length;
    int32_t pos[64][2];
};

struct Ctx
{
    hyas_params P;
    Costmap cm;
    // CollisionDetection members (include/collisiondetection.h:299-349)
    float resolution_ = 1, origin_x_ = 0, origin_y_ = 0;
    std::vector<int8_t> grid;              // nav_msgs::OccupancyGrid data
    std::vector<Polygon> obstacle_vec_;
    std::unordered_map<uint32_t, std::vector<Polygon>> in_range_obstacle_map_;
    // adaptive step size / steering (include/collisiondetection.h:328-345)
    std::unordered_map<uint32_t, float> min_distance_map_, in_range_obstacle_density_map_, normalized_obstacle_density_map_;
    std::unordered_map<uint32_t, std::vector<std::pair<float, AngleRange>>> distance_angle_range_map_;
    // LookupTable: keyed dense store; NaN = key never emplaced
    std::vector<float> table;
    int table_curve = -1;
    // Mode F
    std::vector<FpConfig> fp;
    int fp_bb = 0;
    // last plan
    std::vector<float> last_path;
    std::vector<int32_t> last_expanded;
    bool linear_duplicate_check = true;

    // ------------------------------------------------------- map ingest
    // src/utility.cpp:166-206
    void translateCostmap()
    {
        int8_t tab[256];
        tab[0] = 0, tab[253] = 99, tab[254] = 100, tab[255] = -1;
        for (int i = 1; i < 253; i++)
            tab[i] = (int8_t)(1 + (97 * (i - 1)) / 251);
        grid.resize((size_t)cm.w * cm.h);
        for (size_t i = 0; i < grid.size(); ++i)
            grid[i] = tab[cm.cost[i]];
    }
    // src/collisiondetection.cpp:362-369
    uint32_t GetNode3DIndexOnGridMap(float x, float y) const
    {
        return (uint32_t)(m_floorf(m_roundf(((y - origin_y_) / resolution_) * 1000) / 1000) * cm.w +
                          m_floorf(m_roundf(((x - origin_x_) / resolution_) * 1000) / 1000));
    }
    // src/collisiondetection.cpp:262-317
    void SetObstacleVec()
    {
        obstacle_vec_.clear();
        float x_limit = origin_x_ + cm.w * resolution_;
        float y_limit = origin_y_ + cm.h * resolution_;
        for (float x = origin_x_; x < x_limit;)
        {
            for (float y = origin_y_; y < y_limit;)
            {
                uint32_t gi = GetNode3DIndexOnGridMap(x, y);
                if (gi < grid.size() && grid[gi])
                {
                    Polygon cur = CreatePolygon(V2{x, y}, resolution_, resolution_);
                    if (!obstacle_vec_.empty() && IsPolygonInNeighbor(obstacle_vec_.back(), cur))
                    {
                        Polygon combined = CombinePolygon(cur, obstacle_vec_.back());
                        obstacle_vec_.pop_back();
                        obstacle_vec_.push_back(combined);
                    }
                    else
                        obstacle_vec_.push_back(cur);
                }
                y = y + resolution_;
            }
            x = x + resolution_;
        }
        if (P.map_boundary_obstacle)
            AddMapBoundaryAsObstacle(obstacle_vec_);
    }
    // src/collisiondetection.cpp:1223-1240: the rectangles are given in cells (grid_ptr_->info.width / height)
    void AddMapBoundaryAsObstacle(std::vector<Polygon> &obstacle_vec)
    {
        obstacle_vec.push_back(CreatePolygon(V2{0, -1}, (float)cm.w, 1));
        obstacle_vec.push_back(CreatePolygon(V2{(float)cm.w, -1}, 1, (float)(cm.h + 2)));
        obstacle_vec.push_back(CreatePolygon(V2{-1, (float)cm.h}, (float)(cm.w + 2), 1));
        obstacle_vec.push_back(CreatePolygon(V2{-1, 0}, 1, (float)cm.h));
    }
    // src/collisiondetection.cpp:319-354
    void SetInRangeObstacle()
    {
        in_range_obstacle_map_.clear();
        float x_limit = origin_x_ + cm.w * resolution_;
        float y_limit = origin_y_ + cm.h * resolution_;
        std::vector<Polygon> v;
        for (float x = origin_x_; x < x_limit;)
        {
            for (float y = origin_y_; y < y_limit;)
            {
                uint32_t idx = GetNode3DIndexOnGridMap(x, y);
                v.clear();
                for (const Polygon &poly : obstacle_vec_)
                {
                    float distance = GetDistanceFromPolygonToPoint(poly, V2{x, y});
                    if (distance < P.obstacle_detection_range)
                        v.push_back(poly);
                }
                in_range_obstacle_map_.emplace(idx, v);
                y = y + resolution_;
            }
            x = x + resolution_;
        }
    }

    // ------------------------------------- adaptive step size / steering: maps built once per costmap
    // src/collisiondetection.cpp:1403-1447
    void SetMinDistanceMap()
    {
        min_distance_map_.clear();
        float x_limit = origin_x_ + cm.w * resolution_;
        float y_limit = origin_y_ + cm.h * resolution_;
        float min_distance = 1000, distance = 0;
        for (float x = origin_x_; x < x_limit;)
        {
            for (float y = origin_y_; y < y_limit;)
            {
                uint32_t current_point_index = GetNode3DIndexOnGridMap(x, y);
                auto it = in_range_obstacle_map_.find(current_point_index);
                if (it != in_range_obstacle_map_.end())
                    for (const Polygon &obstacle : it->second)
                    {
                        distance = GetDistanceFromPolygonToPoint(obstacle, V2{x, y});
                        if (min_distance > distance)
                            min_distance = distance;
                    }
                min_distance_map_.emplace(current_point_index, min_distance);
                y = y + resolution_;
                min_distance = 1000;
            }
            x = x + resolution_;
        }
    }
    // src/collisiondetection.cpp:973-1011
    float FindNoCollisionDistance(float nx, float ny, float radius, float angle)
    {
        float out = radius;
        uint32_t current_index = GetNode3DIndexOnGridMap(nx, ny);
        auto it = in_range_obstacle_map_.find(current_index);
        if (it != in_range_obstacle_map_.end())
            for (const Polygon &in_range_polygon : it->second)
            {
                float temp = GetDistanceFromPolygonToPointAtAngle(in_range_polygon, V2{nx, ny}, angle);
                if (temp < 0)
                    continue;
                if (out > temp)
                    out = temp;
            }
        return out;
    }
    // src/collisiondetection.cpp:940-970 with consider_steering_angle == false: {angle, free distance} every degree
    std::vector<std::pair<float, float>> SweepDistanceAndAngle(float nx, float ny, float radius)
    {
        std::vector<std::pair<float, float>> out;
        AngleRange steering_angle_range(0, ConvertDegToRad(360));
        float angle = steering_angle_range.first;
        for (; angle <= steering_angle_range.first + steering_angle_range.second; angle = angle + ConvertDegToRad(1))
            out.emplace_back(angle, FindNoCollisionDistance(nx, ny, radius, angle));
        return out;
    }
    // src/collisiondetection.cpp:387-500
    void SetDistanceAngleRangeMap()
    {
        distance_angle_range_map_.clear();
        const float obstacle_detection_range_ = P.obstacle_detection_range;
        std::vector<std::pair<float, AngleRange>> distance_angle_range_vec;
        float x_limit = origin_x_ + cm.w * resolution_;
        float y_limit = origin_y_ + cm.h * resolution_;
        for (float x = origin_x_; x < x_limit;)
        {
            for (float y = origin_y_; y < y_limit;)
            {
                distance_angle_range_vec.clear();
                uint32_t current_point_index = GetNode3DIndexOnGridMap(x, y);
                std::vector<std::pair<float, float>> angle_distance_vec = SweepDistanceAndAngle(x, y, obstacle_detection_range_);
                float range_start = -1, range_range = -1, min_distance = 0;
                auto md = min_distance_map_.find(current_point_index);
                if (md != min_distance_map_.end())
                    min_distance = md->second;
                bool free_angle_range_flag = false, obstacle_angle_range_flag = false;
                for (const auto &pair : angle_distance_vec)
                {
                    if (pair.second == obstacle_detection_range_)
                    {
                        if (obstacle_angle_range_flag)
                        {
                            obstacle_angle_range_flag = false;
                            free_angle_range_flag = true;
                            range_range = pair.first - range_start;
                            if (range_start != -1 && range_range != -1)
                            {
                                distance_angle_range_vec.emplace_back(min_distance, AngleRange(range_start, range_range));
                                range_start = pair.first;
                            }
                            else
                                range_start = pair.first;
                        }
                        if (!free_angle_range_flag)
                        {
                            free_angle_range_flag = true;
                            obstacle_angle_range_flag = false;
                            range_start = pair.first;
                        }
                    }
                    else
                    {
                        if (free_angle_range_flag)
                        {
                            free_angle_range_flag = false;
                            obstacle_angle_range_flag = true;
                            range_range = pair.first - range_start;
                            if (range_start != -1 && range_range != -1)
                            {
                                distance_angle_range_vec.emplace_back(obstacle_detection_range_, AngleRange(range_start, range_range));
                                range_start = pair.first;
                            }
                            else
                                range_start = pair.first;
                        }
                        if (!obstacle_angle_range_flag)
                        {
                            free_angle_range_flag = false;
                            obstacle_angle_range_flag = true;
                            range_start = pair.first;
                        }
                    }
                    if (pair.first == angle_distance_vec.back().first)
                    {
                        range_range = pair.first - range_start;
                        if (range_start != -1 && range_range != -1 && range_range != 0)
                            distance_angle_range_vec.emplace_back(min_distance, AngleRange(range_start, range_range));
                    }
                }
                distance_angle_range_map_.emplace(current_point_index, distance_angle_range_vec);
                y = y + resolution_;
            }
            x = x + resolution_;
        }
    }
    // src/collisiondetection.cpp:1013-1103, consider_steering_angle_range_for_obstacle_density == false
    void BuildObstacleDensityMap()
    {
        in_range_obstacle_density_map_.clear();
        float x_limit = origin_x_ + cm.w * resolution_;
        float y_limit = origin_y_ + cm.h * resolution_;
        for (float x = origin_x_; x < x_limit;)
        {
            for (float y = origin_y_; y < y_limit;)
            {
                uint32_t current_point_index = GetNode3DIndexOnGridMap(x, y);
                auto it = in_range_obstacle_map_.find(current_point_index);
                if (it != in_range_obstacle_map_.end())
                    in_range_obstacle_density_map_.emplace(current_point_index, (float)it->second.size());
                y = y + resolution_;
            }
            x = x + resolution_;
        }
    }
    // src/collisiondetection.cpp:1110-1151
    void BuildNormalizedObstacleDensityMap()
    {
        normalized_obstacle_density_map_.clear();
        float normalized_obstacle_density, delta = 0.01;
        float max_density = 0, min_density = 1000000;
        for (const auto &element : in_range_obstacle_density_map_)
        {
            if (element.second > max_density)
                max_density = element.second;
            if (element.second < min_density)
                min_density = element.second;
        }
        for (const auto &element : in_range_obstacle_density_map_)
        {
            normalized_obstacle_density = (element.second - min_density) / (max_density - min_density + delta);
            normalized_obstacle_density_map_.emplace(element.first, normalized_obstacle_density);
        }
    }
    // the hybrid_astar part of UpdateGrid, src/collisiondetection.cpp:39-46
    void buildAdaptiveMaps()
    {
        SetMinDistanceMap();
        SetDistanceAngleRangeMap();
        BuildObstacleDensityMap();
        BuildNormalizedObstacleDensityMap();
    }

    // ------------------------------------- adaptive step size / steering: per node
    // src/collisiondetection.cpp:1153-1177 (index by grid cell: the default)
    float GetNormalizedObstacleDensity(const Node3D &node3d)
    {
        float obstacle_density = 0;
        auto it = normalized_obstacle_density_map_.find(GetNode3DIndexOnGridMap(node3d.x, node3d.y));
        if (it != normalized_obstacle_density_map_.end())
            obstacle_density = it->second;
        return obstacle_density;
    }
    // src/collisiondetection.cpp:1206-1221
    static float GetStepSizeWeight(float normalized_obstacle_density) { return (float)(-0.8 * (double)normalized_obstacle_density + 0.9); }
    // src/collisiondetection.cpp:1364-1381
    static float LimitSteeringAngle(float steering_angle, float steering_angle_limit)
    {
        if (steering_angle > steering_angle_limit)
            return steering_angle_limit;
        else if (steering_angle < -steering_angle_limit)
            return -steering_angle_limit;
        return ConvertDegToRad(m_roundf(ConvertRadToDeg(steering_angle) * 10) / 10);
    }
    // src/collisiondetection.cpp:688-725
    std::vector<std::pair<float, AngleRange>> FindFreeAngleRangeAndObstacleAngleRange(const Node3D &node3d)
    {
        std::vector<std::pair<float, AngleRange>> out;
        AngleRange steering_angle_range(0, ConvertDegToRad(359.9999f));
        auto it = distance_angle_range_map_.find(GetNode3DIndexOnGridMap(node3d.x, node3d.y));
        if (it != distance_angle_range_map_.end())
            for (const auto &pair : it->second)
            {
                AngleRange common = FindCommonAngleRange(pair.second, steering_angle_range);
                if (common.first != -1 && common.second != -1)
                    out.emplace_back(pair.first, common);
            }
        return out;
    }
    // src/collisiondetection.cpp:1449-1484
    float FindStepSize(const Node3D &pred, float steering_angle, const Node3D &goal, float fixed_step_size)
    {
        float step_size = 0, min_distance_to_obstacle = P.obstacle_detection_range;
        float distance_to_goal = GetDistance(pred, goal);
        std::vector<std::pair<float, AngleRange>> step_size_angle_range_vec = FindFreeAngleRangeAndObstacleAngleRange(pred);
        float final_orientation = RadNormalization(pred.t + steering_angle);
        for (const auto &element : step_size_angle_range_vec)
            if (IsAngleRangeInclude(element.second, final_orientation))
            {
                min_distance_to_obstacle = element.first;
                break;
            }
        float weight_step_size = GetStepSizeWeight(GetNormalizedObstacleDensity(pred));
        step_size = weight_step_size * min_distance_to_obstacle;
        if (step_size < fixed_step_size)
            step_size = fixed_step_size;
        if (step_size > distance_to_goal)
            step_size = distance_to_goal;
        return step_size;
    }
    // src/collisiondetection.cpp:1263-1291
    std::pair<float, float> AddOneMoreStepSizeAndSteeringAngle(float angle_to_goal, float step_size, const Node3D &pred, const Node3D &goal)
    {
        float distance_to_goal = GetDistance(pred, goal);
        float new_step_size = (distance_to_goal < step_size) ? distance_to_goal : step_size;
        return std::pair<float, float>(new_step_size, LimitSteeringAngle(angle_to_goal, ConvertDegToRad(30)));
    }
    // src/collisiondetection.cpp:1293-1362.  Without a free range among the node's own ranges the reference fills
    // two shadowing locals and compares the untouched outer pair (10000 > 10000): angle1 is returned.
    float IsCloseToFreeAngleRange(const std::vector<std::pair<float, AngleRange>> &available_angle_range_vec, float angle1, float angle2)
    {
        const float obstacle_detection_range_ = P.obstacle_detection_range;
        bool free_angle_range_in_steering_angle = false;
        float distance_angle1 = 10000, distance_angle2 = 10000;
        for (const auto &pair : available_angle_range_vec)
            if (pair.first == obstacle_detection_range_)
                free_angle_range_in_steering_angle = true;
        if (free_angle_range_in_steering_angle)
            for (const auto &pair : available_angle_range_vec)
                if (pair.first == obstacle_detection_range_)
                {
                    float temp1 = GetAngleDistance(angle1, pair.second), temp2 = GetAngleDistance(angle2, pair.second);
                    distance_angle1 = std::min(temp1, distance_angle1);
                    distance_angle2 = std::min(temp2, distance_angle2);
                }
        if (distance_angle1 > distance_angle2)
            return angle2;
        return angle1;
    }
    // src/collisiondetection.cpp:728-938
    std::vector<std::pair<float, float>> SelectStepSizeAndSteeringAngle(const std::vector<std::pair<float, AngleRange>> &available_angle_range_vec,
                                                                        const Node3D &pred, const Node3D &goal, int number_of_successor,
                                                                        float fixed_step_size, float distance_start_to_goal)
    {
        const float obstacle_detection_range_ = P.obstacle_detection_range;
        std::vector<std::pair<float, float>> out;
        float steering_angle, distance_to_goal = GetDistance(pred, goal);
        float angle_to_goal, orientation_diff, steering_angle_toward_goal = 0;
        if (P.add_one_more_successor)
        {
            float weighting = distance_to_goal / distance_start_to_goal;
            angle_to_goal = GetAngle(pred, goal);
            orientation_diff = -RadNormalization(pred.t - goal.t);
            steering_angle_toward_goal = -RadNormalization(weighting * angle_to_goal + (1 - weighting) * orientation_diff);
        }
        auto push = [&](const std::pair<float, float> &temp) {
            if (!DuplicateCheck(out, temp, false))
                out.emplace_back(temp);
        };
        for (const auto &pair : available_angle_range_vec)
        {
            if (pair.second.second == 0)
            {
                steering_angle = RadNormalization(pair.second.first - pred.t);
                steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                push(std::pair<float, float>(pair.first, steering_angle));
                continue;
            }
            if (pair.first == obstacle_detection_range_)
            {
                if (P.add_one_more_successor)
                    push(AddOneMoreStepSizeAndSteeringAngle(steering_angle_toward_goal,
                                                            FindStepSize(pred, steering_angle_toward_goal, goal, fixed_step_size), pred, goal));
                if (P.fixed_number_of_steering_angle_in_free_angle_range)
                {
                    steering_angle = RadNormalization((float)((double)pair.second.first + 0.5 * (double)pair.second.second - (double)pred.t));
                    steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                    push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                }
                else
                {
                    if (pair.second.second < ConvertDegToRad(3))
                    {
                        steering_angle = RadNormalization((float)((double)pair.second.first + (double)pair.second.second * 0.5 - (double)pred.t));
                        steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                        push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                    }
                    else
                    {
                        float current_angle = pair.second.first;
                        float steering_angle_step_size = ConvertDegToRad(5);
                        while (current_angle <= GetAngleRangeEnd(pair.second))
                        {
                            steering_angle = RadNormalization(current_angle - pred.t);
                            steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                            push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                            current_angle += steering_angle_step_size;
                        }
                    }
                }
            }
            else
            {
                if (P.add_one_more_successor && !P.add_one_more_successor_only_in_free_angle_range)
                    push(AddOneMoreStepSizeAndSteeringAngle(steering_angle_toward_goal,
                                                            FindStepSize(pred, steering_angle_toward_goal, goal, fixed_step_size), pred, goal));
                if (P.steering_angle_towards_free_angle_range_for_obstacle_angle_range)
                {
                    float closest_angle = IsCloseToFreeAngleRange(available_angle_range_vec, pair.second.first, GetAngleRangeEnd(pair.second));
                    steering_angle = RadNormalization(closest_angle - pred.t);
                    steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                    push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                }
                else
                {
                    steering_angle = RadNormalization(pair.second.first - pred.t);
                    steering_angle = LimitSteeringAngle(steering_angle, ConvertDegToRad(30));
                    push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                    steering_angle = RadNormalization(pair.second.first + pair.second.second - pred.t);
                    push(std::pair<float, float>(FindStepSize(pred, steering_angle, goal, fixed_step_size), steering_angle));
                }
            }
        }
        (void)number_of_successor;
        return out;
    }
    // src/collisiondetection.cpp:1383-1401
    std::vector<std::pair<float, float>> FindStepSizeAndSteeringAngle(const Node3D &pred, const Node3D &start, const Node3D &goal,
                                                                      int number_of_successor, float step_size)
    {
        float distance_start_to_goal = GetDistance(start, goal);
        std::vector<std::pair<float, AngleRange>> available_angle_range_vec = FindFreeAngleRangeAndObstacleAngleRange(pred);
        return SelectStepSizeAndSteeringAngle(available_angle_range_vec, pred, goal, number_of_successor, step_size, distance_start_to_goal);
    }

    // ----------------------------------------------- collision (mode POLYGON)
    // src/collisiondetection.cpp:178-188
    bool IsOnGrid(float x, float y) const
    {
        unsigned int mx, my;
        return cm.worldToMap(x, y, mx, my);
    }
    // src/collisiondetection.cpp:573-606
    bool CollisionCheckPolygon(const Polygon &polygon)
    {
        for (const V2 &pt : polygon)
            if (!IsOnGrid(pt.x, pt.y))
                return false;
        uint32_t key = GetNode3DIndexOnGridMap(polygon[0].x, polygon[0].y);
        auto it = in_range_obstacle_map_.find(key);
        if (it != in_range_obstacle_map_.end())
            for (const Polygon &obst : it->second)
                if (IsPolygonIntersectWithPolygon(polygon, obst))
                    return false;
        return true;
    }
    // src/collisiondetection.cpp:608-656
    bool CollisionCheckSegment(V2 start, V2 end)
    {
        if (!IsOnGrid(start.x, start.y) || !IsOnGrid(end.x, end.y))
            return false;
        uint32_t ks = GetNode3DIndexOnGridMap(start.x, start.y);
        uint32_t ke = GetNode3DIndexOnGridMap(end.x, end.y);
        auto it = in_range_obstacle_map_.find(ks);
        if (it != in_range_obstacle_map_.end())
            for (const Polygon &obst : it->second)
                if ((double)GetDistanceFromPolygonToSegment(obst, start, end) <= 0.5 * (double)P.vehicle_width)
                    return false;
        it = in_range_obstacle_map_.find(ke);
        if (it != in_range_obstacle_map_.end())
            for (const Polygon &obst : it->second)
                if ((double)GetDistanceFromPolygonToSegment(obst, start, end) <= 0.5 * (double)P.vehicle_width)
                    return false;
        return true;
    }
    // src/collisiondetection.cpp:189-214
    bool configurationTest(float x, float y, float t)
    {
        unsigned int sx, sy;
        if (!cm.worldToMap(x, y, sx, sy))
            return false;
        if (cm.cost[sx + sy * cm.w] >= 250)
            return false;
        Polygon polygon = CreatePolygon(V2{x, y}, P.vehicle_length, P.vehicle_width, t);
        return CollisionCheckPolygon(polygon);
    }
    // src/collisiondetection.cpp:93-127; has_pred=false is the pred==nullptr arm
    bool IsTraversable(float x, float y, float t, bool has_pred, float px, float py)
    {
        if (!IsOnGrid(x, y))
            return false;
        if (has_pred)
            return configurationTest(x, y, t) && CollisionCheckSegment(V2{x, y}, V2{px, py});
        return configurationTest(x, y, t);
    }

    // --------------------------------------------- collision (mode FOOTPRINT)
    // include/lookup.h:30-284 with Constants:: taken from the params.
    void buildFootprintTable()
    {
        const float cSize = P.fp_cell_size;
        const double width = P.fp_width, length = P.fp_length;
        const int size = (int)std::ceil((std::sqrt(width * width + length * length) + 4) / cSize);
        const int positionResolution = P.position_resolution;
        const int positions = positionResolution * positionResolution;
        const int headings = P.headings;
        const float deltaHeadingRad = (float)(2 * kPi / (double)(float)headings);
        fp_bb = size;
        fp_hcross1 = fp_hcross2 = 0;
        fp.assign((size_t)positions * headings, FpConfig{});
        struct pt
        {
            float x, y;
        };
        std::vector<pt> points(positions);
        for (int i = 0; i < positionResolution; ++i)
            for (int j = 0; j < positionResolution; ++j)
            {
                points[positionResolution * i + j].x = 1.f / positionResolution * j;
                points[positionResolution * i + j].y = 1.f / positionResolution * i;
            }
        std::vector<char> cSpace((size_t)size * size);
        auto sign = [](float v) { return v >= 0 ? 1 : -1; };
        for (int q = 0; q < positions; ++q)
        {
            float theta = 0;
            pt c, p[4], nP[4];
            c.x = (float)size / 2 + points[q].x;
            c.y = (float)size / 2 + points[q].y;
            p[0].x = (float)((double)c.x - length / 2 / (double)cSize);
            p[0].y = (float)((double)c.y - width / 2 / (double)cSize);
            p[1].x = (float)((double)c.x - length / 2 / (double)cSize);
            p[1].y = (float)((double)c.y + width / 2 / (double)cSize);
            p[2].x = (float)((double)c.x + length / 2 / (double)cSize);
            p[2].y = (float)((double)c.y + width / 2 / (double)cSize);
            p[3].x = (float)((double)c.x + length / 2 / (double)cSize);
            p[3].y = (float)((double)c.y - width / 2 / (double)cSize);
            for (int o = 0; o < headings; ++o)
            {
                std::fill(cSpace.begin(), cSpace.end(), 0);
                for (int j = 0; j < 4; ++j)
                {
                    pt temp{p[j].x - c.x, p[j].y - c.y};
                    nP[j].x = temp.x * m_cosf(theta) - temp.y * m_sinf(theta) + c.x;
                    nP[j].y = temp.x * m_sinf(theta) + temp.y * m_cosf(theta) + c.y;
                }
                theta += deltaHeadingRad;
                for (int k = 0; k < 4; ++k)
                {
                    pt start = nP[k], end = (k < 3) ? nP[k + 1] : nP[0];
                    int X = (int)start.x, Y = (int)start.y;
                    cSpace[Y * size + X] = 1;
                    pt t{end.x - start.x, end.y - start.y};
                    int stepX = sign(t.x), stepY = sign(t.y);
                    float tDeltaX = (t.x != 0) ? 1.f / m_fabsf(t.x) : 1000;
                    float tDeltaY = (t.y != 0) ? 1.f / m_fabsf(t.y) : 1000;
                    float tMaxX = (stepX > 0) ? tDeltaX * (1 - (start.x - (long)start.x)) : tDeltaX * (start.x - (long)start.x);
                    float tMaxY = (stepY > 0) ? tDeltaY * (1 - (start.y - (long)start.y)) : tDeltaY * (start.y - (long)start.y);
                    while ((int)end.x != X || (int)end.y != Y)
                    {
                        if (tMaxX < tMaxY && std::abs(X + stepX - (int)end.x) < std::abs(X - (int)end.x))
                        {
                            tMaxX = tMaxX + tDeltaX;
                            X = X + stepX;
                            cSpace[Y * size + X] = 1;
                        }
                        else if (tMaxY < tMaxX && std::abs(Y + stepY - (int)end.y) < std::abs(Y - (int)end.y))
                        {
                            tMaxY = tMaxY + tDeltaY;
                            Y = Y + stepY;
                            cSpace[Y * size + X] = 1;
                        }
                        else if (2 >= std::abs(X - (int)end.x) + std::abs(Y - (int)end.y))
                        {
                            if (std::abs(X - (int)end.x) > std::abs(Y - (int)end.y))
                                X = X + stepX;
                            else
                                Y = Y + stepY;
                            cSpace[Y * size + X] = 1;
                        }
                        else
                            break;
                    }
                }
                fillShape(cSpace, size);
                int count = 0;
                FpConfig &e = fp[(size_t)q * headings + o];
                for (int i = 0; i < size; ++i)
                    for (int j = 0; j < size; ++j)
                        if (cSpace[i * size + j])
                        {
                            if (count < 64)
                            {
                                e.pos[count][0] = j - (int)c.x;
                                e.pos[count][1] = i - (int)c.y;
                            }
                            count++;
                        }
                e.length = count;
            }
        }
    }
    // include/lookup.h:216-240.  hcross1/hcross2 persist across rows and
    // across table entries in the reference (function-scope variables).
    int fp_hcross1 = 0, fp_hcross2 = 0;
    void fillShape(std::vector<char> &cSpace, int size)
    {
        for (int i = 0; i < size; ++i)
        {
            bool inside = false;
            for (int j = 0; j < size; ++j)
            {
                for (int k = 0; k < size; ++k)
                {
                    if (cSpace[i * size + k] && !inside)
                    {
                        fp_hcross1 = k;
                        inside = true;
                    }
                    if (cSpace[i * size + k] && inside)
                        fp_hcross2 = k;
                }
                if (j > fp_hcross1 && j < fp_hcross2 && inside)
                    cSpace[i * size + j] = 1;
            }
        }
    }
    // Footprint-list pose test.  The reference tree keeps only the table
    // (include/collisiondetection.h:306); the test loop is the one the table
    // was designed for: X=(int)x, Y=(int)y, sub-cell (int)((x-X)*pr),
    // heading bin (int)(t/delta); every listed cell must be on the grid and
    // below the lethal threshold used by configurationTest (>= 250 collides).
    bool footprintTest(float x, float y, float t) const
    {
        const int pr = P.position_resolution;
        const float delta = (float)(2 * kPi / (double)(float)P.headings);
        int X = (int)x, Y = (int)y;
        int iX = (int)((x - (long)x) * pr);
        iX = iX > 0 ? iX : 0;
        int iY = (int)((y - (long)y) * pr);
        iY = iY > 0 ? iY : 0;
        int iT = (int)(t / delta);
        if (iT < 0 || iT >= P.headings || iX >= pr || iY >= pr)
            return false;
        int idx = iY * pr * P.headings + iX * P.headings + iT;
        const FpConfig &e = fp[idx];
        int n = e.length < 64 ? e.length : 64;
        for (int i = 0; i < n; ++i)
        {
            int cX = X + e.pos[i][0], cY = Y + e.pos[i][1];
            if (cX < 0 || (uint32_t)cX >= cm.w || cY < 0 || (uint32_t)cY >= cm.h)
                return false;
            if (cm.cost[(size_t)cY * cm.w + cX] >= 250)
                return false;
        }
        return true;
    }

    // ------------------------------------------------------------ LookupTable
    // src/lookup_table.cpp:123-146
    int CalculateNode3DIndex(float x, float y, float theta) const
    {
        const float delta_heading_in_rad = (float)(2 * kPi / (double)(float)P.headings);
        float angle = ((double)m_fabsf(theta) < 1e-4) ? 0 : theta;
        int x_index = (int)(x * P.position_resolution);
        int y_index = (int)(y * P.position_resolution);
        return (int)(angle / delta_heading_in_rad) * (int)cm.w * (int)cm.h * P.position_resolution * P.position_resolution +
               (int)(y_index * P.position_resolution) * (int)cm.w + (int)(x_index);
    }
    double curveDistance(int curve, float x, float y, float theta) const
    {
        // OMPL's {ReedsShepp,Dubins}StateSpace::distance (double state
        // (x, y, yaw) from float loop variables), src/lookup_table.cpp:172-174,
        // :212-214.  Reeds-Shepp: the in-tree ReedsShepp.cpp algorithm.
        // Dubins: OMPL's algorithm restated (ompl_dubins above), double as there.
        if (curve == HYAS_CURVE_RS)
        {
            double q0[3] = {0, 0, 0}, q1[3] = {x, y, theta};
            return (double)P.cd_min_turning_radius * reedsShepp(q0, q1, (double)P.cd_min_turning_radius).total;
        }
        if (curve == HYAS_CURVE_BEZIER) // CalculateCubicBezierLookupV1, src/lookup_table.cpp:226-277
            return bez::tableCost(x, y, theta);
        return ompl_dubins::distance(0., 0., 0., (double)x, (double)y, (double)theta, (double)P.cd_min_turning_radius);
    }
    // The float accumulators of src/lookup_table.cpp:158-183 / :197-222.
    void tableAxes(std::vector<float> &xs, std::vector<float> &ys, std::vector<float> &ts) const
    {
        const float delta_heading_in_rad = (float)(2 * kPi / (double)(float)P.headings);
        for (float x = 0; x < (float)(int)cm.w; x += 1.0f / P.position_resolution)
            xs.push_back(x);
        for (float y = 0; y < (float)(int)cm.h; y += 1.0f / P.position_resolution)
            ys.push_back(y);
        for (float th = 0; (double)th < 2 * kPi; th += delta_heading_in_rad)
            ts.push_back(th);
    }
    size_t tableSize() const
    {
        return (size_t)(P.headings + 1) * cm.w * cm.h * P.position_resolution * P.position_resolution + (size_t)cm.w * P.position_resolution;
    }
    // src/lookup_table.cpp:147-224 (x outer, y, theta inner; first emplace wins;
    // the x==0 && y==0 column is skipped).  Rows are solved on several threads
    // and merged in loop order, which keeps first-write-wins intact.
    void buildTable(int curve, int nthreads)
    {
        std::vector<float> xs, ys, ts;
        tableAxes(xs, ys, ts);
        const float nanv = std::numeric_limits<float>::quiet_NaN();
        table.assign(tableSize(), nanv);
        std::vector<float> raw(xs.size() * ys.size() * ts.size());
        auto work = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i)
                for (size_t j = 0; j < ys.size(); ++j)
                    for (size_t k = 0; k < ts.size(); ++k)
                        raw[(i * ys.size() + j) * ts.size() + k] = (float)curveDistance(curve, xs[i], ys[j], ts[k]);
        };
        std::vector<std::thread> th;
        if (nthreads < 1)
            nthreads = 1;
        size_t per = (xs.size() + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t)
        {
            size_t a = t * per, b = std::min(xs.size(), a + per);
            if (a < b)
                th.emplace_back(work, a, b);
        }
        for (auto &t : th)
            t.join();
        for (size_t i = 0; i < xs.size(); ++i)
            for (size_t j = 0; j < ys.size(); ++j)
            {
                if (xs[i] == 0 && ys[j] == 0)
                    continue;
                for (size_t k = 0; k < ts.size(); ++k)
                {
                    int index = CalculateNode3DIndex(xs[i], ys[j], ts[k]);
                    if (index >= 0 && (size_t)index < table.size() && table[index] != table[index])
                        table[index] = raw[(i * ys.size() + j) * ts.size() + k];
                }
            }
        table_curve = curve;
    }
    // src/lookup_table.cpp:56-96 (+ :113-122)
    float tableCost(float x, float y, float t) const
    {
        float miss = (table_curve == HYAS_CURVE_RS) ? 100000000.f : (table_curve == HYAS_CURVE_BEZIER) ? 10000000.f : 1000000.f; // :59, :85, :101
        int index = CalculateNode3DIndex(x, y, RadToZeroTo2P(t));
        if (index >= 0 && (size_t)index < table.size() && table[index] == table[index])
            return table[index];
        return miss;
    }

    size_t twodSize() const { return (size_t)cm.w * cm.h + cm.w + 1; }
    // ------------------------------------------------------ 2D cost-to-goal
    // src/hybrid_a_star.cpp:1020-1074 with AStar::CreateSuccessor(pred, 8)
    // (src/a_star.cpp:214-277, step = resolution_ :340-346) and
    // AStar::UpdateCostSoFar (:281-288).
    struct N2
    {
        float x, y, g;
        bool has_pred;
        float px, py;
    };
    struct PqCmp // include/hybrid_a_star.h:45-53
    {
        bool operator()(const std::pair<int, double> &l, const std::pair<int, double> &r) const { return l.second >= r.second; }
    };
    void buildAStarCostMap(float gx, float gy, std::vector<float> &cost_map /* -1 = absent */)
    {
        const int W = cm.w;
        const int NC = (int)twodSize(); // unordered_map in the reference: any key fits
        cost_map.assign((size_t)NC, -1.f);
        std::unordered_map<int, N2> open_set;
        std::priority_queue<std::pair<int, double>, std::vector<std::pair<int, double>>, PqCmp> open_pq;
        N2 goal{gx, gy, 0, false, 0, 0};
        int gid = setIdx2D(gx, gy, W, resolution_, origin_x_, origin_y_);
        if (gid < 0 || gid >= NC)
            return;
        open_set.emplace(gid, goal);
        open_pq.emplace(gid, (double)goal.g);
        const float step = resolution_;
        const float delta[3] = {-step, 0, step};
        while (!open_pq.empty())
        {
            int id = open_pq.top().first;
            open_pq.pop();
            N2 &cur = open_set[id];
            if (cur.has_pred)
                cur.g = cur.g + norm(V2{cur.x, cur.y} - V2{cur.px, cur.py});
            int cid = setIdx2D(cur.x, cur.y, W, resolution_, origin_x_, origin_y_);
            if (cid >= 0 && cid < NC && cost_map[cid] < 0)
                cost_map[cid] = cur.g;
            const N2 c = cur; // successors copy the updated node
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                {
                    if (delta[i] == 0 && delta[j] == 0)
                        continue;
                    float xs = c.x + delta[i], ys = c.y + delta[j];
                    if (!IsOnGrid(xs, ys))
                        continue;
                    int nid = setIdx2D(xs, ys, W, resolution_, origin_x_, origin_y_);
                    if (nid < 0 || nid >= NC)
                        continue;
                    if (cost_map[nid] >= 0)
                        continue;
                    auto it = open_set.find(nid);
                    if (it != open_set.end())
                    {
                        if (it->second.g > c.g)
                        {
                            it->second.g = c.g;
                            it->second.has_pred = true;
                            it->second.px = c.x;
                            it->second.py = c.y;
                        }
                    }
                    else
                    {
                        open_set.emplace(nid, N2{xs, ys, c.g, true, c.x, c.y});
                        open_pq.emplace(nid, (double)c.g);
                    }
                }
        }
    }

    // ------------------------------------------------------------ HybridAStar
    float GetDistance(const Node3D &a, const Node3D &b) const { return norm(V2{a.x, a.y} - V2{b.x, b.y}); }
    float headingDelta() const { return (float)((double)(float)2 * kPi / P.headings); }

    // src/hybrid_a_star.cpp:316-366
    float heuristic(const Node3D &n, const Node3D &goal, const std::vector<float> &twod) const
    {
        float gx = m_fabsf(goal.x - n.x), gy = m_fabsf(goal.y - n.y);
        float gt = RadToZeroTo2P(goal.t - n.t);
        float curve_cost = tableCost(gx, gy, gt);
        float twoD = 0;
        int id = setIdx2D(n.x, n.y, cm.w, resolution_, origin_x_, origin_y_);
        if (!twod.empty() && id >= 0 && (size_t)id < twod.size() && twod[id] >= 0)
            twoD = twod[id];
        return std::max(curve_cost, twoD);
    }

    // src/hybrid_a_star.cpp:540-657 (fixed step branch :600-621, reverse :629-654)
    std::vector<std::pair<float, float>> stepSteerPairs(const Node3D &pred, const Node3D &goal, const Node3D &start)
    {
        std::vector<std::pair<float, float>> v;
        float distance_to_goal = GetDistance(pred, goal);
        if (P.adaptive_steering_angle_and_step_size) // :547-584
        {
            float current_normalized_obstacle_density = GetNormalizedObstacleDensity(pred);
            if (current_normalized_obstacle_density > P.constant_density)
            {
                float first = (resolution_ > distance_to_goal) ? distance_to_goal : resolution_;
                for (float a : FormSteeringAngleVec(P.steering_angle, P.number_of_successors))
                    v.emplace_back(first, a);
            }
            else
                v = FindStepSizeAndSteeringAngle(pred, start, goal, P.number_of_successors, resolution_);
        }
        else if (P.adaptive_step_size) // :585-599
        {
            for (float a : FormSteeringAngleVec(P.steering_angle, P.number_of_successors))
                v.emplace_back(FindStepSize(pred, a, goal, resolution_), a);
        }
        else // :600-621
        {
            float step_size = 0.2;
            float first = (step_size > distance_to_goal) ? distance_to_goal : step_size;
            for (float a : FormSteeringAngleVec(P.steering_angle, P.number_of_successors))
                v.emplace_back(first, a);
        }
        if (P.reverse)
        {
            float rstep = -0.2;
            float rfirst = (m_fabsf(rstep) > distance_to_goal) ? -distance_to_goal : rstep;
            for (float a : FormSteeringAngleVec(P.steering_angle, 1))
                v.emplace_back(rfirst, a);
        }
        return v;
    }
    // src/hybrid_a_star.cpp:659-756; emits only traversable successors
    void createSuccessors(const Node3D &pred, int pred_id, const Node3D &goal, const Node3D &start, std::vector<Node3D> &out, int *n_checked)
    {
        out.clear();
        for (const auto &pair : stepSteerPairs(pred, goal, start))
        {
            bool reverse_flag = pair.first < 0;
            if ((double)m_fabsf(pair.first) <= 1e-3)
                continue;
            float steering_angle = pair.second;
            float turning_radius = pair.first / m_fabsf(steering_angle);
            float dx, dy;
            int prem;
            if (steering_angle < 0)
            {
                dx = turning_radius * m_sinf(m_fabsf(steering_angle));
                dy = -(turning_radius * (1 - m_cosf(steering_angle)));
                prem = reverse_flag ? 4 : 1;
            }
            else if ((double)steering_angle > 1e-3)
            {
                dx = turning_radius * m_sinf(m_fabsf(steering_angle));
                dy = (turning_radius * (1 - m_cosf(steering_angle)));
                prem = reverse_flag ? 5 : 2;
            }
            else
            {
                dx = pair.first;
                dy = 0;
                prem = reverse_flag ? 3 : 0;
            }
            Node3D s;
            s.x = pred.x + dx * m_cosf(pred.t) - dy * m_sinf(pred.t);
            s.y = pred.y + dx * m_sinf(pred.t) + dy * m_cosf(pred.t);
            s.t = RadToZeroTo2P(pred.t + steering_angle);
            s.g = pred.g;
            s.h = 0;
            s.reverse = reverse_flag;
            s.prim = prem;
            s.pred = pred_id;
            if (n_checked)
                ++*n_checked;
            if (IsTraversable(s.x, s.y, s.t, true, pred.x, pred.y))
                out.push_back(s);
        }
    }
    // src/hybrid_a_star.cpp:761-811
    float costSoFar(const Node3D &node, const Node3D &pred) const
    {
        float step_size = GetDistance(node, pred);
        float g = node.g;
        int prim = node.prim, pre_prim = pred.prim;
        const float wt = P.penalty_turning, wc = P.penalty_change_of_direction, wr = P.penalty_reverse;
        if (prim < 3)
        {
            if (pre_prim != prim)
                g += (pre_prim > 2) ? step_size * wt * wc : step_size * wt;
            else
                g += step_size;
        }
        else
        {
            if (pre_prim != prim)
                g += (pre_prim < 3) ? step_size * wt * wr * wc : step_size * wt * wr;
            else
                g += step_size * wr;
        }
        return g;
    }

    // src/hybrid_a_star.cpp:371-534.  Returns false when any sample collides.
    bool analyticExpansion(const Node3D &start, const Node3D &goal, std::vector<Node3D> *path_vec, float *length_out)
    {
        if (path_vec)
            path_vec->clear();
        float px = start.x, py = start.y;
        size_t n = 0;
        if (P.curve_type == HYAS_CURVE_DUBINS)
        {
            float q0[3] = {start.x, start.y, start.t}, q1[3] = {goal.x, goal.y, goal.t};
            db::Path path;
            path.param[0] = path.param[1] = path.param[2] = 0, path.type = 0, path.rho = P.min_turning_radius;
            for (int i = 0; i < 3; ++i)
                path.qi[i] = q0[i];
            db::init(q0, q1, P.min_turning_radius, &path);
            float length = db::length(&path);
            if (length_out)
                *length_out = length;
            float x = 0.f;
            while (x < length)
            {
                float q[3];
                db::sample(&path, x, q);
                Node3D nd;
                nd.x = q[0], nd.y = q[1], nd.t = RadToZeroTo2P(q[2]);
                if (!IsTraversable(nd.x, nd.y, nd.t, true, px, py))
                {
                    if (path_vec)
                        path_vec->clear();
                    return false;
                }
                if (path_vec)
                    path_vec->push_back(nd);
                x = (float)((double)x + 0.2 * (double)resolution_);
                px = nd.x, py = nd.y;
                ++n;
            }
        }
        else if (P.curve_type == HYAS_CURVE_BEZIER) // src/hybrid_a_star.cpp:470-521
        {
            const float vs[3] = {start.x, start.y, start.t}, vg[3] = {goal.x, goal.y, goal.t};
            bez::Cubic cubic(vs, vg);
            float length = cubic.GetLength();
            if (length_out)
                *length_out = length;
            float x = 0;
            while (x < length)
            {
                float v[3];
                cubic.GetValueAt(x / length, v);
                Node3D nd;
                nd.x = v[0], nd.y = v[1];
                float curvature = cubic.GetCurvatureAt(x / length);
                nd.t = RadToZeroTo2P(cubic.GetAngleAt(x / length));
                if (!IsTraversable(nd.x, nd.y, nd.t, true, px, py) || !(curvature <= 1 / P.min_turning_radius))
                {
                    if (path_vec)
                        path_vec->clear();
                    return false;
                }
                if (path_vec)
                    path_vec->push_back(nd);
                x = (float)((double)x + 0.1 * (double)resolution_);
                px = nd.x, py = nd.y;
                ++n;
            }
        }
        else
        {
            double q0[3] = {start.x, start.y, start.t}, q1[3] = {goal.x, goal.y, goal.t};
            double length = -1;
            std::vector<std::array<double, 3>> pts;
            rsSample(q0, q1, (double)P.min_turning_radius, 0.1 * (double)resolution_, length, pts);
            if (length_out)
                *length_out = (float)length;
            for (auto &p : pts)
            {
                Node3D nd;
                nd.x = (float)p[0], nd.y = (float)p[1], nd.t = RadToZeroTo2P((float)p[2]);
                if (!IsTraversable(nd.x, nd.y, nd.t, true, px, py))
                {
                    if (path_vec)
                        path_vec->clear();
                    return false;
                }
                if (path_vec)
                    path_vec->push_back(nd);
                px = nd.x, py = nd.y;
                ++n;
            }
        }
        if (n == 0)
            return false; // empty curve: path_vec.size()==0 in the reference
        if (path_vec)
            path_vec->push_back(goal);
        return true;
    }

    // src/collisiondetection.cpp:129-145: both poses and the segment between them
    bool IsTraversablePair(const Node3D &current, const Node3D &previous)
    {
        if (!IsOnGrid(current.x, current.y) || !IsOnGrid(previous.x, previous.y))
            return false;
        return configurationTest(current.x, current.y, current.t) && CollisionCheckSegment(V2{current.x, current.y}, V2{previous.x, previous.y}) &&
               configurationTest(previous.x, previous.y, previous.t);
    }
    // src/utility.cpp:1288-1296 and :2111-2117
    static float GetAngle(const Node3D &start, const Node3D &goal) { return RadToZeroTo2P(m_atan2f(goal.y - start.y, goal.x - start.x)); }
    static float FindSteeringAngle(const Node3D &closest, const Node3D &direction)
    {
        float angle_between = RadNormalization(GetAngle(closest, direction));
        return -RadNormalization(RadNormalization(closest.t) - angle_between);
    }
    // HybridAStar::ShortCut(path, true), src/hybrid_a_star.cpp:1076-1179, with its unsigned loop counters.  out_idx
    // (optional) receives the input indices of the returned nodes.  The steering-angle limit it compares with is
    // max_steering_angle = 30, a number of degrees, against an angle in radians: the test passes for every finite angle.
    // The reference indexes input[size - 2] first, so a path of fewer than two nodes is undefined there; such a path
    // is returned unchanged here.
    std::vector<Node3D> ShortCut(const std::vector<Node3D> &path, std::vector<uint32_t> *out_idx = nullptr)
    {
        std::vector<Node3D> out, input = path;
        std::vector<uint32_t> idx;
        if (input.size() < 2)
        {
            if (out_idx)
                for (uint32_t i = 0; i < input.size(); ++i)
                    out_idx->push_back(i);
            return input;
        }
        Node3D current_point, previous_point;
        uint32_t previous_idx = (uint32_t)input.size() - 2;
        float steering_angle;
        previous_point = input[input.size() - 2];
        for (uint32_t index1 = (uint32_t)input.size() - 1; /* index1 >= 0 */;)
        {
            current_point = input[index1];
            out.emplace_back(current_point);
            idx.push_back(index1);
            for (uint32_t index2 = index1 - 1; /* index2 >= 0 */; index2--)
            {
                if (IsTraversablePair(current_point, input[index2]))
                {
                    previous_point = input[index2];
                    previous_idx = index2;
                    steering_angle = FindSteeringAngle(previous_point, current_point);
                    if (std::abs(steering_angle) < P.max_steering_angle)
                    {
                        if (index2 == 0)
                        {
                            index1 = 0;
                            out.emplace_back(previous_point);
                            idx.push_back(previous_idx);
                            break;
                        }
                        continue;
                    }
                    else
                    {
                        out.emplace_back(previous_point);
                        idx.push_back(previous_idx);
                        index1 = (index2 == 0) ? 0 : index2 - 1;
                        break;
                    }
                }
                else
                {
                    out.emplace_back(previous_point);
                    idx.push_back(previous_idx);
                    index1 = (index2 == 0) ? 0 : index2 - 1;
                    break;
                }
            }
            if (index1 == 0)
                break;
        }
        std::reverse(out.begin(), out.end());
        std::reverse(idx.begin(), idx.end());
        if (out_idx)
            *out_idx = idx;
        return out;
    }

    // include/node3d.h:89-98
    static bool nodeEq(const Node3D &a, const Node3D &b)
    {
        return (std::abs(a.x - b.x) < 1e-2) && (std::abs(a.y - b.y) < 1e-2) && (std::abs(a.t - b.t) < 1e-2);
    }
    // CubicBezier::ConvertCubicBezierToVector3f (src/cubic_bezier.cpp:179-194) appended to out as nodes
    // (Utility::TypeConversion(Vector3f, Node3D), src/utility.cpp:128-133)
    static void appendBezierNodes(const bez::Cubic &c, unsigned number_of_points, std::vector<Node3D> &out)
    {
        for (unsigned i = 0; i < number_of_points + 1; ++i)
        {
            const float t = (float)i / number_of_points;
            float v[3];
            c.GetValueAt(t, v);
            Node3D nd;
            nd.x = v[0], nd.y = v[1], nd.t = RadToZeroTo2P(RadToZeroTo2P(c.GetAngleAt(t)));
            out.push_back(nd);
        }
    }
    // HybridAStar::ConvertToPiecewiseCubicBezierPath (src/hybrid_a_star.cpp:830-883) with PiecewiseCubicBezier::
    // SetAnchorPoints / CalculateCubicBezier / ConvertPiecewiseCubicBezierToVector3f (include/piecewise_cubic_bezier.h:51-60,
    // src/piecewise_cubic_bezier.cpp:221-234, :336-357): one cubic Bezier curve between consecutive anchor points, the
    // traced nodes at least one cell apart up to the node the analytic curve starts from.  CalculateCubicBezier counts
    // to size() - 1 in unsigned arithmetic: an empty anchor list is undefined there and gives no curve here.
    std::vector<Node3D> ConvertToPiecewiseCubicBezierPath(const std::vector<Node3D> &path_, size_t analytical_expansion_index_, const Node3D &start_) const
    {
        std::vector<Node3D> anchor_points_vec, out;
        Node3D end_point;
        end_point.x = 0, end_point.y = 0, end_point.t = 0;
        for (unsigned index = 0; index < path_.size() - 1; ++index)
        {
            if (index == analytical_expansion_index_)
            {
                end_point.x = path_[index].x, end_point.y = path_[index].y, end_point.t = RadToZeroTo2P(path_[index].t);
                anchor_points_vec.push_back(path_[index]);
                break;
            }
            if (GetDistance(path_[index], path_[index + 1]) < resolution_)
                continue;
            anchor_points_vec.push_back(path_[index]);
        }
        const int number_of_points = (int)m_roundf(GetDistance(start_, end_point) / resolution_);
        for (size_t i = 0; i + 1 < anchor_points_vec.size(); ++i)
        {
            const float a[3] = {anchor_points_vec[i].x, anchor_points_vec[i].y, anchor_points_vec[i].t};
            const float b[3] = {anchor_points_vec[i + 1].x, anchor_points_vec[i + 1].y, anchor_points_vec[i + 1].t};
            appendBezierNodes(bez::Cubic(a, b), (unsigned)number_of_points, out);
        }
        return out;
    }
    // src/utility.cpp:2091-2109
    void RemoveDuplicates(std::vector<Node3D> &path) const
    {
        if (path.empty())
            return;
        for (uint32_t index = 0; index < path.size() - 1; index++)
            if ((double)GetDistance(path[index], path[index + 1]) < 1e-3)
            {
                path.erase(path.begin() + index);
                index = index - 1;
            }
    }

    struct HeapItem
    {
        float f;
        uint32_t seq;
        int node;
    };
    struct HeapCmp
    { // lowest f first, then earliest push
        bool operator()(const HeapItem &a, const HeapItem &b) const { return a.f > b.f || (a.f == b.f && a.seq > b.seq); }
    };

    // src/hybrid_a_star.cpp:982-1018 + GetPath :41-311
    void plan(const float s[3], const float g[3], hyas_result *res) { planInto(s, g, res, last_path, last_expanded); }
    // The same search with its outputs in caller-owned vectors: touches no member, so several threads may plan on
    // one context (the map and the table are only read).
    void planInto(const float s[3], const float g[3], hyas_result *res, std::vector<float> &last_path, std::vector<int32_t> &last_expanded)
    {
        std::memset(res, 0, sizeof(*res));
        last_path.clear();
        last_expanded.clear();
        const int W = cm.w, H = cm.h;
        // src/utility.cpp:7-13: yaw normalised into [0, 2pi)
        Node3D start, goal;
        start.x = s[0], start.y = s[1], start.t = RadToZeroTo2P(s[2]);
        goal.x = g[0], goal.y = g[1], goal.t = RadToZeroTo2P(g[2]);
        // makePlan pre-checks, src/hybrid_a_star_planner.cpp:126-155
        if (!IsTraversable(start.x, start.y, start.t, false, 0, 0))
        {
            res->status = HYAS_STATUS_BAD_POSE;
            return;
        }
        {
            unsigned int gx, gy;
            if (!cm.worldToMap(goal.x, goal.y, gx, gy))
            {
                res->status = HYAS_STATUS_BAD_POSE;
                return;
            }
            if (IsTraversable(goal.x, goal.y, goal.t, false, 0, 0) && cm.cost[gx + gy * W] > 252)
            {
                res->status = HYAS_STATUS_BAD_POSE;
                return;
            }
        }
        struct Cell
        {
            float g = 999;
            bool o = false, c = false;
        };
        std::vector<Cell> nodes3D((size_t)W * H * P.headings + (size_t)H * W + W);
        std::vector<Node3D> arena;
        arena.reserve(4096);
        std::vector<float> twod;
        buildAStarCostMap(goal.x, goal.y, twod);
        std::vector<HeapItem> heap;
        HeapCmp cmp;
        uint32_t seq = 0;
        const float dh = headingDelta();
        start.h = heuristic(start, goal, twod);
        int N = (int)start.h;
        start.o = true;
        arena.push_back(start);
        heap.push_back(HeapItem{start.g + start.h, seq++, 0});
        int iPred = setIdx3D(start, W, H, dh, resolution_, origin_x_, origin_y_);
        nodes3D[iPred].g = start.g, nodes3D[iPred].o = true, nodes3D[iPred].c = false;
        int iterations = 0, counter = 0;
        std::vector<Node3D> succ, shot, path;
        auto trace = [&](int id) {
            path.clear();
            while (id >= 0)
            {
                path.push_back(arena[id]);
                if (nodeEq(arena[id], start))
                    break;
                id = arena[id].pred;
            }
            std::reverse(path.begin(), path.end());
        };
        // GetPath's endings (:118-147): trace, RemoveDuplicates, ShortCut; calculatePath removes duplicates once more (:994)
        auto shortcut = [&]() {
            if (P.short_cut)
                path = ShortCut(path);
        };
        auto finish = [&](int status, int id) {
            res->status = status;
            res->g_final = arena[id].g;
            if (status != HYAS_STATUS_SHOT)
            {
                RemoveDuplicates(path);
                shortcut();
            }
            else
                RemoveDuplicates(path); // :172, after the curve was appended
            RemoveDuplicates(path); // calculatePath repeats it (:994)
            res->path_len = (int)path.size();
            double len = 0;
            for (size_t i = 0; i < path.size(); ++i)
            {
                last_path.push_back(path[i].x), last_path.push_back(path[i].y), last_path.push_back(path[i].t);
                if (i)
                    len += (double)GetDistance(path[i], path[i - 1]);
            }
            res->path_length = (float)len;
        };
        while (!heap.empty())
        {
            HeapItem top = heap.front();
            int pid = top.node;
            res->nodes_explored++;
            iPred = setIdx3D(arena[pid], W, H, dh, resolution_, origin_x_, origin_y_);
            iterations++;
            if (nodes3D[iPred].c)
            {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                heap.pop_back();
                continue;
            }
            if (!nodes3D[iPred].o)
                break; // unreachable: a pushed cell is open or closed
            nodes3D[iPred].c = true, nodes3D[iPred].o = false;
            std::pop_heap(heap.begin(), heap.end(), cmp);
            heap.pop_back();
            const Node3D nPred = arena[pid];
            if (iterations > P.max_iterations)
            {
                trace(pid);
                finish(HYAS_STATUS_MAX_ITER, pid);
                return;
            }
            // src/utility.cpp:1132-1163
            {
                float distance = GetDistance(nPred, goal);
                float distance_range = P.goal_range * resolution_;
                float angle_range = (float)(2 * kPi / P.headings);
                if (distance < distance_range && m_fabsf(RadNormalization(nPred.t - goal.t)) <= angle_range)
                {
                    trace(pid);
                    finish(HYAS_STATUS_GOAL, pid);
                    return;
                }
            }
            res->nodes_expanded++;
            last_expanded.push_back(iPred);
            if (P.analytical_expansion)
            {
                bool shoot = false;
                if (P.analytical_expansion_every_point)
                    shoot = true;
                else if (counter == N)
                {
                    N = (int)nPred.h;
                    counter = 0;
                    shoot = true;
                }
                else
                    counter++;
                float slen = 0;
                if (shoot && analyticExpansion(nPred, goal, &shot, &slen))
                {
                    trace(pid);
                    shortcut(); // :165-168, before the curve is appended
                    const size_t analytical_expansion_index = path.size(); // :169, :212
                    path.insert(path.end(), shot.begin(), shot.end());
                    res->shot_length = slen;
                    if (P.piecewise_cubic_bezier_interpolation) // :178-189, :218-229
                    {
                        if (P.analytical_expansion_every_point)
                            RemoveDuplicates(path); // :172 comes before the conversion, :230 after the return
                        std::vector<Node3D> pw = ConvertToPiecewiseCubicBezierPath(path, analytical_expansion_index, start);
                        pw.insert(pw.end(), shot.begin(), shot.end());
                        path.swap(pw); // RemoveDuplicates of :184 / :224 and of :994 follow in finish()
                    }
                    finish(HYAS_STATUS_SHOT, pid);
                    return;
                }
            }
            createSuccessors(nPred, pid, goal, start, succ, &res->successors);
            for (Node3D &nSucc : succ)
            {
                // DuplicateCheck, src/hybrid_a_star.cpp:968-980
                bool dup = false;
                for (const HeapItem &e : heap)
                {
                    const Node3D &q = arena[e.node];
                    if ((double)m_fabsf(q.x - nSucc.x) < 0.05 && (double)m_fabsf(q.y - nSucc.y) < 0.05 &&
                        m_fabsf(q.t - nSucc.t) < ConvertDegToRad(3))
                    {
                        dup = true;
                        break;
                    }
                }
                if (dup)
                    continue;
                int iSucc = setIdx3D(nSucc, W, H, dh, resolution_, origin_x_, origin_y_);
                // second IsTraversable of the reference (:260): same verdict
                if (!IsTraversable(nSucc.x, nSucc.y, nSucc.t, true, nPred.x, nPred.y))
                    continue;
                if (nodes3D[iSucc].c)
                    continue;
                nSucc.g = costSoFar(nSucc, nPred);
                float newG = nSucc.g;
                if (!nodes3D[iSucc].o || newG < nodes3D[iSucc].g)
                {
                    nSucc.h = heuristic(nSucc, goal, twod);
                    if (iPred == iSucc)
                    {
                        if (nSucc.g + nSucc.h > nPred.g + nPred.h + P.tie_breaker)
                            continue;
                        nSucc.pred = nPred.pred;
                    }
                    nSucc.o = true, nSucc.c = false;
                    nodes3D[iSucc].g = nSucc.g, nodes3D[iSucc].o = true, nodes3D[iSucc].c = false;
                    arena.push_back(nSucc);
                    heap.push_back(HeapItem{nSucc.g + nSucc.h, seq++, (int)arena.size() - 1});
                    std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
        }
        res->status = HYAS_STATUS_NO_PATH;
    }
};

} // namespace orc

// ======================================================================= C API
using orc::Ctx;
extern "C" {

void *orc_create(const hyas_params *p)
{
    Ctx *c = new Ctx();
    c->P = *p;
    return c;
}
void orc_destroy(void *h) { delete (Ctx *)h; }

int orc_set_costmap(void *h, const uint8_t *charmap, uint32_t w, uint32_t hgt, double res, double ox, double oy,
                    int build_table, int nthreads)
{
    Ctx *c = (Ctx *)h;
    c->cm.w = w, c->cm.h = hgt, c->cm.res = res, c->cm.ox = ox, c->cm.oy = oy;
    c->cm.cost.assign(charmap, charmap + (size_t)w * hgt);
    c->resolution_ = (float)res, c->origin_x_ = (float)ox, c->origin_y_ = (float)oy;
    c->translateCostmap();
    c->SetObstacleVec();
    c->SetInRangeObstacle();
    if (c->P.adaptive_step_size || c->P.adaptive_steering_angle_and_step_size)
        c->buildAdaptiveMaps();
    c->buildFootprintTable();
    if (build_table)
        c->buildTable(c->P.curve_type, nthreads);
    return 0;
}

// Grid for the footprint-list test only (any size): no obstacle polygons, no lists, no tables.
int orc_set_grid(void *h, const uint8_t *charmap, uint32_t w, uint32_t hgt)
{
    Ctx *c = (Ctx *)h;
    c->cm.w = w, c->cm.h = hgt, c->cm.res = 1, c->cm.ox = 0, c->cm.oy = 0;
    c->cm.cost.assign(charmap, charmap + (size_t)w * hgt);
    c->resolution_ = 1, c->origin_x_ = 0, c->origin_y_ = 0;
    c->buildFootprintTable();
    return 0;
}

int orc_num_obstacles(void *h) { return (int)((Ctx *)h)->obstacle_vec_.size(); }
// 8 floats per obstacle: the 4 distinct vertices
int orc_get_obstacles(void *h, float *out, size_t cap)
{
    Ctx *c = (Ctx *)h;
    size_t n = std::min(cap, c->obstacle_vec_.size());
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k)
            out[i * 8 + 2 * k] = c->obstacle_vec_[i][k].x, out[i * 8 + 2 * k + 1] = c->obstacle_vec_[i][k].y;
    return (int)c->obstacle_vec_.size();
}
// number of in-range obstacles of grid key `key`, -1 if the key is absent
int orc_in_range_count(void *h, uint32_t key)
{
    Ctx *c = (Ctx *)h;
    auto it = c->in_range_obstacle_map_.find(key);
    return it == c->in_range_obstacle_map_.end() ? -1 : (int)it->second.size();
}

// the list of grid key `key` in list order, 8 floats per obstacle; returns its length, -1 if the key is absent
int orc_in_range_list(void *h, uint32_t key, float *out, size_t cap)
{
    Ctx *c = (Ctx *)h;
    auto it = c->in_range_obstacle_map_.find(key);
    if (it == c->in_range_obstacle_map_.end())
        return -1;
    for (size_t i = 0; i < it->second.size() && i < cap; ++i)
        for (int k = 0; k < 4; ++k)
            out[i * 8 + 2 * k] = it->second[i][k].x, out[i * 8 + 2 * k + 1] = it->second[i][k].y;
    return (int)it->second.size();
}

int orc_collision_batch(void *h, int mode, const float *xyt, const float *pred_xy, size_t n, uint8_t *free_out)
{
    Ctx *c = (Ctx *)h;
    for (size_t i = 0; i < n; ++i)
    {
        float x = xyt[3 * i], y = xyt[3 * i + 1], t = xyt[3 * i + 2];
        if (mode == HYAS_MODE_FOOTPRINT)
            free_out[i] = c->footprintTest(x, y, t);
        else
            free_out[i] = c->IsTraversable(x, y, t, pred_xy != nullptr, pred_xy ? pred_xy[2 * i] : 0, pred_xy ? pred_xy[2 * i + 1] : 0);
    }
    return 0;
}

int orc_footprint_table(void *h, int32_t *out, size_t cap)
{
    Ctx *c = (Ctx *)h;
    size_t n = std::min(cap, c->fp.size());
    std::memcpy(out, c->fp.data(), n * sizeof(orc::FpConfig));
    return (int)c->fp.size();
}

int orc_num_successors(void *h)
{
    Ctx *c = (Ctx *)h;
    if (c->P.adaptive_steering_angle_and_step_size)
        return HYAS_ADAPTIVE_SLOTS;
    return (int)orc::FormSteeringAngleVec(c->P.steering_angle, c->P.number_of_successors).size() + (c->P.reverse ? 1 : 0);
}
// adaptive maps of one grid cell (key = GetNode3DIndexOnGridMap): min distance, normalised density and up to cap
// {distance, start, range} triples of distance_angle_range_map_; returns the number of ranges, -1 for an unknown key
int orc_adaptive_cell(void *h, uint32_t key, float *min_distance, float *density, float *ranges3, int cap, int cut)
{
    Ctx *c = (Ctx *)h;
    auto it = c->distance_angle_range_map_.find(key);
    if (it == c->distance_angle_range_map_.end())
        return -1;
    std::vector<std::pair<float, orc::AngleRange>> list = it->second;
    if (cut) // FindFreeAngleRangeAndObstacleAngleRange of a node in this cell: the cell's corner stands in for it
    {
        orc::Node3D nd;
        nd.x = c->origin_x_ + (float)(key % c->cm.w) * c->resolution_, nd.y = c->origin_y_ + (float)(key / c->cm.w) * c->resolution_;
        if (c->GetNode3DIndexOnGridMap(nd.x, nd.y) != key)
            return -2;
        list = c->FindFreeAngleRangeAndObstacleAngleRange(nd);
    }
    *min_distance = c->min_distance_map_.count(key) ? c->min_distance_map_[key] : -1.f;
    *density = c->normalized_obstacle_density_map_.count(key) ? c->normalized_obstacle_density_map_[key] : -1.f;
    int k = 0;
    for (const auto &e : list)
    {
        if (k < cap)
            ranges3[3 * k] = e.first, ranges3[3 * k + 1] = e.second.first, ranges3[3 * k + 2] = e.second.second;
        ++k;
    }
    return k;
}
// the {step size, steering angle} pairs CreateSuccessor would try for one node, reverse pair included
int orc_step_steer_pairs(void *h, const hyas_node *pred, const hyas_goal *goal, const hyas_goal *start, float *pairs2, int cap)
{
    Ctx *c = (Ctx *)h;
    orc::Node3D p, g, st;
    p.x = pred->x, p.y = pred->y, p.t = pred->t;
    g.x = goal->x, g.y = goal->y, g.t = goal->t;
    st.x = start->x, st.y = start->y, st.t = start->t;
    auto v = c->stepSteerPairs(p, g, st);
    for (int k = 0; k < (int)v.size() && k < cap; ++k)
        pairs2[2 * k] = v[k].first, pairs2[2 * k + 1] = v[k].second;
    return (int)v.size();
}

int orc_expand_batch_from(void *h, const hyas_node *preds, const hyas_goal *goals, const hyas_goal *starts, const float *twod, size_t n,
                          hyas_succ *out, uint8_t *count);
int orc_expand_batch(void *h, const hyas_node *preds, const hyas_goal *goals, const float *twod, size_t n,
                     hyas_succ *out, uint8_t *count)
{
    return orc_expand_batch_from(h, preds, goals, nullptr, twod, n, out, count);
}
// starts[i] (optional): the start pose of the search pair i belongs to; the adaptive steering selection weighs with the
// start-goal distance (src/collisiondetection.cpp:1389).  Without it the goal stands in (distance 0).
int orc_expand_batch_from(void *h, const hyas_node *preds, const hyas_goal *goals, const hyas_goal *starts, const float *twod, size_t n,
                          hyas_succ *out, uint8_t *count)
{
    Ctx *c = (Ctx *)h;
    const int K = orc_num_successors(h);
    const size_t cells = c->twodSize();
    std::vector<orc::Node3D> succ;
    std::vector<float> td;
    for (size_t i = 0; i < n; ++i)
    {
        orc::Node3D p, g;
        p.x = preds[i].x, p.y = preds[i].y, p.t = preds[i].t, p.g = preds[i].g, p.prim = preds[i].prim;
        g.x = goals[i].x, g.y = goals[i].y, g.t = goals[i].t;
        orc::Node3D st = g;
        if (starts)
            st.x = starts[i].x, st.y = starts[i].y, st.t = starts[i].t;
        if (twod)
            td.assign(twod + i * cells, twod + (i + 1) * cells);
        c->createSuccessors(p, -1, g, st, succ, nullptr);
        count[i] = (uint8_t)succ.size();
        for (int k = 0; k < K; ++k)
        {
            hyas_succ &o = out[i * K + k];
            std::memset(&o, 0, sizeof(o));
            if ((size_t)k >= succ.size())
                continue;
            orc::Node3D s = succ[k];
            s.g = c->costSoFar(s, p);
            s.h = c->heuristic(s, g, td);
            o.x = s.x, o.y = s.y, o.t = s.t, o.g = s.g, o.h = s.h, o.prim = (int8_t)s.prim, o.free = 1;
            o.idx = orc::setIdx3D(s, c->cm.w, c->cm.h, c->headingDelta(), c->resolution_, c->origin_x_, c->origin_y_);
        }
    }
    return 0;
}

int orc_shot_batch(void *h, const hyas_node *nodes, const hyas_goal *goals, size_t n, uint8_t *ok, float *len)
{
    Ctx *c = (Ctx *)h;
    for (size_t i = 0; i < n; ++i)
    {
        orc::Node3D p, g;
        p.x = nodes[i].x, p.y = nodes[i].y, p.t = nodes[i].t;
        g.x = goals[i].x, g.y = goals[i].y, g.t = goals[i].t;
        float l = 0;
        ok[i] = c->analyticExpansion(p, g, nullptr, &l);
        len[i] = l;
    }
    return 0;
}

int orc_twod_size(void *h) { return (int)((Ctx *)h)->twodSize(); }
int orc_flood(void *h, const hyas_goal *goal, float *out)
{
    Ctx *c = (Ctx *)h;
    std::vector<float> m;
    c->buildAStarCostMap(goal->x, goal->y, m);
    std::memcpy(out, m.data(), m.size() * sizeof(float));
    return 0;
}

int orc_table_dims(void *h, int32_t *nx, int32_t *ny, int32_t *ns)
{
    Ctx *c = (Ctx *)h;
    std::vector<float> xs, ys, ts;
    c->tableAxes(xs, ys, ts);
    *nx = (int)xs.size(), *ny = (int)ys.size(), *ns = (int)ts.size();
    return 0;
}
// raw costs in [slice][iy][ix] order for heading-loop iterations [b, e)
int orc_table_build(void *h, int curve, int b, int e, float *out)
{
    Ctx *c = (Ctx *)h;
    std::vector<float> xs, ys, ts;
    c->tableAxes(xs, ys, ts);
    for (int k = b; k < e; ++k)
        for (size_t j = 0; j < ys.size(); ++j)
            for (size_t i = 0; i < xs.size(); ++i)
                out[((size_t)(k - b) * ys.size() + j) * xs.size() + i] = (float)c->curveDistance(curve, xs[i], ys[j], ts[k]);
    return 0;
}
int orc_build_table(void *h, int curve, int nthreads)
{
    ((Ctx *)h)->buildTable(curve, nthreads);
    return 0;
}
int orc_table_lookup(void *h, const float *xyt, size_t n, float *cost)
{
    Ctx *c = (Ctx *)h;
    for (size_t i = 0; i < n; ++i)
        cost[i] = c->tableCost(xyt[3 * i], xyt[3 * i + 1], xyt[3 * i + 2]);
    return 0;
}
// sum over emplaced keys (double) and their count, for checksum pins
double orc_table_checksum(void *h, int64_t *count)
{
    Ctx *c = (Ctx *)h;
    double s = 0;
    int64_t n = 0;
    for (float v : c->table)
        if (v == v)
            s += (double)v, ++n;
    if (count)
        *count = n;
    return s;
}

int orc_plan(void *h, const float *start, const float *goal, hyas_result *res)
{
    ((Ctx *)h)->plan(start, goal, res);
    return 0;
}
// Thread-safe form of orc_plan + orc_get_path + orc_get_expanded: outputs go to the caller's buffers
// (path_n / exp_n receive the full lengths; at most the caps are copied).
int orc_plan_r(void *h, const float *start, const float *goal, hyas_result *res, float *path_xyt, size_t path_cap, int32_t *path_n,
               int32_t *exp_idx, size_t exp_cap, int32_t *exp_n)
{
    std::vector<float> path;
    std::vector<int32_t> expanded;
    ((Ctx *)h)->planInto(start, goal, res, path, expanded);
    *path_n = (int32_t)(path.size() / 3), *exp_n = (int32_t)expanded.size();
    if (path_xyt)
        std::memcpy(path_xyt, path.data(), std::min(path_cap * 3, path.size()) * sizeof(float));
    if (exp_idx)
        std::memcpy(exp_idx, expanded.data(), std::min(exp_cap, expanded.size()) * sizeof(int32_t));
    return 0;
}

// HybridAStar::ShortCut on one path of n poses {x, y, t}; out_idx receives the input indices of the returned nodes
// (at most n + 1 of them); returns their number.
int orc_shortcut(void *h, const float *xyt, uint32_t n, uint32_t *out_idx)
{
    std::vector<orc::Node3D> path(n);
    for (uint32_t i = 0; i < n; ++i)
        path[i].x = xyt[3 * i], path[i].y = xyt[3 * i + 1], path[i].t = xyt[3 * i + 2];
    std::vector<uint32_t> idx;
    ((Ctx *)h)->ShortCut(path, &idx);
    for (size_t i = 0; i < idx.size(); ++i)
        out_idx[i] = idx[i];
    return (int)idx.size();
}

int orc_get_path(void *h, float *xyt, size_t cap)
{
    Ctx *c = (Ctx *)h;
    size_t n = c->last_path.size() / 3;
    std::memcpy(xyt, c->last_path.data(), std::min(cap, n) * 3 * sizeof(float));
    return (int)n;
}
int orc_get_expanded(void *h, int32_t *idx, size_t cap)
{
    Ctx *c = (Ctx *)h;
    size_t n = c->last_expanded.size();
    std::memcpy(idx, c->last_expanded.data(), std::min(cap, n) * sizeof(int32_t));
    return (int)n;
}

// ---- raw function access for known-answer tests --------------------------
double orc_rs_distance(double rho, const double *q0, const double *q1) { return rho * orc::reedsShepp(q0, q1, rho).total; }
// cubic Bezier: the table cost of one entry; and of one shot curve its length and at parameter t the value, heading and curvature
float orc_bezier_table_cost(float x, float y, float theta) { return orc::bez::tableCost(x, y, theta); }
float orc_bezier_eval(const float *start, const float *goal, float t, float *out5)
{
    orc::bez::Cubic c(start, goal);
    float v[3];
    c.GetValueAt(t, v);
    out5[0] = v[0], out5[1] = v[1], out5[2] = v[2], out5[3] = c.GetAngleAt(t), out5[4] = c.GetCurvatureAt(t);
    return c.GetLength();
}
int orc_rs_path(double rho, const double *q0, const double *q1, int *type, double *len5)
{
    orc::RsPath p = orc::reedsShepp(q0, q1, rho);
    *type = p.type;
    for (int i = 0; i < 5; ++i)
        len5[i] = p.len[i];
    return 0;
}
int orc_rs_sample(double rho, const double *q0, const double *q1, double step, double *length, double *pts, size_t cap)
{
    std::vector<std::array<double, 3>> v;
    orc::rsSample(q0, q1, rho, step, *length, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i)
        pts[3 * i] = v[i][0], pts[3 * i + 1] = v[i][1], pts[3 * i + 2] = v[i][2];
    return (int)v.size();
}
// out: [type, p0, p1, p2, length] as floats; returns dubins_init's code
int orc_dubins(const float *q0, const float *q1, float rho, float *out)
{
    orc::db::Path p;
    p.param[0] = p.param[1] = p.param[2] = 0, p.type = -1;
    int rc = orc::db::init(q0, q1, rho, &p);
    out[0] = (float)p.type, out[1] = p.param[0], out[2] = p.param[1], out[3] = p.param[2], out[4] = orc::db::length(&p);
    return rc;
}
int orc_dubins_sample(const float *q0, const float *q1, float rho, float t, float *q)
{
    orc::db::Path p;
    if (orc::db::init(q0, q1, rho, &p) != 0)
        return -1;
    return orc::db::sample(&p, t, q);
}

// Utility:: known-answer access; op codes documented in tests/test_oracle_kat.py
double orc_utility(int op, const float *a)
{
    using namespace orc;
    auto P = [&](int i) { return V2{a[2 * i], a[2 * i + 1]}; };
    switch (op)
    {
    case 0:
        return IsIntersect(P(0), P(1), P(2), P(3));
    case 1:
    {
        V2 r = FindIntersectionPoint(P(0), P(1), P(2), P(3));
        return (double)r.x * 100000.0 + (double)r.y;
    }
    case 2:
        return OnSegment(P(0), P(1), P(2));
    case 3:
        return IsAboveSegment(P(0), P(1), P(2));
    case 4:
    { // polygon = rectangle (0..3) closed, point 4
        Polygon poly{P(0), P(1), P(2), P(3), P(0)};
        return IsInsidePolygon(poly, P(4));
    }
    case 5:
        return GetDistanceFromPolygonToPoint(CreatePolygon(P(0)), P(1));
    case 6:
        return GetDistanceFromSegmentToPoint(P(0), P(1), P(2));
    case 7:
        return CreatePolygon(P(0), a[2], a[3], a[4])[(int)a[5]].x;
    case 8:
        return IsPolygonIntersectWithPolygon(CreatePolygon(P(0)), CreatePolygon(P(1)));
    case 9:
        return GetDistanceFromPolygonToSegment(CreatePolygon(P(0)), P(1), P(2));
    case 10:
        return (double)CombinePolygon(CreatePolygon(P(0)), CreatePolygon(P(1))).size();
    case 11:
        return IsSegmentIntersectWithPolygon(CreatePolygon(P(0)), P(1), P(2));
    case 12:
        return GetUnitVector(P(0), P(1)).x * 10.0 + GetUnitVector(P(0), P(1)).y;
    case 13:
        return IsPointOnLine(P(0), P(1), P(2));
    case 14:
        return GetDistanceFromPointToLine(P(0), P(1), P(2));
    case 15:
        return RadToZeroTo2P(a[0]);
    case 16:
        return RadNormalization(a[0]);
    case 17:
        return ConvertDegToRad(a[0]);
    // angle ranges: arguments in degrees, converted as the reference's tests do (ConvertDegToRad of a float)
    case 18:
        return IsOverlap(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])), AngleRange(ConvertDegToRad(a[2]), ConvertDegToRad(a[3])));
    case 19:
        return IsAngleRangeInclude(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])), AngleRange(ConvertDegToRad(a[2]), ConvertDegToRad(a[3])));
    case 20:
        return IsAngleRangeInclude(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])), ConvertDegToRad(a[2]));
    case 21: // include test against the end of a second range
        return IsAngleRangeInclude(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])),
                                   GetAngleRangeEnd(AngleRange(ConvertDegToRad(a[2]), ConvertDegToRad(a[3]))));
    case 22:
        return ConvertRadToDeg(FindAngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1]))[0].first);
    case 23:
        return ConvertRadToDeg(GetAngleRangeEnd(FindAngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1]))[0]));
    case 24:
        return FindCommonAngleRange(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])), AngleRange(ConvertDegToRad(a[2]), ConvertDegToRad(a[3]))).first;
    case 25:
        return GetAngleRangeEnd(FindCommonAngleRange(AngleRange(ConvertDegToRad(a[0]), ConvertDegToRad(a[1])), AngleRange(ConvertDegToRad(a[2]), ConvertDegToRad(a[3]))));
    case 26:
        return GetAngleDistance(ConvertDegToRad(a[0]), AngleRange(ConvertDegToRad(a[1]), ConvertDegToRad(a[2])));
    case 27:
        return IsEqual(ConvertDegToRad(a[0]), ConvertDegToRad(a[1]));
    case 28:
        return GetDistanceFromPolygonToPointAtAngle(CreatePolygon(P(0)), P(1), ConvertDegToRad(a[4]));
    case 29: // rectangle {origin, width, height}, point, angle in rad
        return GetDistanceFromPolygonToPointAtAngle(CreatePolygon(P(0), a[5], a[6]), P(1), a[4]);
    default:
        return -12345;
    }
}

// Wall-clock helper for bench.py's cpu_baseline: plans queries [0, n) one at a
// time on this thread; returns seconds, fills totals.
double orc_plan_many(void *h, const float *starts, const float *goals, size_t n, hyas_result *res)
{
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i)
        ((Ctx *)h)->plan(starts + 3 * i, goals + 3 * i, res + i);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}
