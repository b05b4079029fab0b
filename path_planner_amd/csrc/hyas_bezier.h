// Cubic Bezier curves of curve_type 2, shared by host and device code.
//
// The reference has two unrelated implementations and uses one each:
//  * the analytic shot (src/hybrid_a_star.cpp:470-521) builds a CubicBezier::CubicBezier (include/cubic_bezier.h:31-44,
//    src/cubic_bezier.cpp): Eigen float matrices, value = G * B * [1 t t^2 t^3]^T;
//  * the heuristic table (src/lookup_table.cpp:226-277) builds a Bezier::Bezier<3> (include/bezier.h) and takes its
//    100-interval polyline length (:744-768) in Bernstein form (:100-103, :681-690).
// Both are float arithmetic with a few double detours (std::pow of a float and an integer promotes to double); this
// file keeps every rounding step where the reference has it.  Two facts are taken on trust because no output of the
// reference pins them: pow / powf return the correctly rounded power for the exponents 0..3, and Eigen (3.3 / 3.4)
// reduces the four products of a coefficient as (p0 + p1) + (p2 + p3) (ProductEvaluators.h coeff(), Redux.h
// redux_novec_unroller).
#pragma once
#include "../../include/hyas_math.h"

namespace hyas
{
namespace bz
{
// ---- shot curve: power basis ---------------------------------------------------------------------------------------
// Rows x and y of geometrical_constraint_matrix_ * basis_matrix_.  The heading row of the product is never read: the
// shot takes a sample's heading from the first derivative (src/hybrid_a_star.cpp:483).
struct Curve
{
    float gb[2][4];
};

HY_HD float dot4(const float a[4], float b0, float b1, float b2, float b3) { return (a[0] * b0 + a[1] * b1) + (a[2] * b2 + a[3] * b3); }

// CalculateControlPoints with use_random_ == false (src/cubic_bezier.cpp:113-131), then G * B with
// B = [1 -3 3 -1; 0 3 -6 3; 0 0 3 -3; 0 0 0 1] (include/cubic_bezier.h:33-36)
HY_HD Curve make(const float start[3], const float goal[3])
{
    const float dx = goal[0] - start[0], dy = goal[1] - start[1];
    const float third = hy_sqrtf(dx * dx + dy * dy) / 3;
    float g[2][4];
    g[0][0] = start[0], g[1][0] = start[1];
    g[0][1] = start[0] + hy_cosf(start[2]) * third, g[1][1] = start[1] + hy_sinf(start[2]) * third;
    g[0][2] = goal[0] - hy_cosf(goal[2]) * third, g[1][2] = goal[1] - hy_sinf(goal[2]) * third;
    g[0][3] = goal[0], g[1][3] = goal[1];
    Curve c;
    for (int r = 0; r < 2; ++r)
    {
        c.gb[r][0] = dot4(g[r], 1.f, 0.f, 0.f, 0.f);
        c.gb[r][1] = dot4(g[r], -3.f, 3.f, 0.f, 0.f);
        c.gb[r][2] = dot4(g[r], 3.f, -6.f, 3.f, 0.f);
        c.gb[r][3] = dot4(g[r], -1.f, 3.f, -3.f, 1.f);
    }
    return c;
}

// One parameter value: the three coefficient vectors of src/cubic_bezier.cpp:16-51.  std::pow(float, unsigned) is the
// double overload; the unsigned `i - 1 < 0` tests never hold, and the terms they were meant to skip are
// 0 * pow(t, 2^32 - 1) = 0 for the t in [0, 1) the shot uses.
struct Powers
{
    float t1, t2, t3; // t, t^2, t^3
    float d2, d3;     // 2t, 3t^2
    float s3;         // 6t
};
HY_HD Powers powers(float t)
{
    const double td = (double)t, sq = td * td;
    Powers p;
    p.t1 = t, p.t2 = (float)sq, p.t3 = (float)(sq * td);
    p.d2 = (float)(2.0 * td), p.d3 = (float)(3.0 * sq);
    p.s3 = (float)(6.0 * td);
    return p;
}
HY_HD void value(const Curve &c, const Powers &p, float &x, float &y) // GetValueAt, :149-156
{
    x = dot4(c.gb[0], 1.f, p.t1, p.t2, p.t3), y = dot4(c.gb[1], 1.f, p.t1, p.t2, p.t3);
}
HY_HD void first(const Curve &c, const Powers &p, float &x, float &y) // GetFirstOrderDerivativeValueAt, :52-60
{
    x = dot4(c.gb[0], 0.f, 1.f, p.d2, p.d3), y = dot4(c.gb[1], 0.f, 1.f, p.d2, p.d3);
}
HY_HD void second(const Curve &c, const Powers &p, float &x, float &y) // GetSecondOrderDerivativeValueAt, :61-68
{
    x = dot4(c.gb[0], 0.f, 0.f, 2.f, p.s3), y = dot4(c.gb[1], 0.f, 0.f, 2.f, p.s3);
}
// GetAngleAt (:158-165) and GetCurvatureAt (:224-241) of one parameter value
HY_HD void angle_curvature(const Curve &c, const Powers &p, float &angle, float &curvature)
{
    float fx, fy, sx, sy;
    first(c, p, fx, fy);
    second(c, p, sx, sy);
    angle = hy_atan2f(fy, fx);
    const float n = hy_sqrtf(fx * fx + fy * fy);
    if (n != 0)
    {
        const double nd = (double)n;
        curvature = (float)((double)hy_fabsf(fx * sy - sx * fy) / ((nd * nd) * nd));
    }
    else
        curvature = 100000;
}
// parameter of polyline vertex i of CalculateLength (:166-178): i / 100.0 narrowed to the `const float &t` argument
HY_HD float length_param(unsigned i) { return (float)((double)i / 100.0); }

// ---- table curve: Bernstein form -----------------------------------------------------------------------------------
#define HYAS_BZ_INTERVALS 100
// Parameter of polyline vertex i of Bezier::length(100): t starts at 0 and grows by dt = 1.0f / 100 per interval; the
// vertex of interval i is evaluated at t + dt, the value t then takes itself (include/bezier.h:750-763).
HY_HD float lib_param(unsigned i)
{
    const float dt = 1.0f / (float)HYAS_BZ_INTERVALS;
    float t = 0.0f;
    for (unsigned k = 0; k < i; ++k)
        t += dt;
    return t;
}
// binomialCoefficients[n] * polynomialCoefficients[n].valueAt(t), the factor control point n is multiplied by
// (include/bezier.h:687): float(binomial) * float(pow(1.0f - t, 3 - n) * pow(t, float(n))), the first pow in double
HY_HD void lib_weights(float t, float w[4])
{
    const double u = (double)(1.0f - t), td = (double)t;
    const double up[4] = {1.0, u, u * u, (u * u) * u};
    const float tp[4] = {1.f, t, (float)(td * td), (float)((td * td) * td)};
    const float binom[4] = {1.f, 3.f, 3.f, 1.f};
    for (int n = 0; n < 4; ++n)
        w[n] = binom[n] * (float)(up[3 - n] * (double)tp[n]);
}
// Control points of the table entry (x, y, theta), src/lookup_table.cpp:251-260: from the origin with heading 0
HY_HD void lib_control(float x, float y, float theta, float px[4], float py[4])
{
    const float t = hy_sqrtf(x * x + y * y) / 3;
    px[0] = 0.f, py[0] = 0.f;
    px[1] = 0.f + hy_cosf(0.f) * t, py[1] = 0.f + hy_sinf(0.f) * t;
    px[2] = x - hy_cosf(theta) * t, py[2] = y - hy_sinf(theta) * t;
    px[3] = x, py[3] = y;
}
HY_HD float lib_axis(const float w[4], const float p[4]) // Bezier::valueAt(t, axis), include/bezier.h:681-690
{
    float sum = 0;
    for (int n = 0; n < 4; ++n)
        sum += w[n] * p[n];
    return sum;
}
} // namespace bz
} // namespace hyas
