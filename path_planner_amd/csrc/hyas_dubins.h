// hyas_dubins.h -- Dubins curves in float, as the reference evaluates them for
// the analytic shot of curve_type 0 (src/dubins.cpp:81-314; caller
// src/hybrid_a_star.cpp:378-427).  Host+device.  Every expression keeps the
// reference's operand order and its float/double mix: the <math.h> float
// overloads on float arguments, double where a literal like 2.0, 6., 8. or
// M_PI promotes the sub-expression.
#pragma once
#include "../../include/hyas_math.h"

#define DB_HD __host__ __device__ inline

namespace hyas
{
namespace db
{

struct Path
{
    float qi[3];    // start configuration
    float param[3]; // normalised segment lengths
    float rho;
    int type;       // word index, order LSL LSR RSL RSR RLR LRL (include/dubins.h:30-35)
};

// segment kinds of word w (src/dubins.cpp:63-70): 0 left, 1 straight, 2 right
DB_HD int seg_kind(int w, int s)
{
    const int packed[6] = {0x010, 0x210, 0x012, 0x212, 0x202, 0x020}; // nibble s = kind of segment s
    return (packed[w] >> (4 * s)) & 0xf;
}

// x - y*floor(x/y), src/dubins.cpp:81-84
DB_HD float fmodr(float x, float y) { return x - y * hy_floorf(x / y); }
DB_HD float mod2pi(float t) { return fmodr(t, (float)(2 * HY_PI)); }

// One of the six words (src/dubins.cpp:168-265).  sa, sb, ca, cb, cab are
// sinf/cosf of alpha, beta and cosf(alpha - beta).
DB_HD bool word(int w, float alpha, float beta, float d, float sa, float sb, float ca, float cb, float cab, float *o)
{
    switch (w)
    {
    case 0: // LSL
    {
        float tmp0 = d + sa - sb;
        float p2 = 2 + (d * d) - (2 * cab) + (2 * d * (sa - sb));
        if (p2 < 0)
            return false;
        float tmp1 = hy_atan2f((cb - ca), tmp0);
        o[0] = mod2pi(-alpha + tmp1), o[1] = hy_sqrtf(p2), o[2] = mod2pi(beta - tmp1);
        return true;
    }
    case 1: // LSR
    {
        float p2 = -2 + (d * d) + (2 * cab) + (2 * d * (sa + sb));
        if (p2 < 0)
            return false;
        float p = hy_sqrtf(p2);
        float tmp2 = (float)((double)hy_atan2f((-ca - cb), (d + sa + sb)) - hy_atan2(-2.0, (double)p));
        o[0] = mod2pi(-alpha + tmp2), o[1] = p, o[2] = mod2pi(-mod2pi(beta) + tmp2);
        return true;
    }
    case 2: // RSL
    {
        float p2 = (d * d) - 2 + (2 * cab) - (2 * d * (sa + sb));
        if (p2 < 0)
            return false;
        float p = hy_sqrtf(p2);
        float tmp2 = (float)((double)hy_atan2f((ca + cb), (d - sa - sb)) - hy_atan2(2.0, (double)p));
        o[0] = mod2pi(alpha - tmp2), o[1] = p, o[2] = mod2pi(beta - tmp2);
        return true;
    }
    case 3: // RSR
    {
        float tmp0 = d - sa + sb;
        float p2 = 2 + (d * d) - (2 * cab) + (2 * d * (sb - sa));
        if (p2 < 0)
            return false;
        float tmp1 = hy_atan2f((ca - cb), tmp0);
        o[0] = mod2pi(alpha - tmp1), o[1] = hy_sqrtf(p2), o[2] = mod2pi(-beta + tmp1);
        return true;
    }
    case 4: // RLR
    {
        float tmp = (float)((6. - (double)(d * d) + (double)(2 * cab) + (double)(2 * d * (sa - sb))) / 8.);
        if (hy_fabs((double)tmp) > 1)
            return false;
        float p = mod2pi((float)(2 * HY_PI - (double)hy_acosf(tmp)));
        float t = mod2pi(alpha - hy_atan2f(ca - cb, d - sa + sb) + mod2pi((float)((double)p / 2.)));
        o[0] = t, o[1] = p, o[2] = mod2pi(alpha - beta - t + mod2pi(p));
        return true;
    }
    default: // LRL
    {
        float tmp = (float)((6. - (double)(d * d) + (double)(2 * cab) + (double)(2 * d * (-sa + sb))) / 8.);
        if (hy_fabs((double)tmp) > 1)
            return false;
        float p = mod2pi((float)(2 * HY_PI - (double)hy_acosf(tmp)));
        float t = mod2pi((float)((double)(-alpha - hy_atan2f(ca - cb, d + sa - sb)) + (double)p / 2.));
        o[0] = t, o[1] = p, o[2] = mod2pi(mod2pi(beta) - alpha - t + mod2pi(p));
        return true;
    }
    }
}

// dubins_init (src/dubins.cpp:91-140): shortest word, the first wins a tie.
// 0 on success, 3 for rho <= 0, 4 when no word connects.
DB_HD int init(const float q0[3], const float q1[3], float rho, Path &path)
{
    const float dx = q1[0] - q0[0], dy = q1[1] - q0[1];
    const float D = hy_sqrtf(dx * dx + dy * dy);
    const float d = D / rho;
    if (rho <= 0.)
        return 3;
    const float theta = mod2pi(hy_atan2f(dy, dx));
    const float alpha = mod2pi(q0[2] - theta), beta = mod2pi(q1[2] - theta);
    path.qi[0] = q0[0], path.qi[1] = q0[1], path.qi[2] = q0[2];
    path.rho = rho;
    const float sa = hy_sinf(alpha), sb = hy_sinf(beta), ca = hy_cosf(alpha), cb = hy_cosf(beta), cab = hy_cosf(alpha - beta);
    float best = __builtin_inff();
    int best_word = -1;
    for (int w = 0; w < 6; ++w)
    {
        float o[3];
        if (word(w, alpha, beta, d, sa, sb, ca, cb, cab, o))
        {
            const float cost = o[0] + o[1] + o[2];
            if (cost < best)
                best = cost, best_word = w, path.param[0] = o[0], path.param[1] = o[1], path.param[2] = o[2];
        }
    }
    if (best_word < 0)
        return 4;
    path.type = best_word;
    return 0;
}

// dubins_path_length, src/dubins.cpp:142-150
DB_HD float length(const Path &p)
{
    float l = 0.f;
    l += p.param[0];
    l += p.param[1];
    l += p.param[2];
    return l * p.rho;
}

// dubins_segment, src/dubins.cpp:267-284 (float variant the path sampler calls)
DB_HD void segment(float t, const float qi[3], float qt[3], int kind)
{
    if (kind == 0)
    {
        qt[0] = qi[0] + hy_sinf(qi[2] + t) - hy_sinf(qi[2]);
        qt[1] = qi[1] - hy_cosf(qi[2] + t) + hy_cosf(qi[2]);
        qt[2] = qi[2] + t;
    }
    else if (kind == 2)
    {
        qt[0] = qi[0] - hy_sinf(qi[2] - t) + hy_sinf(qi[2]);
        qt[1] = qi[1] + hy_cosf(qi[2] - t) - hy_cosf(qi[2]);
        qt[2] = qi[2] - t;
    }
    else
    {
        qt[0] = qi[0] + hy_cosf(qi[2]) * t;
        qt[1] = qi[1] + hy_sinf(qi[2]) * t;
        qt[2] = qi[2];
    }
}

// dubins_path_sample, src/dubins.cpp:286-314; 2 when t is outside [0, length)
DB_HD int sample(const Path &p, float t, float q[3])
{
    if (t < 0 || t >= length(p))
        return 2;
    const float tp = t / p.rho;
    const float qi[3] = {0, 0, p.qi[2]};
    const int k0 = seg_kind(p.type, 0), k1 = seg_kind(p.type, 1), k2 = seg_kind(p.type, 2);
    const float p1 = p.param[0], p2 = p.param[1];
    float q1[3], q2[3];
    segment(p1, qi, q1, k0);
    segment(p2, q1, q2, k1);
    if (tp < p1)
        segment(tp, qi, q, k0);
    else if (tp < (p1 + p2))
        segment(tp - p1, q1, q, k1);
    else
        segment(tp - p1 - p2, q2, q, k2);
    q[0] = q[0] * p.rho + p.qi[0];
    q[1] = q[1] * p.rho + p.qi[1];
    q[2] = mod2pi(q[2]);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The Dubins distance of the heuristic table (src/lookup_table.cpp:147-185): the reference calls
// ompl::base::DubinsStateSpace::distance, double arithmetic, which is not part of the reference tree.  This restates
// OMPL's published algorithm (ompl/base/spaces/src/DubinsStateSpace.cpp, releases 1.4 - 1.6: the six words of Shkel and
// Lumelsky's classification with OMPL's own tolerances DUBINS_EPS = 1e-6 and DUBINS_ZERO = -1e-7), words tried in
// OMPL's order LSL, RSR, RSL, LSR, RLR, LRL with a strictly shorter word replacing the current one.  A word that does
// not connect has OMPL's default length, the largest double.  No reference fixture holds an output of it.
namespace ompl
{
#define DB_OMPL_EPS 1e-6
#define DB_OMPL_ZERO (-1e-7)
#define DB_OMPL_MAX 1.7976931348623157e308
DB_HD double mod2pi(double x)
{
    const double twopi = 2. * HY_PI;
    if (x < 0 && x > DB_OMPL_ZERO)
        return 0;
    double xm = x - twopi * hy_floor(x / twopi);
    if (twopi - xm < .5 * DB_OMPL_EPS)
        xm = 0.;
    return xm;
}
// length t + p + q of word w (0 LSL, 1 RSR, 2 RSL, 3 LSR, 4 RLR, 5 LRL) between headings alpha and beta at distance d
DB_HD double word_length(int w, double d, double alpha, double beta, double sa, double ca, double sb, double cb)
{
    const double twopi = 2. * HY_PI;
    double t, p, q;
    if (w == 0)
    {
        const double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sa - sb));
        if (!(tmp >= DB_OMPL_ZERO))
            return DB_OMPL_MAX;
        const double theta = hy_atan2(cb - ca, d + sa - sb);
        t = mod2pi(-alpha + theta), p = hy_sqrt(tmp > 0. ? tmp : 0.), q = mod2pi(beta - theta);
    }
    else if (w == 1)
    {
        const double tmp = 2. + d * d - 2. * (ca * cb + sa * sb - d * (sb - sa));
        if (!(tmp >= DB_OMPL_ZERO))
            return DB_OMPL_MAX;
        const double theta = hy_atan2(ca - cb, d - sa + sb);
        t = mod2pi(alpha - theta), p = hy_sqrt(tmp > 0. ? tmp : 0.), q = mod2pi(-beta + theta);
    }
    else if (w == 2)
    {
        const double tmp = d * d - 2. + 2. * (ca * cb + sa * sb - d * (sa + sb));
        if (!(tmp >= DB_OMPL_ZERO))
            return DB_OMPL_MAX;
        p = hy_sqrt(tmp > 0. ? tmp : 0.);
        const double theta = hy_atan2(ca + cb, d - sa - sb) - hy_atan2(2., p);
        t = mod2pi(alpha - theta), q = mod2pi(beta - theta);
    }
    else if (w == 3)
    {
        const double tmp = -2. + d * d + 2. * (ca * cb + sa * sb + d * (sa + sb));
        if (!(tmp >= DB_OMPL_ZERO))
            return DB_OMPL_MAX;
        p = hy_sqrt(tmp > 0. ? tmp : 0.);
        const double theta = hy_atan2(-ca - cb, d + sa + sb) - hy_atan2(-2., p);
        t = mod2pi(-alpha + theta), q = mod2pi(-beta + theta);
    }
    else if (w == 4)
    {
        const double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb + d * (sa - sb)));
        if (!(hy_fabs(tmp) < 1.))
            return DB_OMPL_MAX;
        p = twopi - hy_acos(tmp);
        const double theta = hy_atan2(ca - cb, d - sa + sb);
        t = mod2pi(alpha - theta + .5 * p), q = mod2pi(alpha - beta - t + p);
    }
    else
    {
        const double tmp = .125 * (6. - d * d + 2. * (ca * cb + sa * sb - d * (sa - sb)));
        if (!(hy_fabs(tmp) < 1.))
            return DB_OMPL_MAX;
        p = twopi - hy_acos(tmp);
        const double theta = hy_atan2(-ca + cb, d + sa - sb);
        t = mod2pi(-alpha + theta + .5 * p), q = mod2pi(beta - alpha - t + p);
    }
    return t + p + q;
}
// DubinsStateSpace::distance((x1, y1, th1), (x2, y2, th2)) for turning radius rho (not symmetric: the default)
DB_HD double distance(double x1, double y1, double th1, double x2, double y2, double th2, double rho)
{
    const double dx = x2 - x1, dy = y2 - y1, d = hy_sqrt(dx * dx + dy * dy) / rho, th = hy_atan2(dy, dx);
    const double alpha = mod2pi(th1 - th), beta = mod2pi(th2 - th);
    if (d < DB_OMPL_EPS && hy_fabs(alpha - beta) < DB_OMPL_EPS)
        return rho * (0. + d + 0.);
    double sa, ca, sb, cb;
    hy_sincos(alpha, &sa, &ca), hy_sincos(beta, &sb, &cb);
    double best = word_length(0, d, alpha, beta, sa, ca, sb, cb);
    for (int w = 1; w < 6; ++w)
    {
        const double len = word_length(w, d, alpha, beta, sa, ca, sb, cb);
        if (len < best)
            best = len;
    }
    return rho * best;
}
} // namespace ompl

} // namespace db
} // namespace hyas
