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

} // namespace db
} // namespace hyas
