/*
 * hyas_math.h -- portable, bit-reproducible elementary functions.
 *
 * The reference planner calls glibc's sin/cos/atan2/acos/asin/fmod (float and
 * double overloads) all over its hot path, e.g.
 *   src/utility.cpp:260, :490-493      (acosf, cosf, sinf)
 *   src/hybrid_a_star.cpp:692-740      (sinf, cosf)
 *   src/ReedsShepp.cpp:50-70, :576-625 (double sin/cos/atan2/asin/acos/fmod)
 * A device libm is not bit-identical to glibc, so the GPU kernels and the CPU
 * oracle both evaluate these functions through THIS header instead.  Every
 * function is built from IEEE-754 +,-,*,/ and sqrt only (no fused
 * multiply-add: compile every translation unit that includes this header with
 * -ffp-contract=off), therefore the same input gives the same bits on an
 * x86-64 host and on a gfx950 device.
 *
 * Accuracy: about 1 ulp in double; the float forms evaluate in double and
 * round once, so they are the correctly rounded float result in all but
 * ~1e-9 of the cases.  They are NOT guaranteed to equal glibc bit for bit.
 *
 * Domain: |x| < 2^20 for sin/cos (arguments on the planner path are bounded
 * by a few turns).  NaN in -> NaN out.
 */
#ifndef HYAS_MATH_H
#define HYAS_MATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define HY_HD __host__ __device__ inline
/* the larger routines (sin, cos, atan2, fmod): inlined by default, which measured faster in the
 * search kernels than real calls (no register save/restore through scratch memory);
 * -DHYAS_MATH_CALLS makes them calls, for smaller code */
#ifdef HYAS_MATH_CALLS
#define HY_HD_CALL __host__ __device__ inline __attribute__((noinline))
#else
#define HY_HD_CALL HY_HD
#endif
#else
#define HY_HD inline
#define HY_HD_CALL inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define HY_PI 3.14159265358979323846 /* == M_PI */
#define HY_PI_LO 1.2246467991473532e-16
#define HY_PIO2 1.5707963267948966
#define HY_PIO2_LO 6.123233995736766e-17

HY_HD double hy_fabs(double x) { return __builtin_fabs(x); }
HY_HD float hy_fabsf(float x) { return __builtin_fabsf(x); }
HY_HD double hy_sqrt(double x) { return __builtin_sqrt(x); }
HY_HD float hy_sqrtf(float x) { return __builtin_sqrtf(x); }
HY_HD double hy_floor(double x) { return __builtin_floor(x); }
HY_HD float hy_floorf(float x) { return __builtin_floorf(x); }
HY_HD bool hy_isnan(double x) { return x != x; }

/* std::round(float): halfway cases away from zero. */
HY_HD float hy_roundf(float x)
{
    float a = hy_fabsf(x);
    if (!(a < 8388608.0f)) /* |x| >= 2^23, inf or NaN: already integral */
        return x;
    float f = hy_floorf(a);
    float r = (a - f >= 0.5f) ? f + 1.0f : f; /* a - f is exact */
    return (x < 0.0f || (x == 0.0f && __builtin_signbit(x))) ? -r : r;
}

/* C fmod: exact remainder with the sign of x.  Long division on the
 * significands; no rounding anywhere. */
HY_HD_CALL double hy_fmod(double x, double y)
{
    if (hy_isnan(x) || hy_isnan(y) || y == 0.0 || hy_fabs(x) > 1.7976931348623157e308)
        return (x * y) / (x * y); /* NaN */
    double ax = hy_fabs(x), ay = hy_fabs(y);
    if (ay > 1.7976931348623157e308 || ax < ay)
        return x;
    /* Subtract y * 2^k for decreasing k.  Each subtraction is exact because
     * ax and t share an exponent range of one binade (Sterbenz). */
    double t = ay;
    while (t <= ax * 0.5)
        t = t + t; /* exact doubling; stops before overflow since t <= ax */
    while (true)
    {
        if (ax >= t)
            ax = ax - t;
        if (t == ay)
            break;
        t = t * 0.5;
    }
    return (x < 0.0) ? -ax : ax;
}

/* ---- sin / cos ---------------------------------------------------------- */

/* r = x - k*pi/2 with pi/2 split into 33+33+53 bits: k*P1 and k*P2 are exact
 * for |k| < 2^20. */
HY_HD double hy_rem_pio2(double x, int *quadrant)
{
    const double INV_PIO2 = 0.6366197723675814; /* 0x1.45f306dc9c883p-1 */
    const double P1 = 1.5707963267341256;       /* 0x1.921fb54400000p+0  */
    const double P2 = 6.077100506303966e-11;    /* 0x1.0b4611a600000p-34 */
    const double P3 = 2.0222662487959506e-21;   /* 0x1.3198a2e037073p-69 */
    double q = x * INV_PIO2;
    int k = (int)(q + (q >= 0.0 ? 0.5 : -0.5));
    double fk = (double)k;
    double r = ((x - fk * P1) - fk * P2) - fk * P3;
    *quadrant = k & 3;
    return r;
}

/* Taylor kernels on |r| <= pi/4 (+ a hair); truncation error < 1e-19. */
HY_HD double hy_sin_kernel(double r)
{
    double z = r * r;
    double p = -2.8114572543455206e-15;          /* -1/17! */
    p = p * z + 7.647163731819816e-13;           /*  1/15! */
    p = p * z + -1.6059043836821613e-10;         /* -1/13! */
    p = p * z + 2.505210838544172e-08;           /*  1/11! */
    p = p * z + -2.7557319223985893e-06;         /* -1/9!  */
    p = p * z + 0.0001984126984126984;           /*  1/7!  */
    p = p * z + -0.008333333333333333;           /* -1/5!  */
    p = p * z + 0.16666666666666666;             /*  1/3! (sign folded below) */
    /* sin r = r - r^3/6 + ... ; p currently holds 1/6 - z/120 + ...  */
    return r - (r * z) * p;
}

HY_HD double hy_cos_kernel(double r)
{
    double z = r * r;
    double p = 4.779477332387385e-14;            /*  1/16! */
    p = p * z + -1.1470745597729725e-11;         /* -1/14! */
    p = p * z + 2.08767569878681e-09;            /*  1/12! */
    p = p * z + -2.755731922398589e-07;          /* -1/10! */
    p = p * z + 2.48015873015873e-05;            /*  1/8!  */
    p = p * z + -0.001388888888888889;           /* -1/6!  */
    p = p * z + 0.041666666666666664;            /*  1/4!  */
    /* cos r = 1 - z/2 + z^2 * p */
    return 1.0 - (0.5 * z - (z * z) * p);
}

HY_HD_CALL double hy_sin(double x)
{
    int q;
    double r = hy_rem_pio2(x, &q);
    switch (q)
    {
    case 0:
        return hy_sin_kernel(r);
    case 1:
        return hy_cos_kernel(r);
    case 2:
        return -hy_sin_kernel(r);
    default:
        return -hy_cos_kernel(r);
    }
}

HY_HD_CALL double hy_cos(double x)
{
    int q;
    double r = hy_rem_pio2(x, &q);
    switch (q)
    {
    case 0:
        return hy_cos_kernel(r);
    case 1:
        return -hy_sin_kernel(r);
    case 2:
        return -hy_cos_kernel(r);
    default:
        return hy_sin_kernel(r);
    }
}

/* sin and cos of one argument from one range reduction and both kernels:
 * *s == hy_sin(x) and *c == hy_cos(x) bit for bit (the same operations on the
 * same values; only the selection by quadrant is written without branches). */
HY_HD void hy_sincos(double x, double *s, double *c)
{
    int q;
    double r = hy_rem_pio2(x, &q);
    double sk = hy_sin_kernel(r), ck = hy_cos_kernel(r);
    *s = (q == 0) ? sk : (q == 1) ? ck : (q == 2) ? -sk : -ck;
    *c = (q == 0) ? ck : (q == 1) ? -sk : (q == 2) ? -ck : sk;
}

/* ---- atan / atan2 / asin / acos ---------------------------------------- */

/* atan on [0, 1]: nearest table point c = i/4, then the odd Taylor series in
 * t = (x - c) / (1 + x c), |t| <= 0.125 -> truncation error < 1e-20.
 * hy_atan01_series: the part after the table point (c, hi = atan(c)) is chosen. */
HY_HD double hy_atan01_series(double x, double c, double hi)
{
    double t = (x - c) / (1.0 + x * c);
    double z = t * t;
    double p = 1.0 / 21.0;
    p = -1.0 / 19.0 + z * p;
    p = 1.0 / 17.0 + z * p;
    p = -1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = -1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = -1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = -1.0 / 3.0 + z * p;
    return hi + (t + (t * z) * p);
}

HY_HD double hy_atan01(double x)
{
    int i = (int)(x * 4.0 + 0.5);
    double c = 0.25 * (double)i;
    /* atan(i/4), picked with selects: wide hardware runs every arm of a branch some lane takes */
    const double lo01 = (i & 1) ? 0.24497866312686414 : 0.0, lo23 = (i & 1) ? 0.6435011087932844 : 0.4636476090008061;
    const double hi = (i >= 4) ? 0.7853981633974483 : (i & 2) ? lo23 : lo01;
    return hy_atan01_series(x, c, hi);
}

/* The same with atan(i/4) read from the caller's copy of HY_QUARTER_ATANS (eight entries, so that no index, not
 * even that of a NaN, reads outside it): for kernels that keep the copy in fast local memory, one read instead
 * of the chain of selects. */
#define HY_QUARTER_ATANS                                                                                                  \
    {                                                                                                                     \
        0.0, 0.24497866312686414, 0.4636476090008061, 0.6435011087932844, 0.7853981633974483, 0.7853981633974483,          \
            0.7853981633974483, 0.7853981633974483                                                                        \
    }
HY_HD double hy_atan01_q(double x, const double *quarter_atans)
{
    int i = (int)(x * 4.0 + 0.5);
    return hy_atan01_series(x, 0.25 * (double)i, quarter_atans[i & 7]);
}

/* atan of a non-negative finite or infinite argument: one pass through the series for both halves, above 1
 * through the reciprocal (0 for inf).  On the device the reciprocal stays inside its branch: lanes whose
 * arguments are all at most 1 skip the division. */
#if defined(__HIP_DEVICE_COMPILE__)
#define HY_KEEP_IN_BRANCH(v) asm volatile("" : "+v"(v))
#else
#define HY_KEEP_IN_BRANCH(v) (void)0
#endif
HY_HD double hy_atan_pos(double x)
{
    const bool small = x <= 1.0;
    double arg = x;
    if (!small)
    {
        arg = 1.0 / x;
        HY_KEEP_IN_BRANCH(arg);
    }
    const double a = hy_atan01(arg);
    return small ? a : (HY_PIO2 - a) + HY_PIO2_LO;
}
HY_HD double hy_atan_pos_q(double x, const double *quarter_atans)
{
    const bool small = x <= 1.0;
    double arg = x;
    if (!small)
    {
        arg = 1.0 / x;
        HY_KEEP_IN_BRANCH(arg);
    }
    const double a = hy_atan01_q(arg, quarter_atans);
    return small ? a : (HY_PIO2 - a) + HY_PIO2_LO;
}

HY_HD double hy_copysign(double m, double s)
{
    return __builtin_copysign(m, s);
}

HY_HD_CALL double hy_atan2(double y, double x)
{
    if (hy_isnan(x) || hy_isnan(y))
        return x + y;
    bool xneg = __builtin_signbit(x);
    if (y == 0.0)
        return xneg ? hy_copysign(HY_PI, y) : y; /* +-0 or +-pi */
    if (x == 0.0)
        return hy_copysign(HY_PIO2, y);
    double ax = hy_fabs(x), ay = hy_fabs(y);
    double a;
    if (ax > 1.7976931348623157e308 && ay > 1.7976931348623157e308)
        a = 0.7853981633974483; /* inf/inf */
    else
        a = hy_atan_pos(ay / ax);
    if (xneg)
        a = (HY_PI - a) + HY_PI_LO;
    return hy_copysign(a, y);
}
/* hy_atan2 through hy_atan01_q */
HY_HD double hy_atan2_q(double y, double x, const double *quarter_atans)
{
    if (hy_isnan(x) || hy_isnan(y))
        return x + y;
    bool xneg = __builtin_signbit(x);
    if (y == 0.0)
        return xneg ? hy_copysign(HY_PI, y) : y;
    if (x == 0.0)
        return hy_copysign(HY_PIO2, y);
    double ax = hy_fabs(x), ay = hy_fabs(y);
    double a;
    if (ax > 1.7976931348623157e308 && ay > 1.7976931348623157e308)
        a = 0.7853981633974483;
    else
        a = hy_atan_pos_q(ay / ax, quarter_atans);
    if (xneg)
        a = (HY_PI - a) + HY_PI_LO;
    return hy_copysign(a, y);
}

HY_HD double hy_asin(double x)
{
    return hy_atan2(x, hy_sqrt((1.0 - x) * (1.0 + x)));
}

HY_HD double hy_acos(double x)
{
    return hy_atan2(hy_sqrt((1.0 - x) * (1.0 + x)), x);
}
HY_HD double hy_asin_q(double x, const double *quarter_atans)
{
    return hy_atan2_q(x, hy_sqrt((1.0 - x) * (1.0 + x)), quarter_atans);
}
HY_HD double hy_acos_q(double x, const double *quarter_atans)
{
    return hy_atan2_q(hy_sqrt((1.0 - x) * (1.0 + x)), x, quarter_atans);
}

/* ---- float forms: evaluate in double, round once ------------------------ */
HY_HD float hy_sinf(float x) { return (float)hy_sin((double)x); }
HY_HD float hy_cosf(float x) { return (float)hy_cos((double)x); }
HY_HD void hy_sincosf(float x, float *s, float *c) /* == (hy_sinf(x), hy_cosf(x)) */
{
    double sd, cd;
    hy_sincos((double)x, &sd, &cd);
    *s = (float)sd, *c = (float)cd;
}
HY_HD float hy_acosf(float x) { return (float)hy_acos((double)x); }
HY_HD float hy_atan2f(float y, float x) { return (float)hy_atan2((double)y, (double)x); }

#endif /* HYAS_MATH_H */
