// hyas_rs.h -- Reeds-Shepp solver laid out for wide execution.
//
// The reference (src/ReedsShepp.cpp:73-503) walks 44 closed-form candidates in
// a fixed order through five family functions (CSC, CCC, CCCC, CCSC, CCSCC)
// and keeps a candidate when it is strictly shorter than the running best.
// Here every candidate is a row of one table -- (base formula, symmetry,
// backwards?, word) -- so 44 lanes can evaluate one candidate each and a
// per-family arg-min (lowest index on ties) reproduces the sequential scan
// exactly; a single thread can also walk the table in order.
#pragma once
#include "../../include/hyas_math.h"

#define RS_HD __host__ __device__ __forceinline__
#ifdef HYAS_INLINE_GEOM
#define RS_HDC RS_HD
#else
#define RS_HDC __host__ __device__ inline __attribute__((noinline))
#endif

namespace hyas
{
namespace rs
{

#define RS_PI HY_PI
// 10 * DBL_EPSILON, src/ReedsShepp.cpp:48
#define RS_ZERO 2.220446049250313e-15

struct Path
{
    int type;      // row of the segment-type table below
    double len[5]; // signed segment lengths, in units of the turning radius
    double total;  // sum of |len|
};

// Segment kinds per word, src/ReedsShepp.cpp:505-525 (1 left, 2 straight, 3 right, 0 none)
RS_HD int seg_kind(int type, int i)
{
    // packed 3 bits per segment
    const unsigned w[18] = {
        1 | 3 << 3 | 1 << 6,           3 | 1 << 3 | 3 << 6,           1 | 3 << 3 | 1 << 6 | 3 << 9,  3 | 1 << 3 | 3 << 6 | 1 << 9,
        1 | 3 << 3 | 2 << 6 | 1 << 9,  3 | 1 << 3 | 2 << 6 | 3 << 9,  1 | 2 << 3 | 3 << 6 | 1 << 9,  3 | 2 << 3 | 1 << 6 | 3 << 9,
        1 | 3 << 3 | 2 << 6 | 3 << 9,  3 | 1 << 3 | 2 << 6 | 1 << 9,  3 | 2 << 3 | 3 << 6 | 1 << 9,  1 | 2 << 3 | 1 << 6 | 3 << 9,
        1 | 2 << 3 | 3 << 6,           3 | 2 << 3 | 1 << 6,           1 | 2 << 3 | 1 << 6,           3 | 2 << 3 | 3 << 6,
        1 | 3 << 3 | 2 << 6 | 1 << 9 | 3 << 12, 3 | 1 << 3 | 2 << 6 | 3 << 9 | 1 << 12};
    return (w[type] >> (3 * i)) & 7;
}

RS_HDC double mod2pi(double x) // :50-58
{
    double v = hy_fmod(x, 2. * RS_PI);
    if (v < -RS_PI)
        v += 2. * RS_PI;
    else if (v > RS_PI)
        v -= 2. * RS_PI;
    return v;
}
RS_HD void polar(double x, double y, double &r, double &theta) // :59-63
{
    r = hy_sqrt(x * x + y * y);
    theta = hy_atan2(y, x);
}
RS_HD void tau_omega(double u, double v, double xi, double eta, double phi, double &tau, double &omega) // :64-70
{
    double delta = mod2pi(u - v), A = hy_sin(u) - hy_sin(delta), B = hy_cos(u) - hy_cos(delta) - 1.;
    double t1 = hy_atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2. * (hy_cos(delta) - hy_cos(v) - hy_cos(u)) + 3;
    tau = (t2 < 0) ? mod2pi(t1 + RS_PI) : mod2pi(t1);
    omega = mod2pi(tau - u + v - phi);
}

// The eight base formulas, :73-108, :160-175, :229-260, :312-345, :449-467.
// (out of line; the result travels by value: reference parameters would live on the stack)
struct TUV
{
    double t, u, v;
    int ok;
};
// (sp, cp): sine and cosine of phi
RS_HD bool base_formula_body_sc(int f, double x, double y, double phi, double sp, double cp, double &t, double &u, double &v)
{
    switch (f)
    {
    case 0: // LpSpLp (8.1)
    {
        polar(x - sp, y - 1. + cp, u, t);
        if (t >= -RS_ZERO)
        {
            v = mod2pi(phi - t);
            if (v >= -RS_ZERO)
                return true;
        }
        return false;
    }
    case 1: // LpSpRp (8.2)
    {
        double t1, u1;
        polar(x + sp, y - 1. - cp, u1, t1);
        u1 = u1 * u1;
        if (u1 >= 4.)
        {
            u = hy_sqrt(u1 - 4.);
            double theta = hy_atan2(2., u);
            t = mod2pi(t1 + theta);
            v = mod2pi(t - phi);
            return t >= -RS_ZERO && v >= -RS_ZERO;
        }
        return false;
    }
    case 2: // LpRmL (8.3/8.4)
    {
        double xi = x - sp, eta = y - 1. + cp, u1, theta;
        polar(xi, eta, u1, theta);
        if (u1 <= 4.)
        {
            u = -2. * hy_asin(.25 * u1);
            t = mod2pi(theta + .5 * u + RS_PI);
            v = mod2pi(phi - t + u);
            return t >= -RS_ZERO && u <= RS_ZERO;
        }
        return false;
    }
    case 3: // LpRupLumRm (8.7)
    {
        double xi = x + sp, eta = y - 1. - cp, rho = .25 * (2. + hy_sqrt(xi * xi + eta * eta));
        if (rho <= 1.)
        {
            u = hy_acos(rho);
            tau_omega(u, -u, xi, eta, phi, t, v);
            return t >= -RS_ZERO && v <= RS_ZERO;
        }
        return false;
    }
    case 4: // LpRumLumRp (8.8)
    {
        double xi = x + sp, eta = y - 1. - cp, rho = (20. - xi * xi - eta * eta) / 16.;
        if (rho >= 0 && rho <= 1)
        {
            u = -hy_acos(rho);
            if (u >= -.5 * RS_PI)
            {
                tau_omega(u, u, xi, eta, phi, t, v);
                return t >= -RS_ZERO && v >= -RS_ZERO;
            }
        }
        return false;
    }
    case 5: // LpRmSmLm (8.9)
    {
        double xi = x - sp, eta = y - 1. + cp, rho, theta;
        polar(xi, eta, rho, theta);
        if (rho >= 2.)
        {
            double r = hy_sqrt(rho * rho - 4.);
            u = 2. - r;
            t = mod2pi(theta + hy_atan2(r, -2.));
            v = mod2pi(phi - .5 * RS_PI - t);
            return t >= -RS_ZERO && u <= RS_ZERO && v <= RS_ZERO;
        }
        return false;
    }
    case 6: // LpRmSmRm (8.10)
    {
        double xi = x + sp, eta = y - 1. - cp, rho, theta;
        polar(-eta, xi, rho, theta);
        if (rho >= 2.)
        {
            t = theta;
            u = 2. - rho;
            v = mod2pi(t + .5 * RS_PI - phi);
            return t >= -RS_ZERO && u <= RS_ZERO && v <= RS_ZERO;
        }
        return false;
    }
    default: // LpRmSLmRp (8.11)
    {
        double xi = x + sp, eta = y - 1. - cp, rho, theta;
        polar(xi, eta, rho, theta);
        if (rho >= 2.)
        {
            u = 4. - hy_sqrt(rho * rho - 4.);
            if (u <= RS_ZERO)
            {
                t = mod2pi(hy_atan2((4 - u) * xi - 2 * eta, -2 * xi + (u - 4) * eta));
                v = mod2pi(t - phi);
                return t >= -RS_ZERO && v >= -RS_ZERO;
            }
        }
        return false;
    }
    }
}

RS_HD bool base_formula_body(int f, double x, double y, double phi, double &t, double &u, double &v)
{
    return base_formula_body_sc(f, x, y, phi, hy_sin(phi), hy_cos(phi), t, u, v);
}

RS_HDC TUV base_formula_call(int f, double x, double y, double phi)
{
    TUV r;
    r.t = r.u = r.v = 0;
    r.ok = base_formula_body(f, x, y, phi, r.t, r.u, r.v) ? 1 : 0;
    return r;
}
RS_HDC TUV base_formula_call_sc(int f, double x, double y, double phi, double sp, double cp)
{
    TUV r;
    r.t = r.u = r.v = 0;
    r.ok = base_formula_body_sc(f, x, y, phi, sp, cp, r.t, r.u, r.v) ? 1 : 0;
    return r;
}
RS_HD bool base_formula(int f, double x, double y, double phi, double &t, double &u, double &v)
{
    const TUV r = base_formula_call(f, x, y, phi);
    t = r.t, u = r.u, v = r.v;
    return r.ok != 0;
}

// Candidate list in reference order.  Groups of four share a base formula and
// run through the symmetries (x,y,phi), (-x,y,-phi) "timeflip", (x,-y,-phi)
// "reflect", (-x,-y,phi) both.  group: base formula, uses backwards frame?,
// family (0 CSC, 1 CCC, 2 CCCC, 3 CCSC, 4 CCSCC), word of the un-reflected
// pair (the reflected pair uses word+1), length layout.
#define RS_NUM_CANDIDATES 44
#define RS_NUM_FAMILIES 5
RS_HD void group_info(int g, int &formula, int &backwards, int &family, int &word, int &layout)
{
    //                    g:  0   1   2   3   4   5   6   7   8   9  10
    const signed char F[11] = {0, 1, 2, 2, 3, 4, 5, 6, 5, 6, 7};
    const signed char B[11] = {0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0};
    const signed char M[11] = {0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 4};
    const signed char W[11] = {14, 12, 0, 0, 2, 2, 4, 8, 6, 10, 16};
    // layout: 0 (t,u,v) 1 (v,u,t) 2 (t,u,-u,v) 3 (t,u,u,v) 4 (t,-pi/2,u,v) 5 (v,u,-pi/2,t) 6 (t,-pi/2,u,-pi/2,v)
    const signed char Lay[11] = {0, 0, 0, 1, 2, 3, 4, 4, 5, 5, 6};
    formula = F[g], backwards = B[g], family = M[g], word = W[g], layout = Lay[g];
}
RS_HD int family_first(int fam)
{
    const int first[6] = {0, 8, 16, 24, 40, 44};
    return first[fam];
}

// From (t, u, v) of the base formula to the path record: the length the family compares, the symmetry's signs,
// the word and the segment layout (:110-158, :177-227, :262-310, :347-447, :469-492).
RS_HD void candidate_tail(int family, int word, int layout, bool flip, bool reflect, double t, double u, double v, Path &p, double &L)
{
    L = (family == 2) ? hy_fabs(t) + 2. * hy_fabs(u) + hy_fabs(v) : hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
    const double hp = flip ? .5 * RS_PI : -.5 * RS_PI; // the fixed quarter turn of CCSC/CCSCC
    if (flip)
        t = -t, u = -u, v = -v;
    p.type = word + (reflect ? 1 : 0);
    double a = 0, b = 0, cc = 0, d = 0, e = 0;
    switch (layout)
    {
    case 0:
        a = t, b = u, cc = v;
        break;
    case 1:
        a = v, b = u, cc = t;
        break;
    case 2:
        a = t, b = u, cc = -u, d = v;
        break;
    case 3:
        a = t, b = u, cc = u, d = v;
        break;
    case 4:
        a = t, b = hp, cc = u, d = v;
        break;
    case 5:
        a = v, b = u, cc = hp, d = t;
        break;
    default:
        a = t, b = hp, cc = u, d = hp, e = v;
        break;
    }
    p.len[0] = a, p.len[1] = b, p.len[2] = cc, p.len[3] = d, p.len[4] = e;
    p.total = hy_fabs(a) + hy_fabs(b) + hy_fabs(cc) + hy_fabs(d) + hy_fabs(e); // :527-537
}


// Evaluates candidate c for the normalised goal (x, y, phi).  On success fills
// p (word, signed lengths, total) and L, the length the family compares
// (|t|+|u|+|v|, with 2|u| in the CCCC family).
RS_HD bool candidate(int c, double x, double y, double phi, Path &p, double &L)
{
    int formula, backwards, family, word, layout;
    group_info(c >> 2, formula, backwards, family, word, layout);
    const int sym = c & 3;
    double X = x, Y = y;
    if (backwards) // :205, :400
    {
        double cp = hy_cos(phi), sp = hy_sin(phi);
        X = x * cp + y * sp;
        Y = x * sp - y * cp;
    }
    const bool flip = (sym & 1) != 0, reflect = (sym & 2) != 0;
    const double ax = flip ? -X : X;
    const double ay = reflect ? -Y : Y;
    const double aphi = (flip != reflect) ? -phi : phi;
    double t, u, v;
    if (!base_formula(formula, ax, ay, aphi, t, u, v))
        return false;
    candidate_tail(family, word, layout, flip, reflect, t, u, v, p, L);
    return true;
}

// candidate() arranged for lanes that evaluate different candidates side by side: what differs between the closed
// forms is a few additions and comparisons, kept in small switches, while every costly function (sincos, sqrt,
// atan2, the fmod inside mod2pi) sits in code all candidates pass through with their own arguments.  Forty-four
// lanes of one wave then spend the time of the longest form instead of the sum of the forms they hold.  Every
// candidate performs exactly the operations of base_formula_body() on exactly the same values, so the results are
// equal bit for bit; arguments of a candidate already known to fail are replaced by harmless ones.
// (s0, c0): sine and cosine of phi; (sp, cp): of the candidate's own angle, -phi when exactly one of its two
// symmetries is on, else phi.
RS_HD bool candidate_staged_sc(int c, double x, double y, double phi, double s0, double c0, double sp, double cp, Path &p, double &L)
{
    int formula, backwards, family, word, layout;
    group_info(c >> 2, formula, backwards, family, word, layout);
    const int sym = c & 3;
    double X = x, Y = y;
    if (backwards) // :205, :400
    {
        X = x * c0 + y * s0;
        Y = x * s0 - y * c0;
    }
    const bool flip = (sym & 1) != 0, reflect = (sym & 2) != 0;
    const double ax = flip ? -X : X;
    const double ay = reflect ? -Y : Y;
    const double aphi = (flip != reflect) ? -phi : phi;

    // stage 1: the polar form every closed form starts from
    const bool plus = formula == 1 || formula == 3 || formula == 4 || formula == 6 || formula == 7;
    const double xi = plus ? ax + sp : ax - sp;
    const double eta = plus ? ay - 1. - cp : ay - 1. + cp;
    const double pa = (formula == 6) ? -eta : xi, pb = (formula == 6) ? xi : eta;
    const double R = hy_sqrt(pa * pa + pb * pb);
    const double TH = hy_atan2(pb, pa);

    // stage 2: one square root and one arc tangent
    bool ok = true;
    double sq_arg = 0., rho = 0., w = 0.;
    switch (formula)
    {
    case 0:
        ok = TH >= -RS_ZERO;
        break;
    case 1:
    {
        const double u1 = R * R;
        ok = u1 >= 4.;
        sq_arg = u1 - 4.;
        break;
    }
    case 2:
        ok = R <= 4.;
        w = .25 * R;
        sq_arg = (1.0 - w) * (1.0 + w);
        break;
    case 3:
        rho = .25 * (2. + R);
        ok = rho <= 1.;
        sq_arg = (1.0 - rho) * (1.0 + rho);
        break;
    case 4:
        rho = (20. - xi * xi - eta * eta) / 16.;
        ok = rho >= 0 && rho <= 1;
        sq_arg = (1.0 - rho) * (1.0 + rho);
        break;
    case 5:
    case 7:
        ok = R >= 2.;
        sq_arg = R * R - 4.;
        break;
    default: // 6
        ok = R >= 2.;
        break;
    }
    const double SQ = hy_sqrt(ok ? sq_arg : 0.);
    double at_y = 0., at_x = 1., u = 0.;
    switch (formula)
    {
    case 1:
        at_y = 2., at_x = SQ;
        break;
    case 2:
        at_y = w, at_x = SQ;
        break;
    case 3:
    case 4:
        at_y = SQ, at_x = rho;
        break;
    case 5:
        at_y = SQ, at_x = -2.;
        break;
    case 7:
        u = 4. - SQ;
        ok = ok && u <= RS_ZERO;
        at_y = (4 - u) * xi - 2 * eta, at_x = -2 * xi + (u - 4) * eta;
        break;
    default:
        break;
    }
    if (!ok)
        at_y = 0., at_x = 1.;
    const double AT = hy_atan2(at_y, at_x);

    // stage 3: the two four-arc forms
    double t = 0., v = 0.;
    if (formula == 3 || formula == 4)
    {
        u = (formula == 3) ? AT : -AT;
        if (formula == 4)
            ok = ok && u >= -.5 * RS_PI;
        if (ok)
        {
            tau_omega(u, (formula == 3) ? -u : u, xi, eta, aphi, t, v);
            ok = t >= -RS_ZERO && ((formula == 3) ? v <= RS_ZERO : v >= -RS_ZERO);
        }
    }
    else
    {
        // stage 4: up to two reductions to (-pi, pi]
        double m1 = 0.;
        switch (formula)
        {
        case 1:
            u = SQ;
            m1 = TH + AT;
            break;
        case 2:
            u = -2. * AT;
            m1 = TH + .5 * u + RS_PI;
            break;
        case 5:
            u = 2. - SQ;
            m1 = TH + AT;
            break;
        case 7:
            m1 = AT;
            break;
        case 0:
            u = R;
            break;
        default: // 6
            u = 2. - R;
            break;
        }
        const double M1 = mod2pi(ok ? m1 : 0.);
        t = (formula == 0 || formula == 6) ? TH : M1;
        double m2;
        switch (formula)
        {
        case 0:
            m2 = aphi - t;
            break;
        case 2:
            m2 = aphi - t + u;
            break;
        case 5:
            m2 = aphi - .5 * RS_PI - t;
            break;
        case 6:
            m2 = t + .5 * RS_PI - aphi;
            break;
        default: // 1, 7
            m2 = t - aphi;
            break;
        }
        v = mod2pi(ok ? m2 : 0.);
        switch (formula)
        {
        case 0:
            ok = ok && v >= -RS_ZERO;
            break;
        case 1:
        case 7:
            ok = ok && t >= -RS_ZERO && v >= -RS_ZERO;
            break;
        case 2:
            ok = ok && t >= -RS_ZERO && u <= RS_ZERO;
            break;
        default: // 5, 6
            ok = ok && t >= -RS_ZERO && u <= RS_ZERO && v <= RS_ZERO;
            break;
        }
    }
    if (!ok)
        return false;
    candidate_tail(family, word, layout, flip, reflect, t, u, v, p, L);
    return true;
}

RS_HD bool candidate_staged(int c, double x, double y, double phi, Path &p, double &L)
{
    const int sym = c & 3;
    double s0, c0, sp, cp;
    hy_sincos(phi, &s0, &c0);
    hy_sincos((((sym & 1) != 0) != ((sym & 2) != 0)) ? -phi : phi, &sp, &cp);
    return candidate_staged_sc(c, x, y, phi, s0, c0, sp, cp, p, L);
}

// Starting bound of each family's scan: path.length() minus the fixed quarter
// turns its comparison leaves out (:348, :470).
RS_HD double family_bound(int fam, double best_total)
{
    if (fam == 3)
        return best_total - .5 * RS_PI;
    if (fam == 4)
        return best_total - RS_PI;
    return best_total;
}

RS_HD Path empty_path() // include/ReedsShepp.h:65-67
{
    Path p;
    p.type = 0;
    p.len[0] = 1.7976931348623157e308;
    p.len[1] = p.len[2] = p.len[3] = p.len[4] = 0;
    p.total = 1.7976931348623157e308;
    return p;
}

// One thread walks all candidates in order (:494-503).
RS_HD Path solve_serial(double x, double y, double phi)
{
    Path best = empty_path();
    for (int fam = 0; fam < RS_NUM_FAMILIES; ++fam)
    {
        double lmin = family_bound(fam, best.total);
        for (int c = family_first(fam); c < family_first(fam + 1); ++c)
        {
            Path p;
            double L;
            if (candidate(c, x, y, phi, p, L) && lmin > L)
            {
                best = p;
                lmin = L;
            }
        }
    }
    return best;
}

// The same walk with the two sine/cosine pairs all 44 candidates use (of phi and of -phi) taken once: what the
// heuristic-table kernel runs, one goal per lane (29.7 -> 22.7 ms for 79 M entries).  Neighbouring table entries fail
// the same candidates at the same tests, so the plain form's early exits cost little there: the side-by-side form,
// which evaluates every candidate to the end, measured 32.5 ms in that kernel, and sharing the polar forms of
// (xi, eta) between candidates as well (twelve serve 28 candidates) 29.5 ms, since it needs the walk unrolled.
RS_HD Path solve_hoisted(double x, double y, double phi)
{
    const double s_pos = hy_sin(phi), c_pos = hy_cos(phi), s_neg = hy_sin(-phi), c_neg = hy_cos(-phi);
    const double xb = x * c_pos + y * s_pos, yb = x * s_pos - y * c_pos; // the backwards frame, :205, :400
    Path best = empty_path();
    for (int fam = 0; fam < RS_NUM_FAMILIES; ++fam)
    {
        double lmin = family_bound(fam, best.total);
        for (int c = family_first(fam); c < family_first(fam + 1); ++c)
        {
            int formula, backwards, family, word, layout;
            group_info(c >> 2, formula, backwards, family, word, layout);
            const bool flip = (c & 1) != 0, reflect = (c & 2) != 0, neg = flip != reflect;
            const double X = backwards ? xb : x, Y = backwards ? yb : y;
            const TUV r = base_formula_call_sc(formula, flip ? -X : X, reflect ? -Y : Y, neg ? -phi : phi, neg ? s_neg : s_pos, neg ? c_neg : c_pos);
            if (!r.ok)
                continue;
            Path p;
            double L;
            candidate_tail(family, word, layout, flip, reflect, r.t, r.u, r.v, p, L);
            if (lmin > L)
            {
                best = p;
                lmin = L;
            }
        }
    }
    return best;
}

// ---- length of the shortest path only: what the heuristic table stores ------------------------------------------
// The 44 candidates regrouped by symmetry instead of by family, so that what candidates of one symmetry share is
// formed once: five closed forms start from the polar form of (x - sin phi, y - 1 + cos phi) and six from that of
// (x + sin phi, y - 1 - cos phi), in the forward or the backwards frame, and two of them take the same second square
// root.  Every candidate that can be valid still performs the operations of base_formula_body_sc() and
// candidate_tail() on the same values in the same order, so its lengths are equal bit for bit; the families' scans
// are replaced by a per-family minimum that remembers the candidate's position in the reference order (first of
// equals wins, as a strict "shorter than" scan does), and the families are then combined in their order with the
// bounds of family_bound().
// A candidate is dropped before its arc tangents where the signs of their arguments already say that its first arc
// t comes out below -RS_ZERO, which the closed form rejects: atan2(y, x) has the sign of y, and for y < 0 it is
// below -RS_DEAD unless x > 0 and -y <= RS_DEAD x.  RS_DEAD is several times RS_ZERO plus the rounding of the sums
// involved, so that only candidates the closed form rejects are dropped.
#define RS_DEAD 1e-14

// v with its sign bit flipped where `mask` has it set
RS_HD double flip_sign(double v, unsigned long long mask)
{
    unsigned long long b;
    __builtin_memcpy(&b, &v, 8);
    b ^= mask;
    __builtin_memcpy(&v, &b, 8);
    return v;
}

// mod2pi() for sums of a few angles: fmod's long division by 2 pi written out for |x| < 8 pi.  Each subtraction
// is exact (the minuend lies within a factor two of the subtrahend), so the remainder is fmod's, bit for bit; the
// fold into [-pi, pi] works on the magnitude (a - b == -(b - a) in IEEE arithmetic) and the sign of x goes on last,
// zero included, as fmod's does.  Anything else (NaN included) sets `beyond`, and the caller falls back on the
// general form.
RS_HD double mod2pi_few_turns(double x, bool &beyond)
{
    const double T = 2. * RS_PI;
    double ax = hy_fabs(x);
    beyond = beyond || !(ax < 4. * T);
    if (ax >= 2. * T) // seldom: a branch of its own, which lanes skip together
    {
        ax = ax - 2. * T;
        HY_KEEP_IN_BRANCH(ax);
    }
    ax = (ax >= T) ? ax - T : ax;
    ax = (ax > RS_PI) ? ax - T : ax;
    unsigned long long xbits;
    __builtin_memcpy(&xbits, &x, 8);
    return flip_sign(ax, xbits & 0x8000000000000000ull);
}

// tau_omega(u, v, ...) for v == -u (same == false) or v == u (same == true), the two uses there are (:229-260):
// cos(v) is cos(u) either way (hy_cos is even bit for bit: the reduction and both kernels are), and for v == u
// delta is mod2pi(u - u) = 0 with sine 0 and cosine 1.
RS_HD void tau_omega_pm(double u, bool same, double xi, double eta, double phi, double &tau, double &omega, bool &beyond, const double *qa)
{
    double su, cu, sd = 0., cd = 1.;
    hy_sincos(u, &su, &cu);
    const double v = same ? u : -u;
    if (!same)
        hy_sincos(mod2pi_few_turns(u - v, beyond), &sd, &cd);
    const double A = su - sd, B = cu - cd - 1.;
    const double t1 = hy_atan2_q(eta * A - xi * B, xi * A + eta * B, qa), t2 = 2. * (cd - cu - cu) + 3;
    tau = mod2pi_few_turns((t2 < 0) ? t1 + RS_PI : t1, beyond);
    omega = mod2pi_few_turns(tau - u + v - phi, beyond);
}

struct FamilyBest
{
    double l, total; // the length the family compares, and the path's length
    int at;          // candidate number
};
RS_HD void offer(FamilyBest &b, bool ok, int c, double L, double total)
{
    const bool take = ok && (L < b.l || (L == b.l && c < b.at));
    b.l = take ? L : b.l;
    b.total = take ? total : b.total;
    b.at = take ? c : b.at;
}
// Where equal L means equal totals (CSC: the total is L itself) or the candidates arrive in reference order
// (CCSCC: one group, its symmetries in order), which of two equals is kept makes no difference.
RS_HD void offer_any_of_equals(FamilyBest &b, bool ok, double L, double total)
{
    const bool take = ok && L < b.l;
    b.l = take ? L : b.l;
    b.total = take ? total : b.total;
}
// atan2(y, x) is certainly below -RS_ZERO
RS_HD bool angle_is_negative(double y, double x) { return y < 0. && !(-y <= RS_DEAD * x); }
// Sums and differences of such angles are the angle of a product of complex numbers: atan2(ay, ax) + atan2(by, bx) is
// that of (ax + i ay)(bx + i by), minus an angle is the conjugate, and phi itself is (cos phi, sin phi).  Where the
// product's imaginary part is negative (positive) by more than 1e-13 of a bound m of its modulus -- the product of the
// factors' |re| + |im|, which times a few 1e-16 also bounds the rounding in it -- the true angle lies inside
// (-pi + 1e-13, -1e-13) (inside (1e-13, pi - 1e-13)), and the computed sum, a few 1e-16 away, is folded by mod2pi to
// the same side: a form that needs that angle to be >= -RS_ZERO (<= RS_ZERO) is certain to fail.
struct Product
{
    double re, im, m;
};
RS_HD Product product(double ar, double ai, double br, double bi)
{
    Product c;
    c.re = ar * br - ai * bi, c.im = ar * bi + ai * br, c.m = (hy_fabs(ar) + hy_fabs(ai)) * (hy_fabs(br) + hy_fabs(bi));
    return c;
}
RS_HD bool clearly_negative(double im, double m) { return -im > 1e-13 * m; }
RS_HD bool clearly_positive(double im, double m) { return im > 1e-13 * m; }

// The candidates of symmetry s in one frame, (X, Y) being the goal there with the symmetry's signs applied.
// Square roots are taken late: the sign tests first run on bounds of the roots they involve -- with
// R = |xi + i eta| and r = sqrt(R^2 - 4), max(|xi|, |eta|) - 2 <= R - 2 <= r <= R <= |xi| + |eta| -- and a root is only
// taken for a form those leave alive (the tests are linear in the root, so their extreme over the interval decides).
template <bool BACK>
RS_HD void offer_symmetry(int s, double ax, double ay, double aphi, double sp, double cp, FamilyBest *fam, bool &beyond, const double *qa)
{
    const double hp = .5 * RS_PI;
    {
        // forms on (x - sin phi, y - 1 + cos phi): (8.1) forward only, (8.3/8.4) and (8.9) in both frames
        const double xi = ax - sp, eta = ay - 1. + cp;
        const double s2 = xi * xi + eta * eta;
        const double hi = hy_fabs(xi) + hy_fabs(eta), big = hy_fabs(xi) > hy_fabs(eta) ? hy_fabs(xi) : hy_fabs(eta);
        const double lo = big > 2. ? big - 2. : 0.;
        const double reach = 1.5 * hi; // bounds |xi + i eta| (|cos phi| + |sin phi|)
        const double A = sp * xi - cp * eta, B = sp * eta + cp * xi;
        // (8.1): t = atan2(eta, xi) and v = mod2pi(phi - t), the angle of (cos phi + i sin phi)(xi - i eta), both >= 0
        const bool first = !BACK && !angle_is_negative(eta, xi) && !clearly_negative(A, reach);
        // (8.9): t = mod2pi(atan2(eta, xi) + atan2(r, -2)) >= 0, the angle of (xi + i eta)(-2 + i r) with imaginary part
        // xi r - 2 eta; v = mod2pi(phi - pi/2 - t), the angle of (sin phi - i cos phi) conj(that product) with imaginary
        // part 2 B - r A, <= 0
        const double m9 = hi * (2. + hi);
        bool third = s2 >= 3.5 && !clearly_negative((xi >= 0. ? xi * hi : xi * lo) - 2. * eta, m9) &&
                     !clearly_positive(2. * B - (A >= 0. ? A * hi : A * lo), 1.5 * m9);
        if (first || s2 <= 16.5 || third)
        {
            const double R = hy_sqrt(s2);
            const bool second = R <= 4.;
            double r = 0.;
            third = third && R >= 2.;
            if (third)
            {
                r = hy_sqrt(R * R - 4.);
                const Product p9 = product(xi, eta, -2., r);
                third = !clearly_negative(p9.im, p9.m) && !clearly_positive(-(sp * p9.im + cp * p9.re), 1.5 * p9.m);
            }
            if (first || second || third)
            {
                const double TH = hy_atan2_q(eta, xi, qa);
                if (first) // LpSpLp, group 0
                {
                    const double t = TH, u = R;
                    const double v = mod2pi_few_turns(aphi - t, beyond);
                    const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
                    offer_any_of_equals(fam[0], t >= -RS_ZERO && v >= -RS_ZERO, L, L);
                }
                if (second) // LpRmL, groups 2 and 3
                {
                    const double u = -2. * hy_asin_q(.25 * R, qa);
                    const double t = mod2pi_few_turns(TH + .5 * u + RS_PI, beyond);
                    const double v = mod2pi_few_turns(aphi - t + u, beyond);
                    const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
                    const double total = BACK ? hy_fabs(v) + hy_fabs(u) + hy_fabs(t) : L;
                    offer(fam[1], t >= -RS_ZERO && u <= RS_ZERO, (BACK ? 12 : 8) + s, L, total);
                }
                if (third) // LpRmSmLm, groups 6 and 8
                {
                    const double u = 2. - r;
                    const double t = mod2pi_few_turns(TH + hy_atan2_q(r, -2., qa), beyond);
                    const double v = mod2pi_few_turns(aphi - .5 * RS_PI - t, beyond);
                    const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
                    const double total = BACK ? hy_fabs(v) + hy_fabs(u) + hp + hy_fabs(t) : hy_fabs(t) + hp + hy_fabs(u) + hy_fabs(v);
                    offer(fam[3], t >= -RS_ZERO && u <= RS_ZERO && v <= RS_ZERO, (BACK ? 32 : 24) + s, L, total);
                }
            }
        }
    }
    {
        // forms on (x + sin phi, y - 1 - cos phi): (8.10) in both frames, the others forward only
        const double xi = ax + sp, eta = ay - 1. - cp;
        const double hi = hy_fabs(xi) + hy_fabs(eta);
        const double A = xi * sp - eta * cp;
        // (8.10): t = atan2(xi, -eta) >= 0; v = mod2pi(t + pi/2 - phi), the angle of (-xi - i eta)(cos phi - i sin phi), <= 0
        const bool fourth = !angle_is_negative(xi, -eta) && !clearly_positive(A, 1.5 * hi);
        if (BACK && !fourth)
            return;
        const double s2 = xi * xi + eta * eta;
        bool second = false, eleventh = false;
        if (!BACK)
        {
            const double big = hy_fabs(xi) > hy_fabs(eta) ? hy_fabs(xi) : hy_fabs(eta);
            const double lo = big > 2. ? big - 2. : 0., m = hi * (2. + hi), B = xi * cp + eta * sp;
            // (8.2), with u = sqrt(R^2 - 4): t is the angle of (xi + i eta)(u + 2 i), imaginary part 2 xi + eta u, >= 0;
            // v = mod2pi(t - phi), that of the product times (cos phi - i sin phi), imaginary part 2 B - u A, >= 0
            second = s2 >= 3.5 && !clearly_negative(2. * xi + (eta >= 0. ? eta * hi : eta * lo), m) &&
                     !clearly_negative(2. * B - (A >= 0. ? A * lo : A * hi), 1.5 * m);
            // (8.11), with the same root (and u = 4 - root <= 0): t is the angle of (xi + i eta)(-2 + i root) but for
            // roundings, imaginary part root xi - 2 eta, >= 0; v = mod2pi(t - phi), imaginary part root B + 2 A, >= 0
            eleventh = s2 >= 3.5 && hi >= 3.9 && !clearly_negative((xi >= 0. ? xi * hi : xi * lo) - 2. * eta, m) &&
                       !clearly_negative(2. * A + (B >= 0. ? B * hi : B * lo), 1.5 * m);
            if (!(fourth || second || eleventh || s2 <= 20.5))
                return;
        }
        const double R = hy_sqrt(s2);
        if (R >= 2. && fourth) // LpRmSmRm, groups 7 and 9
        {
            const double t = hy_atan2_q(xi, -eta, qa), u = 2. - R;
            const double v = mod2pi_few_turns(t + .5 * RS_PI - aphi, beyond);
            const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
            const double total = BACK ? hy_fabs(v) + hy_fabs(u) + hp + hy_fabs(t) : hy_fabs(t) + hp + hy_fabs(u) + hy_fabs(v);
            offer(fam[3], t >= -RS_ZERO && u <= RS_ZERO && v <= RS_ZERO, (BACK ? 36 : 28) + s, L, total);
        }
        if (BACK)
            return;
        const double u1 = R * R;
        double SQ = 0.;
        if ((second || eleventh) && (u1 >= 4. || R >= 2.))
            SQ = hy_sqrt(u1 - 4.); // shared by (8.2) and (8.11)
        // LpSpRp, group 1
        if (second && u1 >= 4.)
        {
            const Product p2 = product(xi, eta, SQ, 2.);
            if (!clearly_negative(p2.im, p2.m) && !clearly_negative(p2.im * cp - p2.re * sp, 1.5 * p2.m))
            {
                const double u = SQ;
                const double t = mod2pi_few_turns(hy_atan2_q(eta, xi, qa) + hy_atan2_q(2., u, qa), beyond);
                const double v = mod2pi_few_turns(t - aphi, beyond);
                const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
                offer_any_of_equals(fam[0], t >= -RS_ZERO && v >= -RS_ZERO, L, L);
            }
        }
        if (s2 <= 4.5) // (8.7) needs R <= 2
        {
            const double rho = .25 * (2. + R);
            if (rho <= 1.) // LpRupLumRm, group 4
            {
                const double u = hy_acos_q(rho, qa);
                double t, v;
                tau_omega_pm(u, false, xi, eta, aphi, t, v, beyond, qa);
                const double L = hy_fabs(t) + 2. * hy_fabs(u) + hy_fabs(v);
                offer(fam[2], t >= -RS_ZERO && v <= RS_ZERO, 16 + s, L, hy_fabs(t) + hy_fabs(u) + hy_fabs(u) + hy_fabs(v));
            }
        }
        if (s2 <= 20.5) // (8.8) needs 20 - xi^2 - eta^2 >= 0
        {
            const double rho = (20. - xi * xi - eta * eta) / 16.;
            if (rho >= 0 && rho <= 1) // LpRumLumRp, group 5
            {
                const double u = -hy_acos_q(rho, qa);
                if (u >= -.5 * RS_PI)
                {
                    double t, v;
                    tau_omega_pm(u, true, xi, eta, aphi, t, v, beyond, qa);
                    const double L = hy_fabs(t) + 2. * hy_fabs(u) + hy_fabs(v);
                    offer(fam[2], t >= -RS_ZERO && v >= -RS_ZERO, 20 + s, L, hy_fabs(t) + hy_fabs(u) + hy_fabs(u) + hy_fabs(v));
                }
            }
        }
        if (eleventh && R >= 2.) // LpRmSLmRp, group 10
        {
            const double u = 4. - SQ;
            if (u <= RS_ZERO)
            {
                // t = mod2pi(atan2(ty, tx)), and mod2pi leaves what atan2 returns as it is
                const double ty = (4 - u) * xi - 2 * eta, tx = -2 * xi + (u - 4) * eta;
                // ... >= 0; v = mod2pi(t - phi), the angle of (tx + i ty)(cos phi - i sin phi), >= 0
                if (!angle_is_negative(ty, tx) && !clearly_negative(ty * cp - tx * sp, 1.5 * (hy_fabs(tx) + hy_fabs(ty))))
                {
                    const double t = hy_atan2_q(ty, tx, qa);
                    const double v = mod2pi_few_turns(t - aphi, beyond);
                    const double L = hy_fabs(t) + hy_fabs(u) + hy_fabs(v);
                    offer_any_of_equals(fam[4], t >= -RS_ZERO && v >= -RS_ZERO, L, hy_fabs(t) + hp + hy_fabs(u) + hp + hy_fabs(v));
                }
            }
        }
    }
}

// rs::solve_serial(x, y, phi).total; quarter_atans: the caller's copy of HY_QUARTER_ATANS
RS_HD double shortest_length(double x, double y, double phi, const double *quarter_atans)
{
    // sine is odd and cosine even bit for bit (hy_rem_pio2 and both kernels are), so -phi needs no second pass
    double s_pos, c_pos;
    hy_sincos(phi, &s_pos, &c_pos);
    const double xb = x * c_pos + y * s_pos, yb = x * s_pos - y * c_pos; // the backwards frame, :205, :400
    bool beyond = false; // some angle sum was outside what mod2pi_few_turns() reduces
    FamilyBest fam[RS_NUM_FAMILIES];
    for (int f = 0; f < RS_NUM_FAMILIES; ++f)
        fam[f].l = __builtin_inf(), fam[f].total = 0., fam[f].at = RS_NUM_CANDIDATES;
#pragma unroll 1
    for (int s = 0; s < 4; ++s)
    {
        const unsigned long long SIGN = 0x8000000000000000ull;
        const unsigned long long flip = (s & 1) ? SIGN : 0, reflect = (s & 2) ? SIGN : 0, neg = flip ^ reflect;
        const double aphi = flip_sign(phi, neg), sp = flip_sign(s_pos, neg);
        offer_symmetry<false>(s, flip_sign(x, flip), flip_sign(y, reflect), aphi, sp, c_pos, fam, beyond, quarter_atans);
        offer_symmetry<true>(s, flip_sign(xb, flip), flip_sign(yb, reflect), aphi, sp, c_pos, fam, beyond, quarter_atans);
    }
    if (beyond) // never with finite goals whose heading difference is a few turns at most
        return solve_serial(x, y, phi).total;
    double best = 1.7976931348623157e308; // empty_path().total
    for (int f = 0; f < RS_NUM_FAMILIES; ++f)
        if (fam[f].l < family_bound(f, best))
            best = fam[f].total;
    return best;
}

// Goal in the frame of q0, scaled by the turning radius (:544-550).
RS_HD void normalise(const double q0[3], const double q1[3], double rho, double &x, double &y, double &phi)
{
    double dx = q1[0] - q0[0], dy = q1[1] - q0[1];
    phi = q1[2] - q0[2];
    double c, s;
    hy_sincos(q0[2], &s, &c);
    x = (c * dx + s * dy) / rho;
    y = (-s * dx + c * dy) / rho;
}

// State at arc length seg (units of rho) along the path from q0 (:576-625).
RS_HDC void interpolate(const double q0[3], const Path &path, double seg, double rho, double s[3])
{
    if (seg < 0.0)
        seg = 0.0;
    if (seg > path.total)
        seg = path.total;
    double phi, v;
    s[0] = s[1] = 0.0;
    s[2] = q0[2];
    for (int i = 0; i < 5 && seg > 0; ++i)
    {
        const double li = path.len[i];
        if (li < 0)
        {
            v = (-seg < li) ? li : -seg; // std::max(-seg, li)
            seg += v;
        }
        else
        {
            v = (li < seg) ? li : seg; // std::min(seg, li)
            seg -= v;
        }
        phi = s[2];
        const int kind = seg_kind(path.type, i);
        if (kind == 1)
        {
            s[0] += (hy_sin(phi + v) - hy_sin(phi));
            s[1] += (-hy_cos(phi + v) + hy_cos(phi));
            s[2] = phi + v;
        }
        else if (kind == 3)
        {
            s[0] += (-hy_sin(phi - v) + hy_sin(phi));
            s[1] += (hy_cos(phi - v) - hy_cos(phi));
            s[2] = phi - v;
        }
        else if (kind == 2)
        {
            s[0] += (v * hy_cos(phi));
            s[1] += (v * hy_sin(phi));
        }
    }
    s[0] = s[0] * rho + q0[0];
    s[1] = s[1] * rho + q0[1];
}

// ---- sampling many points of one path ------------------------------------
// interpolate() walks the segments in order and adds each one's increment to
// (x, y, phi).  For a point inside segment j the increments of segments 0..j-1
// are those of the complete segments, identical for every such point, so they
// are accumulated once (same operations, same order) and a sample only adds
// its own partial increment: two sin/cos instead of up to twenty.
struct Prefix
{
    double x[6], y[6], phi[6]; // state in front of segment i (units of rho, origin at q0)
    double s[5], c[5];         // sin(phi[i]), cos(phi[i])
};

RS_HDC void build_prefix(const Path &path, double phi0, Prefix &P)
{
    P.x[0] = 0.0, P.y[0] = 0.0, P.phi[0] = phi0;
    for (int i = 0; i < 5; ++i)
    {
        const double phi = P.phi[i], v = path.len[i];
        const double sp = hy_sin(phi), cp = hy_cos(phi);
        P.s[i] = sp, P.c[i] = cp;
        double x = P.x[i], y = P.y[i], nphi = phi;
        const int kind = seg_kind(path.type, i);
        if (kind == 1)
        {
            x += (hy_sin(phi + v) - sp);
            y += (-hy_cos(phi + v) + cp);
            nphi = phi + v;
        }
        else if (kind == 3)
        {
            x += (-hy_sin(phi - v) + sp);
            y += (hy_cos(phi - v) - cp);
            nphi = phi - v;
        }
        else if (kind == 2)
        {
            x += (v * cp);
            y += (v * sp);
        }
        P.x[i + 1] = x, P.y[i + 1] = y, P.phi[i + 1] = nphi;
    }
}

// Same result as interpolate(q0, path, seg, rho, s), bit for bit.
RS_HD void interpolate_prefix(const double q0[3], const Path &path, const Prefix &P, double seg, double rho, double s[3])
{
    if (seg < 0.0)
        seg = 0.0;
    if (seg > path.total)
        seg = path.total;
    double x = 0.0, y = 0.0, phi = q0[2];
    int part = -1; // the segment the point lies inside, if any (always the last one the walk reaches)
    double v_part = 0.0;
    for (int i = 0; i < 5 && seg > 0; ++i)
    {
        const double li = path.len[i];
        double v;
        if (li < 0)
        {
            v = (-seg < li) ? li : -seg;
            seg += v;
        }
        else
        {
            v = (li < seg) ? li : seg;
            seg -= v;
        }
        if (v == li)
        { // the whole segment: its increment is the shared one
            x = P.x[i + 1], y = P.y[i + 1], phi = P.phi[i + 1];
            continue;
        }
        part = i, v_part = v;
    }
    if (part >= 0)
    {
        // one sine/cosine pair per point, outside the walk: lanes that stop in different segments share this code
        const int i = part;
        const double v = v_part, p0 = P.phi[i];
        const int kind = seg_kind(path.type, i);
        if (kind == 1 || kind == 3)
        {
            const double a = (kind == 1) ? p0 + v : p0 - v;
            double sa, ca;
            hy_sincos(a, &sa, &ca);
            if (kind == 1)
            {
                x = P.x[i] + (sa - P.s[i]);
                y = P.y[i] + (-ca + P.c[i]);
            }
            else
            {
                x = P.x[i] + (-sa + P.s[i]);
                y = P.y[i] + (ca - P.c[i]);
            }
            phi = a;
        }
        else if (kind == 2)
        {
            x = P.x[i] + (v * P.c[i]);
            y = P.y[i] + (v * P.s[i]);
            phi = p0;
        }
        else
            x = P.x[i], y = P.y[i], phi = p0;
    }
    s[0] = x * rho + q0[0];
    s[1] = y * rho + q0[1];
    s[2] = phi;
}

} // namespace rs
} // namespace hyas
