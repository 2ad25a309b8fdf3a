// mxv_exact.hpp — correctly rounded sin / cos and the Acrobot step evaluated with them: the COLD path behind the termination test.
//
// Why it exists.  AcrobotEnv._terminal (gym/envs/classic_control/acrobot.py:232-235) compares -cos(t1) - cos(t2 + t1) with 1.0.  The
// hot path (mxv_device.hpp) takes eight 1.5-ulp sincos per RK4 step and rebuilds the shifted / summed cosines by angle addition: its
// post-step angles and its height agree with the reference's to a few fp64 ulps — not to the bit, so a height within a few ulps of 1.0
// could land on the other side of the threshold (measured on tests/golden/Acrobot_p1_threshold.npz, 4096 states bisected onto the
// threshold: 37 masks differ, 32 of them because of the terminal cosines, 5 because the post-step ANGLES differ in the last bit;
// profiles/r4/r4a_acrobot_threshold_flip_split.jsonl).  The reference's own libm (glibc 2.35 / NumPy's scalar loops) returns the correctly
// rounded value for all but ~1 in 10^3..10^4 arguments, so: whenever the hot path's height lies within 2^-40 of the threshold — a band
// ~4000x wider than anything the hot path's error can bridge, hit by ~1 step in 10^12 — the WHOLE step is evaluated again here the way
// the reference writes it (every cosine taken directly of the reference's own rounded argument, no angle addition, IEEE `/`), with
// sin / cos correctly rounded; state, observation and mask of that env-step are then the reference's wherever its libm is correctly
// rounded.  Cost on the hot path: one compare + one wave-uniform branch per step.
//
// cr_sincos: x -> (RN(sin x), RN(cos x)) for |x| < 2^19.  k = rint(x 2/pi); r = x - k pi/2 as a double-double against pi/2 split into
// 33 + 33 + 33 + 106 bits (the first three products are exact, so cancellation near multiples of pi/2 costs nothing: |error| of
// r < 2^-150 |k|); sin r and cos r on |r| <= pi/4 by their Taylor series to r^29 / r^28 in double-double Horner form (truncation
// < 2^-112, arithmetic ~2^-100 relative); the high word of the normalised result is the rounded value unless the exact value lies within
// ~2^-100 (relative) of a rounding boundary — probability ~2^-47 per call INSIDE a branch taken once in 10^12 steps.  Checked against
// 400-bit mpmath on the CPU (tests/test_exact_trig.py: this very header compiled by g++).
//
// The header is plain C++ (FMA through __builtin_fma): the including translation unit defines MXV_XFN (function qualifiers) and
// MXV_XCONST (qualifiers of the constant tables) and MXV_XCOLD (qualifiers of the three non-inlined entry points) — `__device__ ...` in
// mxv_device.hpp, nothing special in the host-side test shim.
#pragma once
#include <stdint.h>

#ifndef MXV_XFN
#define MXV_XFN static inline
#endif
#ifndef MXV_XCONST
#define MXV_XCONST static const
#endif
#ifndef MXV_XCOLD   // the three entry points below are never inlined on the device: each keeps its own (small) register footprint, so
#define MXV_XCOLD static   // a kernel that calls into this file once in 10^12 steps keeps the register budget of its K-step loop
#endif

namespace mxv {
namespace exact {

struct DD {
    double hi, lo;
};

MXV_XFN DD two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}
MXV_XFN DD fast_two_sum(double a, double b) {  // |a| >= |b| or a == 0
    const double s = a + b;
    return DD{s, b - (s - a)};
}
MXV_XFN DD two_prod(double a, double b) {
    const double p = a * b;
    return DD{p, __builtin_fma(a, b, -p)};
}
MXV_XFN DD dd_add(DD a, DD b) {  // accurate double-double sum (Dekker / Knuth; error < 3 * 2^-106 relative)
    DD s = two_sum(a.hi, b.hi);
    const DD t = two_sum(a.lo, b.lo);
    s = fast_two_sum(s.hi, s.lo + t.hi);
    return fast_two_sum(s.hi, s.lo + t.lo);
}
MXV_XFN DD dd_add_d(DD a, double b) {
    DD s = two_sum(a.hi, b);
    return fast_two_sum(s.hi, s.lo + a.lo);
}
MXV_XFN DD dd_mul(DD a, DD b) {  // error < ~5 * 2^-106 relative
    DD p = two_prod(a.hi, b.hi);
    p.lo = __builtin_fma(a.hi, b.lo, p.lo);
    p.lo = __builtin_fma(a.lo, b.hi, p.lo);
    return fast_two_sum(p.hi, p.lo);
}

// (-1)^k / (2k+1)!  and  (-1)^k / (2k)!,  k = 1..14, as double-doubles (400-bit mpmath, tools/gen_exact_trig_tables.py)
MXV_XCONST double kSinC[14][2] = {
    {-0x1.5555555555555p-3, -0x1.5555555555555p-57},  {0x1.1111111111111p-7, 0x1.1111111111111p-63},
    {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73}, {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
    {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80},  {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
    {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97}, {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
    {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112}, {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
    {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130}, {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
    {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149}, {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157}};
MXV_XCONST double kCosC[14][2] = {
    {-0x1.0000000000000p-1, 0x0.0p+0},                {0x1.5555555555555p-5, 0x1.5555555555555p-59},
    {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65},  {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
    {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76}, {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
    {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92}, {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
    {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107}, {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
    {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124}, {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
    {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143}, {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153}};

// x - k pi/2 as a double-double; k = rint(x 2/pi), |k| < 2^20
MXV_XFN DD reduce_pio2(double x, double k) {
    const double t = __builtin_fma(-k, 0x1.921fb54400000p+0, x);  // pi/2, bits 1..33: the product and the difference are exact
    DD r = two_sum(t, -k * 0x1.0b4611a600000p-34);                // bits 34..66 (exact product)
    r = dd_add_d(r, -k * 0x1.3198a2e000000p-69);                  // bits 67..99 (exact product)
    DD p = two_prod(k, 0x1.b839a252049c1p-104);                   // the next 106 bits
    p.lo = __builtin_fma(k, 0x1.14cf98e804178p-160, p.lo);
    return dd_add(r, DD{-p.hi, -p.lo});
}

// (RN(sin x), RN(cos x)), |x| < 2^19 (see the header comment for "RN")
MXV_XCOLD void cr_sincos(double x, double *sn, double *cs) {
    const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);  // 2/pi
    const DD r = reduce_pio2(x, k);
    const DD z = dd_mul(r, r);
    DD as{kSinC[13][0], kSinC[13][1]}, ac{kCosC[13][0], kCosC[13][1]};
#pragma nounroll
    for (int i = 12; i >= 0; --i) {
        as = dd_add(dd_mul(as, z), DD{kSinC[i][0], kSinC[i][1]});
        ac = dd_add(dd_mul(ac, z), DD{kCosC[i][0], kCosC[i][1]});
    }
    const DD s = dd_add(r, dd_mul(r, dd_mul(z, as)));  // r + r z (S1 + z (S2 + ...))
    const DD c = dd_add_d(dd_mul(z, ac), 1.0);         // 1 + z (C1 + z (C2 + ...))
    const int q = (int)k;
    double ss = (q & 1) ? c.hi : s.hi, cc = (q & 1) ? s.hi : c.hi;
    if (q & 2) ss = -ss;        // sin changes sign in quadrants 2, 3
    if ((q + 1) & 2) cc = -cc;  // cos in quadrants 1, 2
    *sn = ss;
    *cs = cc;
}
MXV_XFN double cr_cos(double x) {
    double s, c;
    cr_sincos(x, &s, &c);
    return c;
}

// AcrobotEnv.step's arithmetic (acrobot.py:196-223, _dsdt 237-277, rk4 418-465, wrap 378-396, bound 399-415, _terminal 232-235) as the
// reference writes it, on correctly rounded sin / cos.  P = the engine's parameter vector (include/mxv.h, MXV_ACROBOT: dt, l1, l2, m1,
// m2, lc1, lc2, I, max_vel_1, max_vel_2, torque_noise_max, nips).  s[4] in/out; sc[4] out = sin t1, cos t1, sin t2, cos t2 of the new
// state (the observation, and the hot path's carried values); returns terminated.
MXV_XFN void acrobot_dsdt_exact(const double *P, const double *sa, double a, double *out) {
    const double m1 = P[3], m2 = P[4], l1 = P[1], lc1 = P[5], lc2 = P[6], I1 = P[7], I2 = P[7];
    const double g = 9.8, pi = 3.141592653589793;
    const double theta1 = sa[0], theta2 = sa[1], dtheta1 = sa[2], dtheta2 = sa[3];
    double s2, c2;
    cr_sincos(theta2, &s2, &c2);
    const double d1 = m1 * (lc1 * lc1) + m2 * ((l1 * l1) + (lc2 * lc2) + 2 * l1 * lc2 * c2) + I1 + I2;  // :252-257
    const double d2 = m2 * ((lc2 * lc2) + l1 * lc2 * c2) + I2;                                          // :258
    const double phi2 = m2 * lc2 * g * cr_cos(theta1 + theta2 - pi / 2.0);                              // :259
    const double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2 +
                        (m1 * lc1 + m2 * l1) * g * cr_cos(theta1 - pi / 2) + phi2;                      // :260-265
    double ddtheta2;
    if (P[11] != 0.0)  // "nips" :266-269
        ddtheta2 = (a + d2 / d1 * phi1 - phi2) / (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
    else               // "book" :270-275
        ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2) / (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
    const double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;  // :276
    out[0] = dtheta1;
    out[1] = dtheta2;
    out[2] = ddtheta1;
    out[3] = ddtheta2;
}

MXV_XFN bool acrobot_step_exact(const double *P, double *s, double torque, double *sc) {
    const double pi = 3.141592653589793;
    const double dt = P[0] - 0, dt2 = dt / 2.0, dt6 = dt / 6.0;  // :210, 449-450, 463
    const double y0[4] = {s[0], s[1], s[2], s[3]};
    // k1 ... k4 (:453-456) accumulated as the reference's k1 + 2 k2 + 2 k3 + k4 is associated (:463)
    double k[4], y[4], acc[4];
    acrobot_dsdt_exact(P, y0, torque, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = k[i]; y[i] = y0[i] + dt2 * k[i]; }
    acrobot_dsdt_exact(P, y, torque, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = acc[i] + 2 * k[i]; y[i] = y0[i] + dt2 * k[i]; }
    acrobot_dsdt_exact(P, y, torque, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = acc[i] + 2 * k[i]; y[i] = y0[i] + dt * k[i]; }
    acrobot_dsdt_exact(P, y, torque, k);
    double ns[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ns[i] = y0[i] + dt6 * (acc[i] + k[i]);
    const double diff = pi - (-pi);  // wrap(x, -pi, pi) :378-396
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        double x = ns[i];
        while (x > pi) x = x - diff;
        while (x < -pi) x = x + diff;
        s[i] = x;
    }
    s[2] = __builtin_fmin(__builtin_fmax(ns[2], -P[8]), P[8]);  // bound :399-415 (no NaN can reach this path: the caller's compare was true)
    s[3] = __builtin_fmin(__builtin_fmax(ns[3], -P[9]), P[9]);
    cr_sincos(s[0], &sc[0], &sc[1]);
    cr_sincos(s[1], &sc[2], &sc[3]);
    return (-sc[1] - cr_cos(s[1] + s[0])) > 1.0;  // :235
}

}  // namespace exact
}  // namespace mxv
