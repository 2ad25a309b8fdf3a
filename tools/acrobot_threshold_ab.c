// Host emulation of the HOT path of Env<MXV_ACROBOT>::step (default parameters; gym_amd/csrc/mxv_device.hpp: same constants, same
// FMA arithmetic) with switches for where the trigonometry comes from — the CPU half of tools/acrobot_threshold_ab.py:
//   set_mode(trig, addition, term): trig 0 = the engine's medium-range sincos, 1 = glibc; addition 1 = shifted / summed cosines by
//   angle addition (the engine), 0 = evaluated directly on the reference's rounded arguments; term 0 = terminal cosines like the
//   stages, 1 = directly with the engine's sincos, 2 = directly with glibc.
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
static const double kPi = 3.141592653589793;
static void kern(double x, double *sn, double *cs) {
    const double z = x * x;
    double r = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    r = fma(z, r, 2.75573137070700676789e-06);
    r = fma(z, r, -1.98412698298579493134e-04);
    r = fma(z, r, 8.33333333332248946124e-03);
    r = fma(z, r, -1.66666666666666324348e-01);
    *sn = fma(x * z, r, x);
    double c = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    c = fma(z, c, -2.75573143513906633035e-07);
    c = fma(z, c, 2.48015872894767294178e-05);
    c = fma(z, c, -1.38888888888741095749e-03);
    c = fma(z, c, 4.16666666666666019037e-02);
#ifdef FDLIBM_COS   /* the cosine of the build profiles/r4/r4a_acrobot_threshold_flip_split.jsonl was recorded with (MXV_FDLIBM_COS = 1) */
    const double hz = 0.5 * z;
    const double t = 1.0 - hz;
    *cs = t + fma(z, z * c, (1.0 - t) - hz);
#else
    c = fma(z, c, -0.5);
    *cs = fma(z, c, 1.0);
#endif
}
static void own_sincos(double x, double *sn, double *cs) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.5707963267341256, x);
    r = fma(-k, 6.077100506303966e-11, r);
    r = fma(-k, 2.0222662487959506e-21, r);
    double s, c;
    kern(r, &s, &c);
    const int q = (int)k;
    double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *sn = ss; *cs = cc;
}
static int g_trig = 0;      // 0 = own sincos, 1 = glibc
static int g_addition = 1;  // 1 = angle addition for shifted cosines (device), 0 = direct evaluation of the reference's arguments
static int g_carry = 1;     // angle addition: 1 = carries the reference's argument roundings (TwoSum residuals), 0 = cos(t - pi/2) = sin t outright
static int g_term = 0;      // terminal test: 0 = like the stages (g_addition/g_trig), 1 = direct own, 2 = direct glibc
static void sc_(double x, double *s, double *c) { if (g_trig) sincos(x, s, c); else own_sincos(x, s, c); }
static double cos_(double x) { double s, c; sc_(x, &s, &c); return c; }
static double two_sum_residual(double a, double b, double sum) { const double bb = sum - a; return (a - (sum - bb)) + (b - bb); }
static const double kHalfPiTail = 6.123233995736766036e-17;
static void dsdt(const double *sa, const double *sc, double a, double *out) {
    const double m1 = 1, m2 = 1, l1 = 1, lc1 = 0.5, lc2 = 0.5, I1 = 1, I2 = 1, g = 9.8;
    const double theta1 = sa[0], theta2 = sa[1], dtheta1 = sa[2], dtheta2 = sa[3];
    const double s1 = sc[0], c1 = sc[1], s2 = sc[2], c2 = sc[3];
    const double halfpi = kPi / 2.0;
    double cos_t12_shift, cos_t1_shift;
    if (g_addition) {
        const double t12 = theta1 + theta2;
        const double a12 = t12 - halfpi;
        const double S12 = fma(s1, c2, c1 * s2), C12 = fma(c1, c2, -(s1 * s2));
        if (g_carry) {
            const double eps12 = kHalfPiTail - two_sum_residual(theta1, theta2, t12) - two_sum_residual(t12, -halfpi, a12);
            cos_t12_shift = fma(eps12, C12, S12);
            const double a1 = theta1 - halfpi;
            cos_t1_shift = fma(kHalfPiTail - two_sum_residual(theta1, -halfpi, a1), c1, s1);
        } else {
            cos_t12_shift = S12;
            cos_t1_shift = s1;
        }
    } else {
        cos_t12_shift = cos_(theta1 + theta2 - halfpi);
        cos_t1_shift = cos_(theta1 - halfpi);
    }
    const double d1 = m1 * (lc1 * lc1) + m2 * ((l1 * l1) + (lc2 * lc2) + 2 * l1 * lc2 * c2) + I1 + I2;
    const double d2 = m2 * ((lc2 * lc2) + l1 * lc2 * c2) + I2;
    const double phi2 = m2 * lc2 * g * cos_t12_shift;
    const double phi1 = -m2 * l1 * lc2 * (dtheta2 * dtheta2) * s2 - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * s2 + (m1 * lc1 + m2 * l1) * g * cos_t1_shift + phi2;
    const double ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * (dtheta1 * dtheta1) * s2 - phi2) / (m2 * (lc2 * lc2) + I2 - (d2 * d2) / d1);
    const double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;
    out[0] = dtheta1; out[1] = dtheta2; out[2] = ddtheta1; out[3] = ddtheta2;
}
static double wrap(double x, double m, double M) { const double diff = M - m; while (x > M) x = x - diff; while (x < m) x = x + diff; return x; }
static double bound(double x, double m, double M) { return fmin(fmax(x, m), M); }
void set_mode(int trig, int addition, int term) { g_trig = trig; g_addition = addition; g_term = term; }
void set_carry(int carry) { g_carry = carry; }
// s[4] in/out; returns terminated; height out
int acro_step(double *s, int ai, double *height) {
    const double torque = (double)(ai - 1), dt = 0.2, dt2 = dt / 2.0;
    const double y0[4] = {s[0], s[1], s[2], s[3]};
    double k1[4], k2[4], k3[4], k4[4], y[4], sc[4];
    sc_(y0[0], &sc[0], &sc[1]); sc_(y0[1], &sc[2], &sc[3]);
    dsdt(y0, sc, torque, k1);
    for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt2 * k1[k];
    sc_(y[0], &sc[0], &sc[1]); sc_(y[1], &sc[2], &sc[3]);
    dsdt(y, sc, torque, k2);
    for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt2 * k2[k];
    sc_(y[0], &sc[0], &sc[1]); sc_(y[1], &sc[2], &sc[3]);
    dsdt(y, sc, torque, k3);
    for (int k = 0; k < 4; ++k) y[k] = y0[k] + dt * k3[k];
    sc_(y[0], &sc[0], &sc[1]); sc_(y[1], &sc[2], &sc[3]);
    dsdt(y, sc, torque, k4);
    const double dt6 = dt / 6.0;
    double ns[4];
    for (int k = 0; k < 4; ++k) ns[k] = y0[k] + dt6 * (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]);
    s[0] = wrap(ns[0], -kPi, kPi); s[1] = wrap(ns[1], -kPi, kPi);
    s[2] = bound(ns[2], -4 * kPi, 4 * kPi); s[3] = bound(ns[3], -9 * kPi, 9 * kPi);
    double s0, c0, s1, c1, cos21;
    const double t21 = s[1] + s[0];
    if (g_term == 0) {
        sc_(s[0], &s0, &c0); sc_(s[1], &s1, &c1);
        if (g_addition) cos21 = g_carry ? fma(two_sum_residual(s[1], s[0], t21), fma(s0, c1, c0 * s1), fma(c0, c1, -(s0 * s1))) : fma(c0, c1, -(s0 * s1));
        else cos21 = cos_(t21);
    } else if (g_term == 1) {
        own_sincos(s[0], &s0, &c0); double d; own_sincos(t21, &d, &cos21);
    } else {
        c0 = cos(s[0]); cos21 = cos(t21);
    }
    *height = -c0 - cos21;
    return (-c0 - cos21) > 1.0;
}
void acro_batch(int n, double *states /*[n][4]*/, const int64_t *actions, uint8_t *term, double *height) {
    for (int i = 0; i < n; ++i) term[i] = (uint8_t)acro_step(states + 4 * i, (int)actions[i], height + i);
}
