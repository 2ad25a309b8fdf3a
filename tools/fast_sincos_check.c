// CPU check of gym_amd/csrc/mxv_device.hpp: sincos_medium (same constants, same FMA arithmetic) against 80-bit sinl/cosl:
//   gcc -O2 -mfma -o /tmp/fast tools/fast_sincos_check.c -lm && /tmp/fast    ->  max ulp err sin 1.466 cos 1.498 over |x| <= 40
//   (the same maxima with -DFDLIBM_COS, the compensated cosine sum of rounds 1-3: the argument reduction bounds both)
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static const double TWO_OVER_PI = 6.36619772367581382433e-01;
static const double PIO2_1  = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
static const double PIO2_2 = 6.077100506303966e-11, PIO2_2T = 2.0222662487959506e-21;
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
#ifdef FDLIBM_COS   /* the compensated sum of rounds 1-3 (MXV_FDLIBM_COS = 1) */
    const double hz = 0.5 * z;
    const double t = 1.0 - hz;
    *cs = t + fma(z, z * c, (1.0 - t) - hz);
#else               /* round 4: plain Horner, two FMAs */
    c = fma(z, c, -0.5);
    *cs = fma(z, c, 1.0);
#endif
}
static void fast_sincos(double x, double *sn, double *cs) {
    const double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_1, x);
    r = fma(-k, PIO2_2, r);
    r = fma(-k, PIO2_2T, r);
    double s, c;
    kern(r, &s, &c);
    const int q = (int)k;
    double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *sn = ss; *cs = cc;
}
static double ulp_err(double got, long double want) {
    double w = (double)want;
    double u = nextafter(fabs(w), INFINITY) - fabs(w);
    return (double)fabsl((long double)got - want) / u;
}
int main() {
    srand48(1);
    double ms = 0, mc = 0; double as=0, ac=0; long n = 20000000;
    for (long i = 0; i < n; ++i) {
        double x = (drand48() * 2 - 1) * 40.0;   /* |x| <= 40: far beyond Acrobot's stage angles */
        if (i % 7 == 0) { int k = (int)(drand48() * 16) - 8; x = k * 1.5707963267948966 + (drand48() - 0.5) * 1e-6; }
        double s, c; fast_sincos(x, &s, &c);
        long double ws = sinl((long double)x), wc = cosl((long double)x);
        double es = ulp_err(s, ws), ec = ulp_err(c, wc);
        if (es > ms) ms = es; if (ec > mc) mc = ec;
        double aes=fabs((double)((long double)s-ws)), aec=fabs((double)((long double)c-wc)); if(aes>as)as=aes; if(aec>ac)ac=aec;
    }
    printf("max ulp err sin %.3f cos %.3f ; max abs err sin %.3e cos %.3e\n", ms, mc, as, ac);
    /* glibc for comparison */
    ms = mc = 0;
    for (long i = 0; i < 2000000; ++i) { double x = (drand48()*2-1)*40.0; double s,c; sincos(x,&s,&c);
        double es = ulp_err(s, sinl((long double)x)), ec = ulp_err(c, cosl((long double)x)); if (es>ms) ms=es; if (ec>mc) mc=ec; }
    printf("glibc max ulp err sin %.3f cos %.3f\n", ms, mc);
    return 0;
}
