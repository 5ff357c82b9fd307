// Accuracy of ilqr_sincos_fast / ilqr_sincos_sum (ilqr_amd/csrc/fast_trig.h) measured on the host against 80-bit
// sinl/cosl (development aid).   g++ -O2 -ffp-contract=off -o /tmp/trig_accuracy tools/trig_accuracy.cpp -lm
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#define __device__
#define __forceinline__ inline
static inline double __builtin_amdgcn_rcp(double d) { return (double)(float)(1.0 / d); }   // low-precision seed like v_rcp_f64
#include "../ilqr_amd/csrc/fast_trig.h"

static double ulp_err(double got, long double want)
{
    if (want == 0) return got == 0 ? 0 : INFINITY;
    int e;
    frexpl(want, &e);
    const long double ulp = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    std::mt19937_64 g(12345);
    const double ranges[] = {0.8, 3.2, 100.0, 5e5, 9.99e5};
    for (double R : ranges) {
        std::uniform_real_distribution<double> U(-R, R);
        double ms = 0, mc = 0; long ds = 0, dc = 0;
        for (long i = 0; i < n; ++i) {
            const double x = U(g);
            double s, c;
            ilqr_sincos_fast(x, s, c);
            const double es = ulp_err(s, sinl((long double)x)), ec = ulp_err(c, cosl((long double)x));
            if (es > ms) ms = es;
            if (ec > mc) mc = ec;
            ds += s != sin(x); dc += c != cos(x);
        }
        printf("|x| < %-8g  max err sin %.3f ulp  cos %.3f ulp   differs from libm: sin %.2f %%  cos %.2f %%\n", R, ms, mc, 100.0 * ds / n, 100.0 * dc / n);
    }
    // near multiples of pi/2 (heavy cancellation in the reduction)
    {
        double ms = 0, mc = 0;
        std::uniform_real_distribution<double> U(-1e-6, 1e-6);
        for (long i = 0; i < n / 4; ++i) {
            const long k = (long)(g() % 200001) - 100000;
            const double x = (double)(k * M_PI_2l) + U(g) * (g() % 2 ? 1.0 : 1e-6);
            double s, c;
            ilqr_sincos_fast(x, s, c);
            const double es = ulp_err(s, sinl((long double)x)), ec = ulp_err(c, cosl((long double)x));
            if (es > ms) ms = es;
            if (ec > mc) mc = ec;
        }
        printf("near k*pi/2      max err sin %.3f ulp  cos %.3f ulp\n", ms, mc);
    }
    // angle addition: absolute error
    {
        double ma = 0;
        std::uniform_real_distribution<double> U(-50.0, 50.0);
        for (long i = 0; i < n / 4; ++i) {
            const double a = U(g), b = U(g);
            double sa, ca, sb, cb, s, c;
            ilqr_sincos_fast(a, sa, ca); ilqr_sincos_fast(b, sb, cb);
            ilqr_sincos_sum(a, b, sa, ca, sb, cb, s, c);
            const double t = a + b;
            const double e1 = (double)fabsl((long double)s - sinl((long double)t)), e2 = (double)fabsl((long double)c - cosl((long double)t));
            if (e1 > ma) ma = e1;
            if (e2 > ma) ma = e2;
        }
        printf("sincos_sum       max abs err %.3g\n", ma);
    }
    // reciprocal + shared division
    {
        double mr = 0, md = 0; long nd = 0;
        std::uniform_real_distribution<double> U(0.01, 100.0);
        for (long i = 0; i < n / 4; ++i) {
            const double d = U(g) * (g() % 2 ? 1 : -1), a = U(g);
            const double r = ilqr_rcp(d), qd = ilqr_div_shared(a, d, r);
            const double er = ulp_err(r, 1.0L / d), ed = ulp_err(qd, (long double)a / d);
            if (er > mr) mr = er;
            if (ed > md) md = ed;
            nd += qd != a / d;
        }
        printf("rcp (%d Newton step(s) on a 24-bit estimate) max err %.3f ulp; shared division max err %.3f ulp, differs from IEEE in %.4f %%\n", ILQR_RCP_NEWTON, mr, md, 400.0 * nd / n);
    }
    return 0;
}
