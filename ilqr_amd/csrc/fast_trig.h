// fast_trig.h -- fp64 sin+cos with ONE shared argument reduction, for the generated dynamics on gfx950.
//
// The reference evaluates numpy sin/cos (glibc, < 0.55 ulp).  ocml's sin()/cos() are accurate but each runs its own
// general range reduction (Payne-Hanek capable), ~80-100 VALU instructions per call, and the dynamics need both sin
// and cos of the same few angles every time step: trig was ~70 % of the line-search kernel.  Here:
//   * |x| < 1e6 (every angle a sane trajectory reaches): n = rint(x * 2/pi), two Cody-Waite rounds with 33-bit pieces of
//     pi/2 (n * piece is exact for |n| < 2^20, so the reduced argument y0 + y1 carries ~118 bits), then the classic
//     minimax kernels for sin and cos on [-pi/4, pi/4] with the tail y1, and a quadrant select.  ~45 instructions for BOTH
//     values.  Measured on the host against 80-bit sinl/cosl: max error 0.78 ulp over |x| < 5e5 (glibc: 0.52), 0.50 ulp near
//     multiples of pi/2; it differs from glibc in 3 % of arguments, always by one ulp.
//   * otherwise (huge, inf, nan): ocml sin()/cos(), out of line, reached through ONE branch per generated function.
// This file is compiled with -ffp-contract=off; every fused operation is an explicit __builtin_fma (the exact
// products n * piece, the polynomials and the combination steps of the two kernels).  tools/trig_accuracy.cpp measures it.
#ifndef ILQR_FAST_TRIG_H
#define ILQR_FAST_TRIG_H

// out of line on purpose: keeps ocml's general reduction (and its registers) out of the time-step loops.  The pair
// comes back BY VALUE (registers): out-pointers would pin sn/cs of every caller to scratch memory, and the scratch
// round trip's s_waitcnt vmcnt(0) also drains the trajectory prefetch of the time-step loop.
struct ilqr_sc { double sinv, cosv; };
__device__ __attribute__((noinline)) ilqr_sc ilqr_sincos_slow(const double x)
{
    ilqr_sc r;
    r.sinv = sin(x);
    r.cosv = cos(x);
    return r;
}

// x with its sign bit xor-ed with `sign` (0 or 0x80000000)
__device__ __forceinline__ double ilqr_flip_sign(const double x, const unsigned sign)
{
    unsigned long long b;
    __builtin_memcpy(&b, &x, 8);
    b ^= (unsigned long long)sign << 32;
    double y;
    __builtin_memcpy(&y, &b, 8);
    return y;
}

// fast path only; the caller patches arguments with !(|x| < 1e6) through ilqr_sincos_slow (one branch per function)
__device__ __forceinline__ void ilqr_sincos_fast(const double x, double& s_out, double& c_out)
{
    constexpr double invpio2 = 6.36619772367581382433e-01;
    constexpr double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    constexpr double pio2_2 = 6.07710050630396597660e-11;   // next 33 bits
    constexpr double pio2_2t = 2.02226624879595063154e-21;  // pi/2 - (pio2_1 + pio2_2)
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = __builtin_rint(x * invpio2);
    const int q = (int)fn;
    const double t = __builtin_fma(-fn, pio2_1, x);          // fn * pio2_1 is exact, and so is this difference
    double w = fn * pio2_2;                                   // exact product
    const double r = t - w;
    w = __builtin_fma(fn, pio2_2t, -((t - r) - w));
    const double y0 = r - w;
    const double y1 = (r - y0) - w;                           // tail of the reduced argument
    const double z = y0 * y0;
    const double v = z * y0;
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
    // y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1), the inner steps fused
    const double ks = y0 - __builtin_fma(-v, S1, __builtin_fma(z, __builtin_fma(-v, rs, 0.5 * y1), -y1));
    const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    // wc + (((1 - wc) - hz) + (z * z * pc - y0 * y1))
    const double kc = wc + (((1.0 - wc) - hz) + __builtin_fma(z * z, pc, -(y0 * y1)));
    // quadrant: swap on bit 0 of q; sin changes sign on bit 1 of q, cos on bit 1 of q + 1 (sign bits set by shifts,
    // no compares)
    const double s = (q & 1) ? kc : ks;
    const double c = (q & 1) ? ks : kc;
    const unsigned ss = ((unsigned)q << 30) & 0x80000000u;
    const unsigned sc = ((unsigned)(q + 1) << 30) & 0x80000000u;
    s_out = ilqr_flip_sign(s, ss);
    c_out = ilqr_flip_sign(c, sc);
}

// sin and cos of t = fl(a + b) -- the argument the reference hands to numpy -- from the pairs of a and b: angle addition
// on the exact sum a + b = t + e (two-sum), then first order in the rounding error e (|e| < 2^-33 for |a|,|b| < 1e6, so
// the neglected e^2/2 is < 1e-20).  12 instructions instead of a third range reduction + kernels (~50).  Absolute error
// <= 3e-16 (the inputs' 0.78 ulp carried through), i.e. NOT relative near zeros of the result: the generator uses it only
// in f, where every trig value is scaled by dt before it is added to the state.
__device__ __forceinline__ void ilqr_sincos_sum(const double a, const double b, const double sa, const double ca,
                                                const double sb, const double cb, double& s_out, double& c_out)
{
    const double t = a + b;
    const double bb = t - a;
    const double e = (a - (t - bb)) + (b - bb);
    const double S = __builtin_fma(sa, cb, ca * sb);
    const double C = __builtin_fma(ca, cb, -(sa * sb));
    s_out = __builtin_fma(-C, e, S);
    c_out = __builtin_fma(S, e, C);
}

// the same from pairs that are themselves carried along a rollout (TrigCarry): plain angle addition, i.e. sin and cos of
// the exact sum a + b rather than of its rounding -- the difference, |e| <= ulp(a + b) / 2, is below those pairs' own error
__device__ __forceinline__ void ilqr_sincos_sum_carried(const double sa, const double ca, const double sb, const double cb,
                                                        double& s_out, double& c_out)
{
    s_out = __builtin_fma(sa, cb, ca * sb);
    c_out = __builtin_fma(ca, cb, -(sa * sb));
}

// 1/d for the shared-reciprocal division below: hardware estimate + ILQR_RCP_NEWTON Newton steps.  The reciprocal is
// only ever used by ilqr_div_shared, whose residual correction is itself a Newton step on the quotient (relative error
// of the result = the reciprocal's, squared): ONE step on the estimate is enough for a quotient that equals the IEEE
// one in all but rare half-ulp cases (tools/trig_accuracy.cpp with a 24-bit stand-in for the estimate: reciprocal up
// to 16 ulp off, no quotient off in 10^6; the example solves' parity margins are digit for digit those of two steps).
// 3 instructions per divisor instead of the 11 of the compiler's own division.  d = 0 gives NaN (the IEEE quotient
// a/0 is inf or NaN: the trajectory is lost either way).
#ifndef ILQR_RCP_NEWTON
#define ILQR_RCP_NEWTON 1
#endif
__device__ __forceinline__ double ilqr_rcp(const double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < ILQR_RCP_NEWTON; ++i) r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    return r;
}

// a / d with a shared, (nearly always) correctly rounded reciprocal rd = 1/d: one multiply and one fused residual correction
// (Markstein): equals the IEEE quotient except in rare half-ulp cases; exponent extremes are not rescaled (the
// divisors of the generated dynamics are O(1) physical quantities).
__device__ __forceinline__ double ilqr_div_shared(const double a, const double d, const double rd)
{
    const double q = a * rd;
    const double e = __builtin_fma(-d, q, a);
    return __builtin_fma(e, rd, q);
}

#endif
