// polympc_amd — sin / cos / exp restated so that the HOST and the DEVICE produce the same bits.
//
// Why: the reference evaluates the user's dynamics with glibc's sin / cos / exp (through Eigen's AutoDiffScalar). The
// device maths library (ocml) is a different implementation: its results differ from glibc's in the last bit on a few
// per cent of the arguments, and an SQP trajectory amplifies a last-bit difference of one node's dynamics over its
// iterations. With ONE implementation — the classic fdlibm / msun algorithms below: Cody–Waite reduction by pi/2 in three
// steps of 33-bit constants, degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]; exp by reduction with ln 2 and the
// degree-5 rational form — written only with IEEE-754 operations that are correctly rounded on both sides (add, multiply,
// fma, division, integer arithmetic on the bit pattern) the CPU restatement used by the tests and the HIP kernels agree
// bit for bit, and every trajectory test can demand identity instead of a tolerance.
// Accuracy: < 1 ulp (checked against glibc in tests/test_oracle_pins.py) for |x| < 2^19 * pi/2 (sin / cos) and everywhere (exp).
// Beyond that range the same reduction loses accuracy gradually (absolute error ~ 1e-26 |x|^2) and from 2^50 on (sin, cos) is defined as (0, 1) — still
// identical on both sides; no collocation problem has angles of 10^6 rad.
//
// This header has no dependencies and is included by the product (pmpc_ad.hpp) and, as a maths library, by the CPU checker
// (its AD header). Compile with -ffp-contract=off (both build recipes do); the routines use explicit fma where they want one.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PMPC_MATH_HD __host__ __device__
#else
#define PMPC_MATH_HD
#endif

namespace pmpc {
namespace detmath {

#if defined(__clang__)
#define PMPC_MATH_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define PMPC_MATH_NOCONTRACT
#endif

PMPC_MATH_HD inline double from_bits(unsigned long long u) { return __builtin_bit_cast(double, u); }
PMPC_MATH_HD inline unsigned long long to_bits(double x) { return __builtin_bit_cast(unsigned long long, x); }
PMPC_MATH_HD inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

struct SinCos { double s, c; };

// sine kernel on [-pi/4, pi/4], argument x + y (y: tail of the reduced argument)
PMPC_MATH_HD inline double ksin(double x, double y) {
    PMPC_MATH_NOCONTRACT
    const double S1 = from_bits(0xBFC5555555555549ull), S2 = from_bits(0x3F8111111110F8A6ull), S3 = from_bits(0xBF2A01A019C161D5ull),
                 S4 = from_bits(0x3EC71DE357B1FE7Dull), S5 = from_bits(0xBE5AE5E68A2B9CEBull), S6 = from_bits(0x3DE5D93A5ACFD57Cull);
    const double z = x * x;
    const double v = z * x;
    const double r = fma_(z, fma_(z, fma_(z, fma_(z, S6, S5), S4), S3), S2);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
// cosine kernel on [-pi/4, pi/4]
PMPC_MATH_HD inline double kcos(double x, double y) {
    PMPC_MATH_NOCONTRACT
    const double C1 = from_bits(0x3FA555555555554Cull), C2 = from_bits(0xBF56C16C16C15177ull), C3 = from_bits(0x3EFA01A019CB1590ull),
                 C4 = from_bits(0xBE927E4F809C52ADull), C5 = from_bits(0x3E21EE9EBDB4B1C4ull), C6 = from_bits(0xBDA8FAE9BE8838D4ull);
    const double z = x * x;
    const double r = z * fma_(z, fma_(z, fma_(z, fma_(z, fma_(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4 (a little more next to the boundaries); returns n mod 4.
// One straight-line path for every finite x (two rarely taken refinement branches): fn = rint(x * 2/pi) by the magic-number addition, whose
// low mantissa bits ARE the integer (n mod 4 without a conversion); first step with fma (exact below 2^19 pi/2, where fn * PIO2_1 is exact
// anyway, and correctly rounded above).
PMPC_MATH_HD inline int rem_pio2(double x, double& y0, double& y1) {
    PMPC_MATH_NOCONTRACT
    const double INVPIO2 = from_bits(0x3FE45F306DC9C883ull);
    const double PIO2_1 = from_bits(0x3FF921FB54400000ull), PIO2_1T = from_bits(0x3DD0B4611A626331ull);
    const double PIO2_2 = from_bits(0x3DD0B4611A600000ull), PIO2_2T = from_bits(0x3BA3198A2E037073ull);
    const double PIO2_3 = from_bits(0x3BA3198A2E000000ull), PIO2_3T = from_bits(0x397B839A252049C1ull);
    const double MAGIC = 6755399441055744.0;                                    // 1.5 * 2^52
    const double tm = x * INVPIO2 + MAGIC;
    const double fn = tm - MAGIC;
    const int n = (int)(unsigned)to_bits(tm);                                   // low 32 bits of the mantissa = fn mod 2^32 (two's complement)
    double r = fma_(-fn, PIO2_1, x);
    double w = fn * PIO2_1T;
    const int j = (int)((to_bits(x) >> 52) & 0x7ff);
    double y = r - w;
    int i = j - (int)((to_bits(y) >> 52) & 0x7ff);
    if (i > 16) {                                                               // cancellation: second step, good to 118 bits
        double t = r;
        w = fn * PIO2_2; r = t - w; w = fn * PIO2_2T - ((t - r) - w); y = r - w;
        i = j - (int)((to_bits(y) >> 52) & 0x7ff);
        if (i > 49) {                                                           // third step, 151 bits
            t = r;
            w = fn * PIO2_3; r = t - w; w = fn * PIO2_3T - ((t - r) - w); y = r - w;
        }
    }
    // |x| >= 2^50 (the spacing of doubles exceeds 1/8: no phase information left) -> reduced argument 0, i.e. (sin, cos) = (0, 1);
    // inf / NaN -> NaN. Selects, not branches.
    const double ax = x < 0 ? -x : x;
    const bool huge = !(ax < 1125899906842624.0);
    y0 = huge ? (x - x) : y;
    y1 = huge ? 0.0 : ((r - y) - w);
    return huge ? 0 : (n & 3);
}

// both values with one argument reduction
PMPC_MATH_HD inline SinCos sincos(double x) {
    double y0, y1;
    const int n = rem_pio2(x, y0, y1);
    const double ks = ksin(y0, y1), kc = kcos(y0, y1);
    SinCos r;
    r.s = (n & 1) ? kc : ks;
    r.c = (n & 1) ? ks : kc;
    if (n & 2) r.s = -r.s;
    if ((n + 1) & 2) r.c = -r.c;
    return r;
}
PMPC_MATH_HD inline double sin(double x) { return sincos(x).s; }
PMPC_MATH_HD inline double cos(double x) { return sincos(x).c; }

// exp: k = rint(x / ln 2), r = x - k ln 2 in two pieces, the degree-5 rational form on r, scaling by 2^k in two factors (one rounding even
// when the result is subnormal). Straight-line: the special cases are selects on the argument at the end.
PMPC_MATH_HD inline double exp(double x) {
    PMPC_MATH_NOCONTRACT
    const double LN2HI = from_bits(0x3FE62E42FEE00000ull), LN2LO = from_bits(0x3DEA39EF35793C76ull), INVLN2 = from_bits(0x3FF71547652B82FEull);
    const double P1 = from_bits(0x3FC555555555553Eull), P2 = from_bits(0xBF66C16C16BEBD93ull), P3 = from_bits(0x3F11566AAF25DE2Cull),
                 P4 = from_bits(0xBEBBBD41C5D26BF1ull), P5 = from_bits(0x3E66376972BEA4D0ull);
    const double MAGIC = 6755399441055744.0;
    // the argument is clamped for the arithmetic (results outside the clamp are replaced below): k stays in [-1080, 1030]
    const double xc = x > 712.0 ? 712.0 : (x < -748.0 ? -748.0 : x);
    const double tm = xc * INVLN2 + MAGIC;
    const double t = tm - MAGIC;
    const int k = (int)(unsigned)to_bits(tm);
    const double hi = xc - t * LN2HI;                                           // exact product (32-bit constant)
    const double lo = t * LN2LO;
    const double r = hi - lo;
    const double z = r * r;
    const double c = r - z * fma_(z, fma_(z, fma_(z, fma_(z, P5, P4), P3), P2), P1);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k1 = k >> 1, k2 = k - k1;
    const double s1 = from_bits((unsigned long long)(unsigned)(k1 + 1023) << 52), s2 = from_bits((unsigned long long)(unsigned)(k2 + 1023) << 52);
    double e = (y * s1) * s2;
    e = x > 709.782712893384 ? from_bits(0x7FF0000000000000ull) : e;
    e = x < -745.1332191019412 ? 0.0 : e;
    return x != x ? x + x : e;
}

}  // namespace detmath
}  // namespace pmpc
