// lean_math.hip.h -- short float64 exp / log / log1p / 1/x for the latency-bound serial chains of the
// kernels (prior part, neutral and dynamic kernels, the NUTS leaf's weights).
#pragma once
#include <hip/hip_runtime.h>
namespace dc {
// ------------------------------------------------------------------ lean float64 math
// The z side of the float64 kernels is a handful of exp / log / log1p / 1/x per lane on the
// critical path of a latency-bound launch, and the device library's versions are 100-250
// dependent instructions each (~0.5-0.7 us measured: three of them were 2.9 us of one phase).
// These are the textbook reductions with no extended-precision tail: ~1-2 ulp, 20-35
// instructions.  (Checked against libm by bplhip_selftest_math, tests/test_gpu_lean_math.py.)
namespace lean {
// exp: x = k ln2 + r (Cody-Waite, ln2 split so that k*ln2_hi is exact), Taylor to r^13 on
// |r| <= ln2/2 (truncation 4e-18), scaled by ldexp.  Clamped to +-750: exp -> inf / 0 there.
__device__ __forceinline__ double exp(double x) {
    x = x > 750.0 ? 750.0 : (x < -750.0 ? -750.0 : x);  // (a NaN passes both comparisons)
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                   // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);       // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);      // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);      // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);     // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);       // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);      // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);      // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);      // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);     // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);     // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
// 1/x for a normal x that is neither 0 nor inf: hardware estimate + two Newton steps
__device__ __forceinline__ double rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
    return y;
}
// log of m * 2^e with m in [sqrt(1/2), sqrt(2)): f = m - 1, s = f / (2 + f),
// log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the classic degree-7 minimax R (fdlibm's
// published coefficients, |error| < 2^-58).  0 -> -inf, negative or NaN -> NaN, inf -> inf.
__device__ __forceinline__ double log(double a) {
    double m = __builtin_amdgcn_frexp_mant(a);           // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(a);
    const bool low = m < 0.7071067811865476;
    m = low ? 2.0 * m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * rcp(2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = __builtin_fma(R, z, 1.531383769920937332e-01);
    R = __builtin_fma(R, z, 1.818357216161805012e-01);
    R = __builtin_fma(R, z, 2.222219843214978396e-01);
    R = __builtin_fma(R, z, 2.857142874366239149e-01);
    R = __builtin_fma(R, z, 3.999999999940941908e-01);
    R = __builtin_fma(R, z, 6.666666666666735130e-01);
    R *= z;
    const double hfsq = 0.5 * f * f;
    const double de = (double)e;
    const double lo = __builtin_fma(s, hfsq + R, de * 1.90821492927058770002e-10);
    double r = __builtin_fma(de, 6.93147180369123816490e-01, f - (hfsq - lo));
    r = a == 0.0 ? -__builtin_inf() : r;
    r = a == __builtin_inf() ? a : r;
    r = a >= 0.0 ? r : __builtin_nan("");
    return r;
}
// log(1 + x) for x >= 0 (here x = exp(-|z|) <= 1): the rounding error of 1 + x is put back
__device__ __forceinline__ double log1p_pos(double x) {
    const double u = 1.0 + x;
    const double c = x - (u - 1.0);
    return log(u) + c * rcp(u);
}
}  // namespace lean
}  // namespace dc
