/*
 * oracle/count/flop_count.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Exact count of the floating-point operations of the CPU restatement ("estimate to be replaced by an
 * exact count from the CPU restatement").  The oracle's C sources are compiled a second time as C++ with every
 * `double` replaced by the counting type below (oracle/count/Makefile: -include this header), so that each add,
 * multiply, divide, square root and transcendental call the column solvers execute increments a per-thread counter.
 * Comparisons, min/max/abs, negation, conversions and copies are not counted.  The arithmetic itself is unchanged
 * (the wrapped value is a double and every operation is the plain one), so the counting library also returns the
 * oracle's results.
 */
#pragma once
#define ORACLE_COUNT_FLOPS 1   /* the sources skip their test-aid bookkeeping in this build */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double oracle_real_t;

struct OracleFlopCounters {
    unsigned long long add, mul, div, sqrt_, trans;   /* trans: exp, log, pow, sin, cos, atan, cbrt, erf */
};
extern thread_local OracleFlopCounters oracle_flops;

struct cdbl {
    oracle_real_t v;
    cdbl() = default;
    constexpr cdbl(oracle_real_t x) : v(x) {}
    constexpr cdbl(float x) : v(x) {}
    constexpr cdbl(int x) : v(x) {}
    constexpr cdbl(unsigned x) : v(x) {}
    constexpr cdbl(long x) : v((oracle_real_t)x) {}
    constexpr cdbl(unsigned long x) : v((oracle_real_t)x) {}
    constexpr cdbl(long long x) : v((oracle_real_t)x) {}
    explicit operator oracle_real_t() const { return v; }
    explicit operator float() const { return (float)v; }
    explicit operator int() const { return (int)v; }
    explicit operator unsigned() const { return (unsigned)v; }
    explicit operator long() const { return (long)v; }
    explicit operator unsigned long() const { return (unsigned long)v; }
    explicit operator long long() const { return (long long)v; }
    explicit operator bool() const { return v != 0.0; }
    cdbl &operator+=(cdbl o) { ++oracle_flops.add; v += o.v; return *this; }
    cdbl &operator-=(cdbl o) { ++oracle_flops.add; v -= o.v; return *this; }
    cdbl &operator*=(cdbl o) { ++oracle_flops.mul; v *= o.v; return *this; }
    cdbl &operator/=(cdbl o) { ++oracle_flops.div; v /= o.v; return *this; }
};
static_assert(sizeof(cdbl) == sizeof(oracle_real_t), "layout");
inline cdbl operator+(cdbl a, cdbl b) { ++oracle_flops.add; return cdbl(a.v + b.v); }
inline cdbl operator-(cdbl a, cdbl b) { ++oracle_flops.add; return cdbl(a.v - b.v); }
inline cdbl operator*(cdbl a, cdbl b) { ++oracle_flops.mul; return cdbl(a.v * b.v); }
inline cdbl operator/(cdbl a, cdbl b) { ++oracle_flops.div; return cdbl(a.v / b.v); }
inline cdbl operator-(cdbl a) { return cdbl(-a.v); }
inline cdbl operator+(cdbl a) { return a; }
inline bool operator<(cdbl a, cdbl b) { return a.v < b.v; }
inline bool operator>(cdbl a, cdbl b) { return a.v > b.v; }
inline bool operator<=(cdbl a, cdbl b) { return a.v <= b.v; }
inline bool operator>=(cdbl a, cdbl b) { return a.v >= b.v; }
inline bool operator==(cdbl a, cdbl b) { return a.v == b.v; }
inline bool operator!=(cdbl a, cdbl b) { return a.v != b.v; }
#define ORACLE_MIXED(T)                                                                          \
    inline cdbl operator+(cdbl a, T b) { return a + cdbl(b); }  inline cdbl operator+(T a, cdbl b) { return cdbl(a) + b; } \
    inline cdbl operator-(cdbl a, T b) { return a - cdbl(b); }  inline cdbl operator-(T a, cdbl b) { return cdbl(a) - b; } \
    inline cdbl operator*(cdbl a, T b) { return a * cdbl(b); }  inline cdbl operator*(T a, cdbl b) { return cdbl(a) * b; } \
    inline cdbl operator/(cdbl a, T b) { return a / cdbl(b); }  inline cdbl operator/(T a, cdbl b) { return cdbl(a) / b; } \
    inline bool operator<(cdbl a, T b) { return a.v < b; }   inline bool operator<(T a, cdbl b) { return a < b.v; }   \
    inline bool operator>(cdbl a, T b) { return a.v > b; }   inline bool operator>(T a, cdbl b) { return a > b.v; }   \
    inline bool operator<=(cdbl a, T b) { return a.v <= b; } inline bool operator<=(T a, cdbl b) { return a <= b.v; } \
    inline bool operator>=(cdbl a, T b) { return a.v >= b; } inline bool operator>=(T a, cdbl b) { return a >= b.v; } \
    inline bool operator==(cdbl a, T b) { return a.v == b; } inline bool operator==(T a, cdbl b) { return a == b.v; } \
    inline bool operator!=(cdbl a, T b) { return a.v != b; } inline bool operator!=(T a, cdbl b) { return a != b.v; }
ORACLE_MIXED(oracle_real_t)
ORACLE_MIXED(int)
ORACLE_MIXED(long)
ORACLE_MIXED(unsigned)
ORACLE_MIXED(unsigned long)
ORACLE_MIXED(float)
#undef ORACLE_MIXED

#define ORACLE_TRANS1(f) inline cdbl f(cdbl a) { ++oracle_flops.trans; return cdbl(::f(a.v)); }
ORACLE_TRANS1(exp) ORACLE_TRANS1(log) ORACLE_TRANS1(sin) ORACLE_TRANS1(cos) ORACLE_TRANS1(atan) ORACLE_TRANS1(cbrt)
ORACLE_TRANS1(erf) ORACLE_TRANS1(erfc) ORACLE_TRANS1(log10) ORACLE_TRANS1(tan) ORACLE_TRANS1(expm1) ORACLE_TRANS1(asin) ORACLE_TRANS1(acos)
#undef ORACLE_TRANS1
inline cdbl pow(cdbl a, cdbl b) { ++oracle_flops.trans; return cdbl(::pow(a.v, b.v)); }
inline cdbl atan2(cdbl a, cdbl b) { ++oracle_flops.trans; return cdbl(::atan2(a.v, b.v)); }
inline cdbl sqrt(cdbl a) { ++oracle_flops.sqrt_; return cdbl(::sqrt(a.v)); }
inline cdbl fabs(cdbl a) { return cdbl(::fabs(a.v)); }
inline cdbl floor(cdbl a) { return cdbl(::floor(a.v)); }
inline cdbl ceil(cdbl a) { return cdbl(::ceil(a.v)); }
inline cdbl rint(cdbl a) { return cdbl(::rint(a.v)); }
inline cdbl nearbyint(cdbl a) { return cdbl(::nearbyint(a.v)); }
inline cdbl round(cdbl a) { return cdbl(::round(a.v)); }
inline cdbl trunc(cdbl a) { return cdbl(::trunc(a.v)); }
inline long lround(cdbl a) { return ::lround(a.v); }
inline cdbl ldexp(cdbl a, int e) { return cdbl(::ldexp(a.v, e)); }
inline cdbl fmax(cdbl a, cdbl b) { return cdbl(::fmax(a.v, b.v)); }
inline cdbl fmin(cdbl a, cdbl b) { return cdbl(::fmin(a.v, b.v)); }
inline cdbl fma(cdbl a, cdbl b, cdbl c) { ++oracle_flops.add; ++oracle_flops.mul; return cdbl(::fma(a.v, b.v, c.v)); }
inline cdbl fmod(cdbl a, cdbl b) { ++oracle_flops.div; return cdbl(::fmod(a.v, b.v)); }
inline cdbl copysign(cdbl a, cdbl b) { return cdbl(::copysign(a.v, b.v)); }
inline void sincos(cdbl a, cdbl *s, cdbl *c) { oracle_flops.trans += 2; ::sincos(a.v, &s->v, &c->v); }
#undef isnan
#undef isfinite
#undef isinf
inline bool isnan(cdbl a) { return __builtin_isnan(a.v); }
inline bool isfinite(cdbl a) { return __builtin_isfinite(a.v); }
inline bool isinf(cdbl a) { return __builtin_isinf(a.v); }

/* complex numbers of the optical-averaging oracle (optics_oracle.c uses C99 <complex.h> in the plain build): pairs of
 * the counting type, every operation spelled out in real operations (a complex product = 4 mul + 2 add, a complex
 * quotient = 6 mul + 3 add + 2 div, |z| = 2 mul + 1 add + 1 sqrt), so that the Mie series is counted like the rest */
struct oracle_cplx {
    cdbl re, im;
    oracle_cplx() = default;
    oracle_cplx(cdbl r, cdbl i) : re(r), im(i) {}
    oracle_cplx &operator+=(oracle_cplx o) { re += o.re; im += o.im; return *this; }
};
inline oracle_cplx operator+(oracle_cplx a, oracle_cplx b) { return {a.re + b.re, a.im + b.im}; }
inline oracle_cplx operator-(oracle_cplx a, oracle_cplx b) { return {a.re - b.re, a.im - b.im}; }
inline oracle_cplx operator-(oracle_cplx a) { return {-a.re, -a.im}; }
inline oracle_cplx operator*(oracle_cplx a, oracle_cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline oracle_cplx operator/(oracle_cplx a, oracle_cplx b)
{
    const cdbl d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
inline oracle_cplx operator+(oracle_cplx a, cdbl b) { return {a.re + b, a.im}; }
inline oracle_cplx operator+(cdbl a, oracle_cplx b) { return {a + b.re, b.im}; }
inline oracle_cplx operator-(oracle_cplx a, cdbl b) { return {a.re - b, a.im}; }
inline oracle_cplx operator-(cdbl a, oracle_cplx b) { return {a - b.re, -b.im}; }
inline oracle_cplx operator*(oracle_cplx a, cdbl b) { return {a.re * b, a.im * b}; }
inline oracle_cplx operator*(cdbl a, oracle_cplx b) { return {a * b.re, a * b.im}; }
inline oracle_cplx operator/(oracle_cplx a, cdbl b) { return {a.re / b, a.im / b}; }
inline oracle_cplx operator/(cdbl a, oracle_cplx b)
{
    const cdbl d = b.re * b.re + b.im * b.im;
    return {(a * b.re) / d, -(a * b.im) / d};
}
#define ORACLE_CMIXED(T)                                                                                     \
    inline oracle_cplx operator+(oracle_cplx a, T b) { return a + cdbl(b); } inline oracle_cplx operator+(T a, oracle_cplx b) { return cdbl(a) + b; } \
    inline oracle_cplx operator-(oracle_cplx a, T b) { return a - cdbl(b); } inline oracle_cplx operator-(T a, oracle_cplx b) { return cdbl(a) - b; } \
    inline oracle_cplx operator*(oracle_cplx a, T b) { return a * cdbl(b); } inline oracle_cplx operator*(T a, oracle_cplx b) { return cdbl(a) * b; } \
    inline oracle_cplx operator/(oracle_cplx a, T b) { return a / cdbl(b); } inline oracle_cplx operator/(T a, oracle_cplx b) { return cdbl(a) / b; }
ORACLE_CMIXED(oracle_real_t)
ORACLE_CMIXED(int)
#undef ORACLE_CMIXED
inline cdbl creal(oracle_cplx a) { return a.re; }
inline cdbl cimag(oracle_cplx a) { return a.im; }
inline cdbl cabs(oracle_cplx a) { return sqrt(a.re * a.re + a.im * a.im); }
inline oracle_cplx csqrt(oracle_cplx a)   /* principal root */
{
    const cdbl mod = cabs(a);
    const cdbl re = sqrt(0.5 * (mod + a.re)), im = sqrt(fmax(0.5 * (mod - a.re), cdbl(0.0)));
    return {re, a.im < 0.0 ? -im : im};
}

#define double cdbl
