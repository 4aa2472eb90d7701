// pow(x, y) with the bits of glibc's pow on an x86-64 host with FMA - the function the reference reaches through
// f64::powf (render.rs:47 gamma, material.rs:200 specular) and the oracle through libm.
//
// glibc >= 2.28 computes pow as exp(y * log(x)) in double-double pieces (sysdeps/ieee754/dbl-64/e_pow.c: a 128-entry
// table for log, a 128-entry table for 2^(k/128), two short polynomials; algorithm and tables from ARM's
// optimized-routines). glibc is not under /root/reference: it is the third-party dependency behind `powf`, pinned here as
// glibc 2.35 (Ubuntu 2.35-0ubuntu3.x, this image, also on the GPU box). Its x86-64 build selects `__pow_fma` on CPUs
// with FMA + AVX2 - a translation unit compiled with -mfma, in which the compiler ALSO contracted most a * b + c of the
// source into fused operations. Which ones is a property of that binary, so the sequence below is the one read off the
// disassembly of __pow_fma in this image's libm.so.6 (every fma() here is one vfmadd there; every separate * and + is
// separate there), and it is pinned by tests: tests/test_pow_exact.py compares this function, built for the host, with
// libm's pow bit for bit on millions of arguments (0 differences), and on the GPU `pt_test_math` op 6 does the same for
// the device build. The constants come out of the same libm.so.6 (tools/gen_pow_tables.py -> pt_pow_tables.h).
//
// On the device this replaces ocml's pow (within 1 ulp of glibc's, 83.7 % bit-equal: the one place where linear f64
// results differed from the oracle's) and is a good deal shorter.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "pt_pow_tables.h"

#if defined(__HIPCC__)
#define PT_POW_FN static __host__ __device__ __forceinline__
#else
#define PT_POW_FN static inline
#endif

PT_POW_FN uint64_t pt_pow_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
PT_POW_FN double pt_pow_f64(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

// 0: y is not an integer, 1: odd integer, 2: even integer (iy = bits of a non-zero finite y)
PT_POW_FN int pt_pow_checkint(uint64_t iy) {
    const int e = (int)(iy >> 52) & 0x7ff;
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ULL << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ULL << (0x3ff + 52 - e))) return 1;
    return 2;
}
PT_POW_FN bool pt_pow_zeroinfnan(uint64_t i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ULL - 1; }

// sign * exp(x + xtail), |xtail| small against x; sign_bias = 0 or 0x800 << 7 (negative result)
PT_POW_FN double pt_pow_exp(double x, double xtail, uint32_t sign_bias) {
    uint32_t abstop = (uint32_t)(pt_pow_bits(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {  // |x| < 2^-54 or >= 512
        if (abstop - 0x3c9u >= 0x80000000u) {  // tiny: the result rounds like 1 + x
            const double one = 1.0 + x;
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409u) {  // |x| >= 1024: underflow / overflow (inf and nan were handled by the caller)
            if (pt_pow_bits(x) >> 63) return sign_bias ? -0.0 : 0.0;                    // __math_uflow
            return sign_bias ? -INFINITY : INFINITY;                                   // __math_oflow
        }
        abstop = 0;  // 512 <= |x| < 1024: the scale may leave the normal range, handled at the end
    }
    // x = k ln2 / 128 + r, |r| <= ln2 / 256
    double kd = fma(x, PT_EXP_INVLN2N, PT_EXP_SHIFT);
    const uint64_t ki = pt_pow_bits(kd);
    kd -= PT_EXP_SHIFT;
    double r = fma(kd, PT_EXP_NEGLN2LON, fma(kd, PT_EXP_NEGLN2HIN, x));
    r = xtail + r;
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = (ki + sign_bias) << 45;
    const double tail = pt_pow_f64(pt_exp_tab[idx]);
    uint64_t sbits = pt_exp_tab[idx + 1] + top;
    const double r2 = r * r;
    double tmp = fma(r2, fma(r, PT_EXP_C3, PT_EXP_C2), r + tail);
    tmp = fma(r2 * r2, fma(r, PT_EXP_C5, PT_EXP_C4), tmp);
    if (abstop == 0) {  // scale * (1 + tmp) where the scale's exponent may be out of range
        if ((ki & 0x80000000u) == 0) {  // k > 0: the exponent may have overflowed by up to 460
            sbits -= 1009ULL << 52;
            const double scale = pt_pow_f64(sbits);
            return 0x1p1009 * fma(scale, tmp, scale);
        }
        sbits += 1022ULL << 52;  // k < 0: care in the subnormal range
        const double scale = pt_pow_f64(sbits);
        const double st = scale * tmp;
        double y = scale + st;
        if (fabs(y) < 1.0) {  // round to the right precision before scaling into the subnormal range
            const double one = y < 0.0 ? -1.0 : 1.0;
            double lo = (scale - y) + st;
            const double hi = one + y;
            lo = ((one - hi) + y) + lo;
            y = (hi + lo) - one;
            if (y == 0.0) y = pt_pow_f64(sbits & 0x8000000000000000ULL);
        }
        return 0x1p-1022 * y;
    }
    const double scale = pt_pow_f64(sbits);
    return fma(scale, tmp, scale);
}

PT_POW_FN double pt_pow_glibc(double x, double y) {
    uint32_t sign_bias = 0;
    uint64_t ix = pt_pow_bits(x);
    const uint64_t iy = pt_pow_bits(y);
    uint32_t topx = (uint32_t)(ix >> 52);
    const uint32_t topy = (uint32_t)(iy >> 52);
    if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
        // x < 2^-1022 (zero, subnormal, negative), inf or nan; or |y| < 2^-65, |y| >= 2^63 or nan
        if (pt_pow_zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0;                                     // x^0 (signalling nans aside)
            if (ix == 0x3ff0000000000000ULL) return 1.0;                     // 1^y
            if (2 * ix > 2 * 0x7ff0000000000000ULL || 2 * iy > 2 * 0x7ff0000000000000ULL) return x + y;  // nan
            if (2 * ix == 2 * 0x3ff0000000000000ULL) return 1.0;             // (-1)^(+-inf)
            if ((2 * ix < 2 * 0x3ff0000000000000ULL) == !(iy >> 63)) return 0.0;  // |x| < 1, y = +inf; |x| > 1, y = -inf
            return y * y;
        }
        if (pt_pow_zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && pt_pow_checkint(iy) == 1) x2 = -x2;
            return (iy >> 63) ? 1.0 / x2 : x2;  // 0^negative = inf (__math_divzero gives the same value)
        }
        // x and y are non-zero finite here
        if (ix >> 63) {  // finite x < 0
            const int yint = pt_pow_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);  // __math_invalid: nan
            if (yint == 1) sign_bias = 0x800u << 7;
            ix &= 0x7fffffffffffffffULL;
            topx &= 0x7ff;
        }
        if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
            if (ix == 0x3ff0000000000000ULL) return 1.0;
            if ((topy & 0x7ff) < 0x3be) return ix > 0x3ff0000000000000ULL ? 1.0 + y : 1.0 - y;  // |y| < 2^-65
            return (ix > 0x3ff0000000000000ULL) == (topy < 0x800) ? INFINITY : 0.0;             // huge |y|: overflow / underflow
        }
        if (topx == 0) {  // subnormal x: normalise so that the exponent goes negative
            ix = pt_pow_bits(x * 0x1p52);
            ix &= 0x7fffffffffffffffULL;
            ix -= 52ULL << 52;
        }
    }
    // log(x) = k ln2 + log(c) + log1p(z / c - 1) as hi + lo, x = 2^k z, z in [0x1.69555p-1, 0x1.69555p0), c = the centre of z's subinterval
    const uint64_t tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) % 128);
    const int64_t k = (int64_t)tmp >> 52;
    const double z = pt_pow_f64(ix - (tmp & (0xfffULL << 52)));
    const double kd = (double)k;
    const double invc = pt_pow_log_tab[3 * i], logc = pt_pow_log_tab[3 * i + 1], logctail = pt_pow_log_tab[3 * i + 2];
    const double r = fma(z, invc, -1.0);  // exact
    const double t1 = fma(kd, PT_POW_LN2HI, logc);
    const double t2 = t1 + r;
    const double lo1 = fma(kd, PT_POW_LN2LO, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = PT_POW_A0 * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double q = fma(ar2, fma(ar2, fma(r, PT_POW_A6, PT_POW_A5), fma(r, PT_POW_A4, PT_POW_A3)), fma(r, PT_POW_A2, PT_POW_A1));
    const double lo = fma(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double lhi = hi + lo;
    const double llo = (hi - lhi) + lo;
    const double ehi = y * lhi;
    const double elo = fma(y, llo, fma(y, lhi, -ehi));
    return pt_pow_exp(ehi, elo, sign_bias);
}
