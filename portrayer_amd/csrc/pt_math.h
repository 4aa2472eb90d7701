// f64 vector arithmetic of the hot path, shared by the gfx950 kernels and the host-side packing code.
//
// Replaces the arithmetic portrayer gets from vek / roots (reference: src/math.rs:22-114). The
// ORDER of operations inside every expression is part of the contract: it is what makes pixels
// identical to the CPU renderer, so nothing here may be re-associated or fused. The build passes
// -ffp-contract=off (Rust/LLVM never contracts a*b+c) and no fast-math flag.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#define PT_NOINLINE static __host__ __device__ __noinline__
#else
#define PT_HD inline
#define PT_NOINLINE inline
#endif

#define PT_EPSILON 0.00001  // math.rs:15
#define PT_GAMMA 2.2        // math.rs:20

struct PtVec3 {
    double x, y, z;
};

PT_HD PtVec3 pt_v3(double x, double y, double z) { PtVec3 v; v.x = x; v.y = y; v.z = z; return v; }
PT_HD PtVec3 operator+(PtVec3 a, PtVec3 b) { return pt_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD PtVec3 operator-(PtVec3 a, PtVec3 b) { return pt_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD PtVec3 operator-(PtVec3 a) { return pt_v3(-a.x, -a.y, -a.z); }
PT_HD PtVec3 operator*(PtVec3 a, PtVec3 b) { return pt_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD PtVec3 operator*(PtVec3 a, double s) { return pt_v3(a.x * s, a.y * s, a.z * s); }
PT_HD PtVec3 operator/(PtVec3 a, double s) { return pt_v3(a.x / s, a.y / s, a.z / s); }
// vek dot: products summed left to right
PT_HD double pt_dot(PtVec3 a, PtVec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_HD PtVec3 pt_cross(PtVec3 a, PtVec3 b) {
    return pt_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD double pt_length(PtVec3 a) { return sqrt(pt_dot(a, a)); }
PT_HD PtVec3 pt_normalized(PtVec3 a) { return a / pt_length(a); }  // vek: self / self.magnitude()

// Rows 0..2 of an affine 4x4 stored row-major as 12 doubles (m[4*r + c]).
// math.rs:45-47 transformed_point (w = 1)
PT_HD PtVec3 pt_xform_point(const double* m, PtVec3 v) {
    return pt_v3(((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3],
                 ((m[4] * v.x + m[5] * v.y) + m[6] * v.z) + m[7],
                 ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11]);
}
// math.rs:49-51 transformed_direction (w = 0); stride = 4 for a 3x4 block, 3 for a packed 3x3
PT_HD PtVec3 pt_xform_dir(const double* m, int stride, PtVec3 v) {
    return pt_v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z,
                 (m[stride] * v.x + m[stride + 1] * v.y) + m[stride + 2] * v.z,
                 (m[2 * stride] * v.x + m[2 * stride + 1] * v.y) + m[2 * stride + 2] * v.z);
}

// std::ops::Range<f64>::contains: start <= t < end (false for NaN)
PT_HD bool pt_in_range(double start, double end, double t) { return start <= t && t < end; }

// roots::find_roots_quadratic + Solutions::find_in_range (math.rs:94-96, :109-113): the first root
// in ascending order that lies in [start, end).
PT_HD bool pt_first_root(double a2, double a1, double a0, double start, double end, double* t) {
    if (a2 == 0.0) {
        if (a1 == 0.0) {
            if (a0 == 0.0 && pt_in_range(start, end, 0.0)) { *t = 0.0; return true; }
            return false;
        }
        double r = -a0 / a1;
        if (pt_in_range(start, end, r)) { *t = r; return true; }
        return false;
    }
    double disc = a1 * a1 - 4.0 * a2 * a0;
    if (disc < 0.0) return false;
    double a2x2 = 2.0 * a2;
    if (disc == 0.0) {
        double r = -a1 / a2x2;
        if (pt_in_range(start, end, r)) { *t = r; return true; }
        return false;
    }
    double sq = sqrt(disc);
    double x1 = (-a1 - sq) / a2x2;
    double x2 = (-a1 + sq) / a2x2;
    double lo = x1 < x2 ? x1 : x2;
    double hi = x1 < x2 ? x2 : x1;
    if (pt_in_range(start, end, lo)) { *t = lo; return true; }
    if (pt_in_range(start, end, hi)) { *t = hi; return true; }
    return false;
}

// Sampling contract (the reference uses thread_rng and cannot be seeded, render.rs:36-43,
// material.rs:106): a counter-based generator keyed by (seed, pixel, sample, draw);
// f64 = (u64 >> 11) * 2^-53 like rand 0.7's Standard distribution.
PT_HD uint64_t pt_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
// The generator in two steps: the key of a pixel (two of the three mixing rounds) and one draw from it. A lane that needs
// several draws of the same pixel computes the key once.
PT_HD uint64_t pt_rng_key(uint64_t seed, uint64_t pixel) {
    uint64_t k = pt_mix64(seed + 0x9e3779b97f4a7c15ULL);
    return pt_mix64(k ^ (pixel * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL));
}
PT_HD double pt_rng_draw(uint64_t key, uint32_t sample, uint32_t draw) {
    uint64_t k = pt_mix64(key ^ (((uint64_t)sample << 32) | (uint64_t)draw));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}
PT_HD double pt_rng_f64(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t draw) {
    return pt_rng_draw(pt_rng_key(seed, pixel), sample, draw);
}

// Smallest double above a positive finite t (used to admit t == t_best for the index tie-break).
PT_HD double pt_next_up(double t) {
    if (!(t < INFINITY) || !(t > 0.0)) return t;
    union { double d; uint64_t u; } c;
    c.d = t; c.u += 1;
    return c.d;
}

// ---- f64 division by a denominator that is divided by again and again (a ray's direction component: every straddled split of a
// k-d walk computes (plane - o) / d, node.rs:90-109).
// The compiler expands an IEEE f64 division into: v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 fma), q0 = n y,
// r = fma(-d, q0, n), q1 = v_div_fmas(r, y, q0), v_div_fixup. For operands in the normal range v_div_scale returns its operand
// unchanged (and clears VCC, so v_div_fmas is a plain fma) and v_div_fixup returns q1 unchanged: the quotient IS
// fma(fma(-d, n y, n), y, n y) with y = the twice-refined reciprocal of d - a function of d alone. pt_rcp_refined() computes that y
// with the very instructions of the expansion, once per ray and axis; pt_div_fast() finishes a division in three instructions, bit
// for bit the hardware sequence's result (not by an error analysis: by being the same operations on the same operands).
// "Normal range" (V_DIV_SCALE_F64 / V_DIV_FIXUP_F64 in the CDNA ISA guide): both operands finite and non-zero, d and 1 / d normal,
// exponent(n) - exponent(d) in (-1022, 768), exponent(n) > 53. pt_div_exp_ok() is a sufficient test: the biased exponent in
// [640, 1407], i.e. 2^-383 <= |x| < 2^385; anything else - zeros, denormals, infinities, NaNs included - takes the real division.
// Checked against `/` on the device over the whole window and its edges (tests/test_gpu_device_parity.py).
PT_HD double pt_rcp_refined(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, y0, 1.0);
    const double y1 = __builtin_fma(y0, e0, y0);
    const double e1 = __builtin_fma(-d, y1, 1.0);
    return __builtin_fma(y1, e1, y1);
#else
    return 1.0 / d;
#endif
}
PT_HD bool pt_div_exp_ok(double x) {
    union { double d; uint32_t u[2]; } c; c.d = x;
    return (((c.u[1] >> 20) & 0x7FFu) - 640u) <= 767u;
}
PT_HD double pt_div_fast(double n, double d, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double q0 = n * y;
    const double r = __builtin_fma(-d, q0, n);
    return __builtin_fma(r, y, q0);
#else
    (void)y;
    return n / d;
#endif
}

