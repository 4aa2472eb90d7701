/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the arithmetic portrayer gets from `vek 0.9.8`, `roots 0.0.5` and libm
 * (reference: src/math.rs:15-114; third-party crates are NOT under /root/reference, pinned in
 * Cargo.lock:669-671 (roots 0.0.5) and Cargo.lock:798-800 (vek 0.9.8)).
 *
 * Nothing in the product (portrayer_amd/, include/) may include this file. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use the oracle, as the checker.
 *
 * Parity status: the operation ORDER inside vek's dot / Mat*Vec / Mat*Mat / inverse and inside
 * roots' quadratic solver is "parity unpinned" at the ULP level (the crates are absent and the
 * Rust binary cannot be built here). What is pinned: math.rs:159-179 (quadratic
 * known answers, 1e-6), bounding_box.rs:171-195 (rotation / scale composition, 1e-3) and the
 * committed renders (image level). The order chosen here is written next to every function and
 * is the order the HIP path follows too.
 *
 * Build with -ffp-contract=off: Rust/LLVM never fuses a*b+c (SURVEY App.B.5).
 */
#ifndef PO_MATH_H
#define PO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define PO_EPSILON 0.00001 /* math.rs:15 */
#define PO_GAMMA 2.2       /* math.rs:20 */

typedef struct { double x, y, z; } po_vec3;
/* Row-major storage m[r][c]; vek stores columns but every use here is through M*v / M*M /
 * inverse / transpose so only the mathematical matrix matters. */
typedef struct { double m[4][4]; } po_mat4;
typedef struct { double start, end; } po_range; /* std::ops::Range<f64>: start <= t < end */

static inline int po_contains(const po_range *r, double t) { return r->start <= t && t < r->end; }

static inline po_vec3 po_v3(double x, double y, double z) { po_vec3 v = {x, y, z}; return v; }
static inline po_vec3 po_add(po_vec3 a, po_vec3 b) { return po_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline po_vec3 po_sub(po_vec3 a, po_vec3 b) { return po_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline po_vec3 po_mul(po_vec3 a, po_vec3 b) { return po_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline po_vec3 po_scale(po_vec3 a, double s) { return po_v3(a.x * s, a.y * s, a.z * s); }
static inline po_vec3 po_divs(po_vec3 a, double s) { return po_v3(a.x / s, a.y / s, a.z / s); }
static inline po_vec3 po_neg(po_vec3 a) { return po_v3(-a.x, -a.y, -a.z); }
/* vek dot = (a*b).sum(), sum folds left to right: ((x + y) + z) */
static inline double po_dot(po_vec3 a, po_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline po_vec3 po_cross(po_vec3 a, po_vec3 b) {
    return po_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double po_magnitude(po_vec3 a) { return sqrt(po_dot(a, a)); }
/* vek normalized() = self / self.magnitude() (component-wise division) */
static inline po_vec3 po_normalized(po_vec3 a) { return po_divs(a, po_magnitude(a)); }
/* vek partial_min / partial_max: `if a <= b {a} else {b}` / `if a >= b {a} else {b}` */
static inline double po_pmin(double a, double b) { return a <= b ? a : b; }
static inline double po_pmax(double a, double b) { return a >= b ? a : b; }
static inline po_vec3 po_vmin(po_vec3 a, po_vec3 b) { return po_v3(po_pmin(a.x, b.x), po_pmin(a.y, b.y), po_pmin(a.z, b.z)); }
static inline po_vec3 po_vmax(po_vec3 a, po_vec3 b) { return po_v3(po_pmax(a.x, b.x), po_pmax(a.y, b.y), po_pmax(a.z, b.z)); }

static inline po_mat4 po_identity(void) {
    po_mat4 r; memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
    return r;
}

/* Mat4 * Mat4: each entry is the row.col dot product summed left to right. */
static inline po_mat4 po_matmul(const po_mat4 *a, const po_mat4 *b) {
    po_mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((a->m[i][0] * b->m[0][j] + a->m[i][1] * b->m[1][j]) + a->m[i][2] * b->m[2][j]) + a->m[i][3] * b->m[3][j];
    return r;
}

/* math.rs:45-47 transformed_point: Vec3::from(M * Vec4::from_point(v)); w = 1 so the last
 * product is exact and is written as a plain add; the w row is dropped (Vec3::from(Vec4)). */
static inline po_vec3 po_xform_point(const po_mat4 *m, po_vec3 v) {
    return po_v3(((m->m[0][0] * v.x + m->m[0][1] * v.y) + m->m[0][2] * v.z) + m->m[0][3],
                 ((m->m[1][0] * v.x + m->m[1][1] * v.y) + m->m[1][2] * v.z) + m->m[1][3],
                 ((m->m[2][0] * v.x + m->m[2][1] * v.y) + m->m[2][2] * v.z) + m->m[2][3]);
}
/* math.rs:49-51 transformed_direction: w = 0, the translation column does not take part. */
static inline po_vec3 po_xform_dir(const po_mat4 *m, po_vec3 v) {
    return po_v3((m->m[0][0] * v.x + m->m[0][1] * v.y) + m->m[0][2] * v.z,
                 (m->m[1][0] * v.x + m->m[1][1] * v.y) + m->m[1][2] * v.z,
                 (m->m[2][0] * v.x + m->m[2][1] * v.y) + m->m[2][2] * v.z);
}

static inline po_mat4 po_transposed(const po_mat4 *a) {
    po_mat4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a->m[j][i];
    return r;
}

/* General 4x4 inverse (vek Mat4::inverted; used by scene.rs:201-205, flat_scene.rs:103-108,
 * bounding_box.rs:68-69, camera.rs:38). Adjugate over determinant via the 2x2 sub-determinants
 * of the top two rows (s0..s5) and bottom two rows (c0..c5). */
static inline po_mat4 po_inverted(const po_mat4 *a) {
    const double (*m)[4] = a->m;
    double s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1];
    double s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    double s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3];
    double s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    double s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3];
    double s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    double c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    double c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    double c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    double c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    double c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    double c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    double id = 1.0 / det;
    po_mat4 r;
    r.m[0][0] = ((m[1][1] * c5 - m[1][2] * c4) + m[1][3] * c3) * id;
    r.m[0][1] = ((-m[0][1] * c5 + m[0][2] * c4) - m[0][3] * c3) * id;
    r.m[0][2] = ((m[3][1] * s5 - m[3][2] * s4) + m[3][3] * s3) * id;
    r.m[0][3] = ((-m[2][1] * s5 + m[2][2] * s4) - m[2][3] * s3) * id;
    r.m[1][0] = ((-m[1][0] * c5 + m[1][2] * c2) - m[1][3] * c1) * id;
    r.m[1][1] = ((m[0][0] * c5 - m[0][2] * c2) + m[0][3] * c1) * id;
    r.m[1][2] = ((-m[3][0] * s5 + m[3][2] * s2) - m[3][3] * s1) * id;
    r.m[1][3] = ((m[2][0] * s5 - m[2][2] * s2) + m[2][3] * s1) * id;
    r.m[2][0] = ((m[1][0] * c4 - m[1][1] * c2) + m[1][3] * c0) * id;
    r.m[2][1] = ((-m[0][0] * c4 + m[0][1] * c2) - m[0][3] * c0) * id;
    r.m[2][2] = ((m[3][0] * s4 - m[3][1] * s2) + m[3][3] * s0) * id;
    r.m[2][3] = ((-m[2][0] * s4 + m[2][1] * s2) - m[2][3] * s0) * id;
    r.m[3][0] = ((-m[1][0] * c3 + m[1][1] * c1) - m[1][2] * c0) * id;
    r.m[3][1] = ((m[0][0] * c3 - m[0][1] * c1) + m[0][2] * c0) * id;
    r.m[3][2] = ((-m[3][0] * s3 + m[3][1] * s1) - m[3][2] * s0) * id;
    r.m[3][3] = ((m[2][0] * s3 - m[2][1] * s1) + m[2][2] * s0) * id;
    return r;
}

/* vek builders compose by PRE-multiplication: m.scaled_3d(s) = S*m etc. (SURVEY App.B.1,
 * pinned by bounding_box.rs:184-195). */
static inline po_mat4 po_scaling(po_vec3 s) {
    po_mat4 r = po_identity(); r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r;
}
static inline po_mat4 po_translation(po_vec3 t) {
    po_mat4 r = po_identity(); r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z; return r;
}
static inline po_mat4 po_rotation_x(double a) {
    double c = cos(a), s = sin(a); po_mat4 r = po_identity();
    r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; return r;
}
static inline po_mat4 po_rotation_y(double a) {
    double c = cos(a), s = sin(a); po_mat4 r = po_identity();
    r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; return r;
}
static inline po_mat4 po_rotation_z(double a) {
    double c = cos(a), s = sin(a); po_mat4 r = po_identity();
    r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; return r;
}
/* f64::to_radians: self * (PI / 180.0) */
static inline double po_to_radians(double deg) { return deg * (3.14159265358979323846 / 180.0); }

/* roots 0.0.5 find_roots_quadratic(a2, a1, a0) as called from math.rs:109-113.
 * Returns the number of roots (0..2), ascending in out[]. Restated from the crate's published
 * algorithm (textbook form; SURVEY App.B.2): a2 == 0 -> linear; disc < 0 -> none; disc == 0 ->
 * one root -a1/(2 a2); else (-a1 -/+ sqrt(disc)) / (2 a2), ordered ascending. */
static inline int po_quadratic(double a2, double a1, double a0, double out[2]) {
    if (a2 == 0.0) {
        if (a1 == 0.0) {
            if (a0 == 0.0) { out[0] = 0.0; return 1; }
            return 0;
        }
        out[0] = -a0 / a1;
        return 1;
    }
    double disc = a1 * a1 - 4.0 * a2 * a0;
    if (disc < 0.0) return 0;
    double a2x2 = 2.0 * a2;
    if (disc == 0.0) { out[0] = -a1 / a2x2; return 1; }
    double sq = sqrt(disc);
    double x1 = (-a1 - sq) / a2x2;
    double x2 = (-a1 + sq) / a2x2;
    if (x1 < x2) { out[0] = x1; out[1] = x2; } else { out[0] = x2; out[1] = x1; }
    return 2;
}

/* math.rs:94-96 Solutions::find_in_range: first root (ascending) with range.contains(t). */
static inline int po_first_root_in_range(double a, double b, double c, const po_range *r, double *t) {
    double roots[2];
    int n = po_quadratic(a, b, c, roots);
    for (int i = 0; i < n; i++)
        if (po_contains(r, roots[i])) { *t = roots[i]; return 1; }
    return 0;
}

/* Sampling contract the build defines because the reference cannot be seeded
 * (render.rs:36-43 uses thread_rng; SURVEY D6 / App.B.4): a counter-based generator keyed by
 * (seed, pixel index, sample index, draw index). Two rounds of the splitmix64 finaliser over the
 * packed key; f64 = (u64 >> 11) * 2^-53 exactly as rand 0.7's Standard f64 distribution. */
static inline uint64_t po_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
static inline double po_rng_f64(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t draw) {
    uint64_t k = po_mix64(seed + 0x9e3779b97f4a7c15ULL);
    k = po_mix64(k ^ (pixel * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL));
    k = po_mix64(k ^ (((uint64_t)sample << 32) | (uint64_t)draw));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

#endif
