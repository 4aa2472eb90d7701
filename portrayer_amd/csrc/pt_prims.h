// Ray / primitive intersection in each primitive's model space.
//
// Replaces the reference's `Primitive::ray_hit` dispatch and the per-primitive tests
// (src/primitive.rs:55-62, src/primitive/{sphere,cube,plane,cylinder,cone,triangle,
// infinite_plane}.rs) and `BoundingBox::test_hit` (src/bounding_box.rs:104-116).
//
// A test returns only the ray parameter t and a small `part` tag (which cube face / cylinder or
// cone part was hit). The hit point and normal are rebuilt for the single winning candidate by
// pt_prim_surface() with the same expressions the reference evaluates at hit time, so they carry
// the same bits; texture coordinates and TBN frames (sphere.rs:53-96, cube.rs:84-139) are not
// computed: untextured materials never read them (SURVEY App.D Q9).
#pragma once

#include "pt_math.h"

// primitive.rs:67-81
enum { PT_SPHERE = 0, PT_TRIANGLE = 1, PT_MESH = 2, PT_KDMESH = 3, PT_PLANE = 4, PT_CUBE = 5, PT_CYLINDER = 6, PT_CONE = 7 };

struct PtRay {
    PtVec3 o, d;
};

PT_HD PtVec3 pt_ray_at(const PtRay& r, double t) { return r.o + r.d * t; }  // ray.rs:125-127

// ray.rs:130-135: origin as a point, direction as a direction, NOT renormalised, so t means the
// same in world and model space. `m` = rows 0..2 of the inverse transform.
PT_HD PtRay pt_ray_to_local(const double* m, const PtRay& r) {
    PtRay out;
    out.o = pt_xform_point(m, r.o);
    out.d = pt_xform_dir(m, 4, r.d);
    return out;
}

// cube.rs:20-27
PT_HD bool pt_cube_contains(PtVec3 p) {
    const double radius = 0.5 + PT_EPSILON;
    return -radius <= p.x && p.x <= radius && -radius <= p.y && p.y <= radius && -radius <= p.z && p.z <= radius;
}

// infinite_plane.rs:48-79 for an axis-aligned face of the unit cube: normal = sign * e_axis,
// point = sign * 0.5 * e_axis. The reference evaluates full 3-term dot products whose other two
// terms are products with 0.0; adding those zeros cannot change a non-zero sum and the sign of a
// zero sum does not matter to the comparisons below, so only the live term is kept.
PT_HD double pt_face_t(const PtRay& r, int axis, double sign) {
    double o = axis == 0 ? r.o.x : (axis == 1 ? r.o.y : r.o.z);
    double d = axis == 0 ? r.d.x : (axis == 1 ? r.d.y : r.d.z);
    double dot_dir_normal = d * sign;
    return -((o - sign * 0.5) * sign) / dot_dir_normal;
}

// cube.rs:38-83: faces in order +x, -x, +y, -y, +z, -z with a shrinking range; part = face index
PT_HD bool pt_cube_hit(const PtRay& r, double start, double end, double* t_out, uint32_t* part) {
    bool found = false;
#pragma unroll
    for (int f = 0; f < 6; f++) {
        double t = pt_face_t(r, f >> 1, (f & 1) ? -1.0 : 1.0);
        if (pt_in_range(start, end, t) && pt_cube_contains(pt_ray_at(r, t))) {
            end = t;
            *t_out = t;
            *part = (uint32_t)f;
            found = true;
        }
    }
    return found;
}

// plane.rs:25-53
PT_HD bool pt_plane_hit(const PtRay& r, double start, double end, double* t_out) {
    double t = -(r.o.y) / r.d.y;  // normal (0,1,0), point 0: -((o - 0).n) / (d.n)
    if (!pt_in_range(start, end, t)) return false;
    PtVec3 p = pt_ray_at(r, t);
    const double radius = 0.5 + PT_EPSILON;
    if (!(-radius <= p.x && p.x <= radius && -radius <= p.z && p.z <= radius)) return false;
    *t_out = t;
    return true;
}

// sphere.rs:26-52
PT_HD bool pt_sphere_hit(const PtRay& r, double start, double end, double* t_out) {
    double a = pt_dot(r.d, r.d);
    double b = 2.0 * pt_dot(r.o, r.d);
    double c = pt_dot(r.o, r.o) - 1.0;
    return pt_first_root(a, b, c, start, end, t_out);
}

// cylinder.rs:28-154: body, then top cap, then bottom cap; part 0 / 1 / 2
PT_HD bool pt_cylinder_hit(const PtRay& r, double start, double end, double* t_out, uint32_t* part) {
    bool found = false;
    double t;
    {
        double a = r.d.x * r.d.x + r.d.z * r.d.z;
        double b = 2.0 * r.o.x * r.d.x + 2.0 * r.o.z * r.d.z;
        double c = r.o.x * r.o.x + r.o.z * r.o.z - 0.25;
        if (pt_first_root(a, b, c, start, end, &t)) {  // first root only (quirk Q1)
            double y = r.o.y + r.d.y * t;
            if (!(y > 0.5 || y < -0.5)) { end = t; *t_out = t; *part = 0; found = true; }
        }
    }
    t = (0.5 - r.o.y) / r.d.y;
    if (pt_in_range(start, end, t)) {
        PtVec3 p = pt_ray_at(r, t);
        if (!((p.x * p.x + p.z * p.z) > 0.25)) { end = t; *t_out = t; *part = 1; found = true; }
    }
    t = (-0.5 - r.o.y) / r.d.y;
    if (pt_in_range(start, end, t)) {
        PtVec3 p = pt_ray_at(r, t);
        if (!((p.x * p.x + p.z * p.z) > 0.25)) { *t_out = t; *part = 2; found = true; }
    }
    return found;
}

// cone.rs:28-187: body (double-cone quadratic, first root only), then base cap; part 0 / 1
PT_HD bool pt_cone_hit(const PtRay& r, double start, double end, double* t_out, uint32_t* part) {
    bool found = false;
    double t;
    {
        const double h_sqr = 1.0, r_sqr = 0.25, HEIGHT = 1.0;
        PtVec3 o = r.o, d = r.d;
        double a = 4.0 * d.y * d.y * r_sqr - 4.0 * h_sqr * (d.x * d.x + d.z * d.z);
        double b = -8.0 * h_sqr * (d.x * o.x + d.z * o.z) - 4.0 * r_sqr * (d.y * HEIGHT - 2.0 * d.y * o.y);
        double c = -4.0 * h_sqr * (o.x * o.x + o.z * o.z) + r_sqr * (h_sqr - 4.0 * HEIGHT * o.y + 4.0 * o.y * o.y);
        if (pt_first_root(a, b, c, start, end, &t)) {
            double y = o.y + d.y * t;
            if (!(y > 0.5 || y < -0.5)) { end = t; *t_out = t; *part = 0; found = true; }
        }
    }
    t = (-0.5 - r.o.y) / r.d.y;
    if (pt_in_range(start, end, t)) {
        PtVec3 p = pt_ray_at(r, t);
        if (!((p.x * p.x + p.z * p.z) > 0.25)) { *t_out = t; *part = 1; found = true; }
    }
    return found;
}

// All five analytic primitives in ONE instruction stream, so that a wavefront whose lanes hold
// different primitive types does not run five separate routines one after the other.
//
// Every unit primitive is a subset of the same two kinds of sub-test, taken in the reference's order
// with a shrinking range end:
//   1. a quadric body: sphere (sphere.rs:48-52), cylinder side (cylinder.rs:44-60), cone side
//      (cone.rs:58-76) — only the three coefficients differ, the root search is shared;
//   2. up to six axis planes +x, -x, +y, -y, +z, -z at +-0.5: the cube's faces (cube.rs:46-83), the
//      cylinder's caps (cylinder.rs:94-104), the cone's base (cone.rs:131-148), and the Plane
//      primitive itself (plane.rs:35-53, the +y slot at height 0). Each is t = (h - o_a) / d_a, which
//      is bit-identical to the reference's -((o - P).n) / (d.n) for an axis-aligned unit normal
//      (x - y == -(y - x) and x / -y == -(x / y) exactly), followed by the type's own acceptance test.
// `part` is the tag pt_prim_surface() expects.
//
// (A version without per-lane branches - every slot's arithmetic for every lane, selects instead of `continue` - was measured in
// round 3: fewer scalar instructions, more vector ones, big-scene 34.4 -> 33.3 Gray/s, KD 35.9 -> 36.2 ms. Not kept.)
PT_HD bool pt_unit_prim_hit(uint32_t type, const PtRay& r, double start, double end, double* t_out, uint32_t* part) {
    bool found = false;
    if (type == PT_SPHERE || type == PT_CYLINDER || type == PT_CONE) {
        double a, b, c;
        if (type == PT_SPHERE) {
            a = pt_dot(r.d, r.d);
            b = 2.0 * pt_dot(r.o, r.d);
            c = pt_dot(r.o, r.o) - 1.0;
        } else if (type == PT_CYLINDER) {
            a = r.d.x * r.d.x + r.d.z * r.d.z;
            b = 2.0 * r.o.x * r.d.x + 2.0 * r.o.z * r.d.z;
            c = r.o.x * r.o.x + r.o.z * r.o.z - 0.25;
        } else {
            const double h_sqr = 1.0, r_sqr = 0.25, HEIGHT = 1.0;
            PtVec3 o = r.o, d = r.d;
            a = 4.0 * d.y * d.y * r_sqr - 4.0 * h_sqr * (d.x * d.x + d.z * d.z);
            b = -8.0 * h_sqr * (d.x * o.x + d.z * o.z) - 4.0 * r_sqr * (d.y * HEIGHT - 2.0 * d.y * o.y);
            c = -4.0 * h_sqr * (o.x * o.x + o.z * o.z) + r_sqr * (h_sqr - 4.0 * HEIGHT * o.y + 4.0 * o.y * o.y);
        }
        double t;
        if (pt_first_root(a, b, c, start, end, &t)) {  // first root only (quirk Q1)
            double y = r.o.y + r.d.y * t;
            if (type == PT_SPHERE || !(y > 0.5 || y < -0.5)) { end = t; *t_out = t; *part = 0; found = true; }
        }
    }
    // slots the type uses, bit k = slot k (+x, -x, +y, -y, +z, -z)
    const uint32_t slots = type == PT_CUBE ? 0x3Fu : (type == PT_CYLINDER ? 0x0Cu : (type == PT_CONE ? 0x08u : (type == PT_PLANE ? 0x04u : 0u)));
    const double radius = 0.5 + PT_EPSILON;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (!(slots & (1u << k))) continue;
        const int axis = k >> 1;
        double h = (k & 1) ? -0.5 : 0.5;
        if (type == PT_PLANE) h = 0.0;
        double o = axis == 0 ? r.o.x : (axis == 1 ? r.o.y : r.o.z);
        double d = axis == 0 ? r.d.x : (axis == 1 ? r.d.y : r.d.z);
        double t = (h - o) / d;
        if (!pt_in_range(start, end, t)) continue;
        PtVec3 p = pt_ray_at(r, t);
        bool ok;
        if (type == PT_CUBE) ok = -radius <= p.x && p.x <= radius && -radius <= p.y && p.y <= radius && -radius <= p.z && p.z <= radius;
        else if (type == PT_PLANE) ok = -radius <= p.x && p.x <= radius && -radius <= p.z && p.z <= radius;
        else ok = !((p.x * p.x + p.z * p.z) > 0.25);
        if (ok) {
            end = t; *t_out = t; found = true;
            *part = type == PT_CUBE ? (uint32_t)k : (type == PT_CYLINDER ? (uint32_t)(k - 1) : (type == PT_CONE ? 1u : 0u));
        }
    }
    return found;
}

// triangle.rs:38-80 (Cramer's rule; test order t, gamma, beta). v = a, b, c as 9 doubles.
PT_HD bool pt_triangle_hit(const double* v, const PtRay& r, double start, double end, double* t_out, double* beta_out, double* gamma_out) {
    double a = v[0] - v[3], b = v[1] - v[4], c = v[2] - v[5];
    double d = v[0] - v[6], e = v[1] - v[7], f = v[2] - v[8];
    double g = r.d.x, h = r.d.y, i = r.d.z;
    double j = v[0] - r.o.x, k = v[1] - r.o.y, l = v[2] - r.o.z;

    double ei_hf = e * i - h * f;
    double gf_di = g * f - d * i;
    double dh_eg = d * h - e * g;
    double m = a * ei_hf + b * gf_di + c * dh_eg;

    double ak_jb = a * k - j * b;
    double jc_al = j * c - a * l;
    double bl_ck = b * l - c * k;

    double t = -(f * ak_jb + e * jc_al + d * bl_ck) / m;
    if (!pt_in_range(start, end, t)) return false;
    double gamma = (i * ak_jb + h * jc_al + g * bl_ck) / m;
    if (gamma < 0.0 || gamma > 1.0) return false;
    double beta = (j * ei_hf + k * gf_di + l * dh_eg) / m;
    if (beta < 0.0 || beta > 1.0 - gamma) return false;
    *t_out = t;
    *beta_out = beta;
    *gamma_out = gamma;
    return true;
}

// The same test from the triangle's EDGE record (PtSceneView::tri_e: corner a, then a - b and a - c, the six differences above
// taken once on the host - the same IEEE subtractions, the same bits). In the walks that hold the record in scalar registers this
// saves the six subtractions and the moves that bring their second operands into vector registers, for every triangle tested.
PT_HD bool pt_triangle_hit_e(const double* te, const PtRay& r, double start, double end, double* t_out, double* beta_out, double* gamma_out) {
    double a = te[3], b = te[4], c = te[5];
    double d = te[6], e = te[7], f = te[8];
    double g = r.d.x, h = r.d.y, i = r.d.z;
    double j = te[0] - r.o.x, k = te[1] - r.o.y, l = te[2] - r.o.z;

    double ei_hf = e * i - h * f;
    double gf_di = g * f - d * i;
    double dh_eg = d * h - e * g;
    double m = a * ei_hf + b * gf_di + c * dh_eg;

    double ak_jb = a * k - j * b;
    double jc_al = j * c - a * l;
    double bl_ck = b * l - c * k;

    // (Sharing one refined reciprocal between the three divisions - pt_math.h's short division - was measured and is SLOWER here: big-soup
    // -2.7 %, macho-cows -4 %, profiles/r04/c35: the reciprocal and three numerators stay live across the early exits.)
    double t = -(f * ak_jb + e * jc_al + d * bl_ck) / m;
    if (!pt_in_range(start, end, t)) return false;
    double gamma = (i * ak_jb + h * jc_al + g * bl_ck) / m;
    if (gamma < 0.0 || gamma > 1.0) return false;
    double beta = (j * ei_hf + k * gf_di + l * dh_eg) / m;
    if (beta < 0.0 || beta > 1.0 - gamma) return false;
    *t_out = t;
    *beta_out = beta;
    *gamma_out = gamma;
    return true;
}

// bounding_box.rs:104-116: `inv` = rows 0..2 of the box's unit-cube inverse transform
PT_HD bool pt_bbox_test_hit(const double* inv, const PtRay& r, double start, double end) {
    PtRay local = pt_ray_to_local(inv, r);
    if (pt_cube_contains(pt_ray_at(local, start))) return true;
    double t; uint32_t part;
    return pt_cube_hit(local, start, end, &t, &part);
}

// Model-space hit point and (un-normalised) normal of an analytic primitive at parameter t.
PT_HD void pt_prim_surface(uint32_t type, uint32_t part, const PtRay& local, double t, PtVec3* p_out, PtVec3* n_out) {
    PtVec3 p = pt_ray_at(local, t);
    PtVec3 n;
    switch (type) {
    case PT_SPHERE: n = p; break;                                   // sphere.rs:64-66
    case PT_PLANE: n = pt_v3(0.0, 1.0, 0.0); break;                 // plane.rs:37
    case PT_CUBE: {                                                 // cube.rs:46-66 face normals
        double s = (part & 1) ? -1.0 : 1.0;
        uint32_t ax = part >> 1;
        n = pt_v3(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
        break;
    }
    case PT_CYLINDER:                                               // cylinder.rs:66-68, :108-110
        n = part == 0 ? pt_v3(p.x, 0.0, p.z) : pt_v3(0.0, part == 1 ? 1.0 : -1.0, 0.0);
        break;
    default: {                                                      // PT_CONE: cone.rs:99-104, :146-148
        if (part == 0) {
            PtVec3 tangent1 = pt_v3(0.0, 0.5, 0.0) - p;
            PtVec3 opposite = pt_v3(-p.x, p.y, -p.z);
            PtVec3 across = opposite - p;
            PtVec3 tangent2 = pt_cross(tangent1, across);
            n = pt_cross(tangent1, tangent2);
        } else {
            n = pt_v3(0.0, -1.0, 0.0);
        }
        break;
    }
    }
    *p_out = p;
    *n_out = n;
}
