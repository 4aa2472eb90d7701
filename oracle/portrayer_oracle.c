/* ORACLE — TEST INFRASTRUCTURE ONLY (rules and parity status: po_math.h, portrayer_oracle.h).
 *
 * Every function cites the reference file:line it restates. The reference is
 * /root/reference (sunjay/portrayer, Rust); nothing here is copied from it — it is Rust, this is
 * C — but the operation ORDER of each expression follows the Rust source so that results agree
 * bit for bit wherever third-party arithmetic (vek / roots / libm) is not involved.
 */
#define _GNU_SOURCE
#include "portrayer_oracle.h"
#include "po_math.h"

#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#define PO_SAMPLE_CHUNK 8

typedef struct { po_vec3 o, d; } ray_t;          /* ray.rs:102-108 */
typedef struct { double m[3][3]; } po_mat3;
/* ray.rs:10-36: tex_coord / normal_map_transform are Option<..>: has_uv / has_tbn */
typedef struct { double t; po_vec3 p, n; int has_uv, has_tbn; double u, v; po_mat3 tbn; } hit_t;

static inline po_vec3 mat3_mul(const po_mat3 *m, po_vec3 v) { /* vek Mat3 * Vec3, row dot products left to right */
    return po_v3((m->m[0][0] * v.x + m->m[0][1] * v.y) + m->m[0][2] * v.z,
                 (m->m[1][0] * v.x + m->m[1][1] * v.y) + m->m[1][2] * v.z,
                 (m->m[2][0] * v.x + m->m[2][1] * v.y) + m->m[2][2] * v.z);
}
static inline po_mat3 mat3_from_cols(po_vec3 c0, po_vec3 c1, po_vec3 c2) { /* Mat3::from_col_arrays */
    po_mat3 m = {{{c0.x, c1.x, c2.x}, {c0.y, c1.y, c2.y}, {c0.z, c1.z, c2.z}}};
    return m;
}

/* ---------------------------------------------------------------------------------------------
 * ray.rs
 * ------------------------------------------------------------------------------------------- */
static inline po_vec3 ray_at(const ray_t *r, double t) { /* ray.rs:125-127 */
    return po_add(r->o, po_scale(r->d, t));
}
static inline ray_t ray_transformed(const ray_t *r, const po_mat4 *m) { /* ray.rs:130-135 */
    ray_t out;
    out.o = po_xform_point(m, r->o);
    out.d = po_xform_dir(m, r->d);
    return out;
}

/* ---------------------------------------------------------------------------------------------
 * primitive/infinite_plane.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct { po_vec3 normal, point; } iplane_t;

static inline int iplane_front(const iplane_t *pl, po_vec3 p) { /* infinite_plane.rs:27-35 which_side: >= 0 is Front */
    return po_dot(po_sub(p, pl->point), pl->normal) >= 0.0;
}
static inline int iplane_hit(const iplane_t *pl, const ray_t *ray, const po_range *range, hit_t *hit) { /* infinite_plane.rs:48-79 */
    double dot_dir_normal = po_dot(ray->d, pl->normal);
    double t = -po_dot(po_sub(ray->o, pl->point), pl->normal) / dot_dir_normal;
    if (!po_contains(range, t)) return 0;
    hit->t = t;
    hit->p = ray_at(ray, t);
    hit->n = pl->normal;
    hit->has_uv = hit->has_tbn = 0;
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * primitive/cube.rs, plane.rs, sphere.rs, cylinder.rs, cone.rs, triangle.rs
 * ------------------------------------------------------------------------------------------- */
static inline int cube_contains(po_vec3 p) { /* cube.rs:20-27 */
    double radius = 0.5 + PO_EPSILON;
    return -radius <= p.x && p.x <= radius && -radius <= p.y && p.y <= radius && -radius <= p.z && p.z <= radius;
}

static int cube_hit(const ray_t *ray, const po_range *init, hit_t *out) { /* cube.rs:38-83; uv/TBN (:84-139) unused by untextured materials */
    static const iplane_t faces[6] = { /* cube.rs:46-66: right, left, top, bottom, near, far */
        {{1, 0, 0}, {0.5, 0, 0}}, {{-1, 0, 0}, {-0.5, 0, 0}},
        {{0, 1, 0}, {0, 0.5, 0}}, {{0, -1, 0}, {0, -0.5, 0}},
        {{0, 0, 1}, {0, 0, 0.5}}, {{0, 0, -1}, {0, 0, -0.5}},
    };
    /* cube.rs:46-66: (uv axis direction, uv offset in the 4x3 cube map) per face */
    static const double uv_axis[6][2] = {{-1, 1}, {1, 1}, {1, -1}, {1, 1}, {1, 1}, {-1, 1}};
    static const double uv_off[6][2] = {{1.0 / 2.0, 1.0 / 3.0}, {0.0, 1.0 / 3.0}, {1.0 / 4.0, 0.0}, {1.0 / 4.0, 2.0 / 3.0}, {1.0 / 4.0, 1.0 / 3.0}, {3.0 / 4.0, 1.0 / 3.0}};
    po_range range = *init;
    int found = 0, face = 0;
    for (int f = 0; f < 6; f++) {
        hit_t h;
        if (iplane_hit(&faces[f], ray, &range, &h) && cube_contains(h.p)) {
            range.end = h.t;
            *out = h;
            face = f;
            found = 1;
        }
    }
    if (!found) return 0;
    /* cube.rs:84-139: computed once for the winning face */
    po_vec3 p = out->p, fn = faces[face].normal;
    double fu, fv;
    if (fn.x != 0.0) { fu = p.z; fv = p.y; } else if (fn.y != 0.0) { fu = p.x; fv = p.z; } else { fu = p.x; fv = p.y; }
    double nu = fu * uv_axis[face][0] + 0.5, nv = 0.5 - fv * uv_axis[face][1];
    out->has_uv = 1;
    out->u = nu / 4.0 + uv_off[face][0];
    out->v = nv / 3.0 + uv_off[face][1];
    po_vec3 to_top = po_normalized(po_sub(po_v3(0.0, 1.0, 0.0), p));
    if (fabs(to_top.x) < PO_EPSILON && fabs(to_top.z) < PO_EPSILON) {
        out->tbn = mat3_from_cols(po_v3(1, 0, 0), fn, fn.y > 0.0 ? po_v3(0, 0, 1) : po_v3(0, 0, -1));
    } else {
        po_vec3 ht = po_cross(to_top, fn);
        po_vec3 vt = po_cross(fn, ht);
        out->tbn = mat3_from_cols(ht, fn, vt);
    }
    out->has_tbn = 1;
    return 1;
}

static int plane_hit(const ray_t *ray, const po_range *range, hit_t *out) { /* plane.rs:25-53 */
    static const iplane_t pl = {{0, 1, 0}, {0, 0, 0}};
    hit_t h;
    if (!iplane_hit(&pl, ray, range, &h)) return 0;
    double radius = 0.5 + PO_EPSILON;
    if (!(-radius <= h.p.x && h.p.x <= radius && -radius <= h.p.z && h.p.z <= radius)) return 0;
    h.has_uv = 1; h.u = h.p.x + 0.5; h.v = h.p.z + 0.5; /* plane.rs:40-43 */
    h.has_tbn = 1; h.tbn = mat3_from_cols(po_v3(1, 0, 0), po_v3(0, 1, 0), po_v3(0, 0, 1)); /* plane.rs:46 */
    *out = h;
    return 1;
}

static int sphere_hit(const ray_t *ray, const po_range *range, hit_t *out) { /* sphere.rs:26-70; uv/TBN (:53-96) unused */
    double a = po_dot(ray->d, ray->d);
    double b = 2.0 * po_dot(ray->o, ray->d);
    double c = po_dot(ray->o, ray->o) - 1.0 * 1.0;
    double t;
    if (!po_first_root_in_range(a, b, c, range, &t)) return 0;
    out->t = t;
    out->p = ray_at(ray, t);
    out->n = out->p;
    /* sphere.rs:53-96 */
    const double PI = 3.14159265358979323846;
    po_vec3 p = out->p, normal = out->p;
    out->has_uv = 2; /* (2: from atan2 / acos - texel_at counts how close such coordinates come to a texel edge) */
    out->u = (PI + atan2(-p.z, p.x)) / (2.0 * PI);
    out->v = acos(p.y) / PI;
    po_vec3 to_top = po_normalized(po_sub(po_v3(0.0, 1.0, 0.0), p));
    if (fabs(to_top.x) < PO_EPSILON && fabs(to_top.z) < PO_EPSILON) {
        out->tbn = mat3_from_cols(po_v3(1, 0, 0), normal, normal.y > 0.0 ? po_v3(0, 0, 1) : po_v3(0, 0, -1));
    } else {
        po_vec3 ht = po_cross(to_top, normal);
        po_vec3 vt = po_cross(normal, ht);
        out->tbn = mat3_from_cols(ht, normal, vt);
    }
    out->has_tbn = 1;
    return 1;
}

static int cylinder_body(const ray_t *ray, const po_range *range, hit_t *out) { /* cylinder.rs:28-77 */
    po_vec3 o = ray->o, d = ray->d;
    double a = d.x * d.x + d.z * d.z;
    double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
    double c = o.x * o.x + o.z * o.z - 0.5 * 0.5;
    double t;
    if (!po_first_root_in_range(a, b, c, range, &t)) return 0;
    po_vec3 p = ray_at(ray, t);
    if (p.y > 0.5 || p.y < -0.5) return 0;
    out->t = t; out->p = p; out->n = po_v3(p.x, 0.0, p.z); out->has_uv = out->has_tbn = 0;
    return 1;
}
static int cylinder_cap(double height, const ray_t *ray, const po_range *range, hit_t *out) { /* cylinder.rs:80-119 */
    double t = (height - ray->o.y) / ray->d.y;
    if (!po_contains(range, t)) return 0;
    po_vec3 p = ray_at(ray, t);
    if ((p.x * p.x + p.z * p.z) > 0.5 * 0.5) return 0;
    out->t = t; out->p = p; out->n = po_v3(0.0, height / fabs(height), 0.0); out->has_uv = out->has_tbn = 0;
    return 1;
}
static int cylinder_hit(const ray_t *ray, const po_range *init, hit_t *out) { /* cylinder.rs:121-154 */
    po_range range = *init;
    int found = 0;
    hit_t h;
    if (cylinder_body(ray, &range, &h)) { range.end = h.t; *out = h; found = 1; }
    if (cylinder_cap(0.5, ray, &range, &h)) { range.end = h.t; *out = h; found = 1; }
    if (cylinder_cap(-0.5, ray, &range, &h)) { *out = h; found = 1; }
    return found;
}

static int cone_body(const ray_t *ray, const po_range *range, hit_t *out) { /* cone.rs:28-114 */
    po_vec3 o = ray->o, d = ray->d;
    const double HEIGHT = 1.0;
    double h_sqr = HEIGHT * HEIGHT;
    double r_sqr = 0.5 * 0.5;
    double a = 4.0 * d.y * d.y * r_sqr - 4.0 * h_sqr * (d.x * d.x + d.z * d.z);
    double b = -8.0 * h_sqr * (d.x * o.x + d.z * o.z) - 4.0 * r_sqr * (d.y * HEIGHT - 2.0 * d.y * o.y);
    double c = -4.0 * h_sqr * (o.x * o.x + o.z * o.z) + r_sqr * (h_sqr - 4.0 * HEIGHT * o.y + 4.0 * o.y * o.y);
    double t;
    if (!po_first_root_in_range(a, b, c, range, &t)) return 0; /* first root only: quirk Q1 */
    po_vec3 p = ray_at(ray, t);
    if (p.y > 0.5 || p.y < -0.5) return 0;
    po_vec3 tip = po_v3(0.0, 0.5, 0.0);
    po_vec3 tangent1 = po_sub(tip, p);
    po_vec3 opposite = po_v3(-p.x, p.y, -p.z);
    po_vec3 across = po_sub(opposite, p);
    po_vec3 tangent2 = po_cross(tangent1, across);
    out->t = t; out->p = p; out->n = po_cross(tangent1, tangent2); out->has_uv = out->has_tbn = 0;
    return 1;
}
static int cone_cap(const ray_t *ray, const po_range *range, hit_t *out) { /* cone.rs:117-157 */
    double height = -0.5;
    double t = (height - ray->o.y) / ray->d.y;
    if (!po_contains(range, t)) return 0;
    po_vec3 p = ray_at(ray, t);
    if ((p.x * p.x + p.z * p.z) > 0.5 * 0.5) return 0;
    out->t = t; out->p = p; out->n = po_v3(0.0, -1.0, 0.0); out->has_uv = out->has_tbn = 0;
    return 1;
}
static int cone_hit(const ray_t *ray, const po_range *init, hit_t *out) { /* cone.rs:159-187 */
    po_range range = *init;
    int found = 0;
    hit_t h;
    if (cone_body(ray, &range, &h)) { range.end = h.t; *out = h; found = 1; }
    if (cone_cap(ray, &range, &h)) { *out = h; found = 1; }
    return found;
}

typedef struct { po_vec3 a, b, c; int has_n; po_vec3 na, nb, nc; int has_uv; double uv[6]; } tri_t; /* triangle.rs:8-19 */

static int triangle_hit(const tri_t *tr, const ray_t *ray, const po_range *range, hit_t *out) { /* triangle.rs:38-147 */
    po_vec3 ab = po_sub(tr->a, tr->b), ac = po_sub(tr->a, tr->c), ao = po_sub(tr->a, ray->o);
    double a = ab.x, b = ab.y, c = ab.z;
    double d = ac.x, e = ac.y, f = ac.z;
    double g = ray->d.x, h = ray->d.y, i = ray->d.z;
    double j = ao.x, k = ao.y, l = ao.z;

    double ei_hf = e * i - h * f;
    double gf_di = g * f - d * i;
    double dh_eg = d * h - e * g;
    double m = a * ei_hf + b * gf_di + c * dh_eg;

    double ak_jb = a * k - j * b;
    double jc_al = j * c - a * l;
    double bl_ck = b * l - c * k;

    double t = -(f * ak_jb + e * jc_al + d * bl_ck) / m;
    if (!po_contains(range, t)) return 0;
    double gamma = (i * ak_jb + h * jc_al + g * bl_ck) / m;
    if (gamma < 0.0 || gamma > 1.0) return 0;
    double beta = (j * ei_hf + k * gf_di + l * dh_eg) / m;
    if (beta < 0.0 || beta > 1.0 - gamma) return 0;

    po_vec3 normal;
    if (tr->has_n) {
        double alpha = 1.0 - beta - gamma;
        normal = po_add(po_add(po_scale(tr->na, alpha), po_scale(tr->nb, beta)), po_scale(tr->nc, gamma));
    } else {
        normal = po_cross(po_sub(tr->b, tr->a), po_sub(tr->c, tr->a));
    }
    out->t = t;
    out->p = ray_at(ray, t);
    out->n = normal;
    out->has_uv = out->has_tbn = 0;
    if (tr->has_uv) { /* triangle.rs:90-138 */
        double ua = tr->uv[0], va = tr->uv[1], ub = tr->uv[2], vb = tr->uv[3], uc = tr->uv[4], vc = tr->uv[5];
        double alpha = 1.0 - beta - gamma;
        double uu = (ua * alpha + ub * beta) + uc * gamma, vv = (va * alpha + vb * beta) + vc * gamma;
        out->has_uv = 1; out->u = uu; out->v = 1.0 - vv;
        po_vec3 edge1 = po_sub(tr->b, tr->a), edge2 = po_sub(tr->c, tr->a);
        double du1 = ub - ua, dv1 = vb - va, du2 = uc - ua, dv2 = vc - va;
        po_vec3 tangent = po_v3(dv2 * edge1.x - dv1 * edge2.x, dv2 * edge1.y - dv1 * edge2.y, dv2 * edge1.z - dv1 * edge2.z);
        po_vec3 bitangent = po_v3(-du2 * edge1.x + du1 * edge2.x, -du2 * edge1.y + du1 * edge2.y, -du2 * edge1.z + du1 * edge2.z);
        double coeff = du1 * dv2 - du2 * dv1;
        tangent = po_normalized(po_divs(tangent, coeff));
        bitangent = po_normalized(po_divs(bitangent, coeff));
        out->tbn = mat3_from_cols(tangent, po_normalized(normal), bitangent);
        out->has_tbn = 1;
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------------
 * bounding_box.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct { po_vec3 min, max; po_mat4 invtrans; } bbox_t;

static bbox_t bbox_new(po_vec3 min, po_vec3 max) { /* bounding_box.rs:55-82 */
    bbox_t b;
    b.min = min; b.max = max;
    po_vec3 size = po_sub(max, min);
    size = po_vmax(size, po_v3(PO_EPSILON, PO_EPSILON, PO_EPSILON));
    po_vec3 center = po_divs(po_add(min, max), 2.0);
    po_mat4 s = po_scaling(size), tr = po_translation(center);
    po_mat4 trans = po_matmul(&tr, &s); /* scaling_3d(size).translated_3d(center) = T * S */
    b.invtrans = po_inverted(&trans);
    return b;
}
static inline double bbox_extent(const bbox_t *b) { /* bounding_box.rs:95-99: SQUARED diagonal (quirk Q3) */
    po_vec3 d = po_sub(b->max, b->min);
    return po_dot(d, d);
}
static int bbox_test_hit(const bbox_t *b, const ray_t *ray, const po_range *range) { /* bounding_box.rs:104-116 */
    ray_t local = ray_transformed(ray, &b->invtrans);
    if (cube_contains(ray_at(&local, range->start))) return 1;
    hit_t h;
    return cube_hit(&local, range, &h);
}
static void bbox_transform_minmax(const po_mat4 *m, po_vec3 bmin, po_vec3 bmax, po_vec3 *omin, po_vec3 *omax) { /* bounding_box.rs:123-148 */
    po_vec3 mn = po_v3(INFINITY, INFINITY, INFINITY), mx = po_v3(-INFINITY, -INFINITY, -INFINITY);
    double xs[2] = {bmin.x, bmax.x}, ys[2] = {bmin.y, bmax.y}, zs[2] = {bmin.z, bmax.z};
    for (int ix = 0; ix < 2; ix++) for (int iy = 0; iy < 2; iy++) for (int iz = 0; iz < 2; iz++) {
        po_vec3 v = po_xform_point(m, po_v3(xs[ix], ys[iy], zs[iz]));
        mn = po_vmin(mn, v);
        mx = po_vmax(mx, v);
    }
    *omin = mn; *omax = mx;
}

/* ---------------------------------------------------------------------------------------------
 * kdtree/leaf.rs + kdtree/node.rs, generic over what a leaf item is
 * ------------------------------------------------------------------------------------------- */
typedef struct kdnode {
    int is_leaf;
    iplane_t sep;          /* node.rs:15-16 */
    int axis;
    struct kdnode *front, *back;
    uint32_t *items; uint32_t n_items; /* leaf.rs:71-79 */
    po_vec3 bmin, bmax;    /* bounds of this node (leaf.rs:73 / node.rs:18) */
} kdnode_t;

typedef struct { int target_max_nodes; long target_max_merit; int max_tries; } kdconf_t; /* leaf.rs:55-67 */

static void items_bounds(const po_vec3 *mins, const po_vec3 *maxs, const uint32_t *items, uint32_t n, po_vec3 *omin, po_vec3 *omax) { /* bounding_box.rs:24-37 */
    if (n == 0) { *omin = po_v3(0, 0, 0); *omax = po_v3(0, 0, 0); return; }
    po_vec3 mn = mins[items[0]], mx = maxs[items[0]];
    for (uint32_t i = 1; i < n; i++) { mn = po_vmin(mn, mins[items[i]]); mx = po_vmax(mx, maxs[items[i]]); }
    *omin = mn; *omax = mx;
}

enum { PART_FRONT, PART_BACK, PART_SHARED };
static inline int partition_item(const po_vec3 *mins, const po_vec3 *maxs, uint32_t it, const iplane_t *sep) { /* leaf.rs:115-130 */
    int fmin = iplane_front(sep, mins[it]), fmax = iplane_front(sep, maxs[it]);
    if (fmin && fmax) return PART_FRONT;
    if (!fmin && !fmax) return PART_BACK;
    return PART_SHARED;
}

/* leaf.rs:89-231 KDLeaf::partitioned. Takes ownership of items. */
static kdnode_t *kd_partitioned(const po_vec3 *mins, const po_vec3 *maxs, uint32_t *items, uint32_t n,
                                po_vec3 bmin, po_vec3 bmax, po_vec3 axis, int max_depth, kdconf_t conf) {
    kdnode_t *node = (kdnode_t *)calloc(1, sizeof *node);
    node->bmin = bmin; node->bmax = bmax;
    if (max_depth == 0 || n <= (uint32_t)conf.target_max_nodes) {
        node->is_leaf = 1; node->items = items; node->n_items = n;
        return node;
    }
    po_vec3 min_axis = po_mul(axis, bmin), max_axis = po_mul(axis, bmax); /* leaf.rs:135-136 */
    iplane_t sep;
    sep.normal = axis;
    sep.point = po_add(min_axis, po_divs(po_sub(max_axis, min_axis), 2.0)); /* leaf.rs:139-142 */
    po_vec3 plane_min = min_axis, plane_max = max_axis; /* leaf.rs:147 */
    for (int tries = 0; tries < conf.max_tries; tries++) { /* leaf.rs:155-201 */
        long front = 0, back = 0, shared = 0;
        for (uint32_t i = 0; i < n; i++) {
            int p = partition_item(mins, maxs, items[i], &sep);
            if (p == PART_FRONT) front++; else if (p == PART_BACK) back++; else shared++;
        }
        long merit = labs(front - back) + shared;
        if (merit <= conf.target_max_merit) break;
        if (front > back) {
            plane_min = sep.point;
            sep.point = po_add(sep.point, po_divs(po_sub(plane_max, sep.point), 2.0));
        } else {
            plane_max = sep.point;
            sep.point = po_add(plane_min, po_divs(po_sub(sep.point, plane_min), 2.0));
        }
    }
    uint32_t *fi = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1)), *bi = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t nf = 0, nb = 0;
    for (uint32_t i = 0; i < n; i++) { /* leaf.rs:204-215 */
        int p = partition_item(mins, maxs, items[i], &sep);
        if (p == PART_FRONT) fi[nf++] = items[i];
        else if (p == PART_BACK) bi[nb++] = items[i];
        else { fi[nf++] = items[i]; bi[nb++] = items[i]; }
    }
    free(items);
    po_vec3 next = po_v3(axis.z, axis.x, axis.y); /* leaf.rs:97-103 next_axis */
    po_vec3 fmin, fmax, kmin, kmax;
    items_bounds(mins, maxs, fi, nf, &fmin, &fmax);
    items_bounds(mins, maxs, bi, nb, &kmin, &kmax);
    node->is_leaf = 0;
    node->sep = sep;
    node->axis = axis.x != 0.0 ? 0 : (axis.y != 0.0 ? 1 : 2);
    node->front = kd_partitioned(mins, maxs, fi, nf, fmin, fmax, next, max_depth - 1, conf);
    node->back = kd_partitioned(mins, maxs, bi, nb, kmin, kmax, next, max_depth - 1, conf);
    return node;
}

static void kd_free(kdnode_t *n) {
    if (!n) return;
    if (n->is_leaf) free(n->items); else { kd_free(n->front); kd_free(n->back); }
    free(n);
}

/* Leaf callback: must follow RayCast's contract (node.rs:37-39): on a hit, range->end = t. */
typedef int (*kd_leaf_fn)(void *ctx, const uint32_t *items, uint32_t n, const ray_t *ray, po_range *range, hit_t *hit, int32_t *id);

static int kd_axis_plane_t(const iplane_t *sep, const ray_t *ray, const po_range *range, double *t_out) { /* node.rs:90-109 */
    po_vec3 pv = po_mul(sep->normal, sep->point), ro = po_mul(sep->normal, ray->o), rd = po_mul(sep->normal, ray->d);
    double plane_value = (pv.x + pv.y) + pv.z;
    double ray_origin = (ro.x + ro.y) + ro.z;
    double ray_direction = (rd.x + rd.y) + rd.z;
    double t = (plane_value - ray_origin) / ray_direction;
    if (po_contains(range, t)) { *t_out = t; return 1; }
    return 0;
}

static int kd_cast(const kdnode_t *node, const ray_t *ray, po_range *range, double extent, kd_leaf_fn leaf, void *ctx,
                   hit_t *hit, int32_t *id, po_stats *st) { /* node.rs:66-203 ray_cast_impl */
    if (node->is_leaf) {
        st->n_leaf++;
        return leaf(ctx, node->items, node->n_items, ray, range, hit, id);
    }
    st->n_split++;
    double t_max = range->start + extent;
    if (!po_contains(range, t_max)) t_max = range->end - PO_EPSILON;
    double t_min = range->start + PO_EPSILON;
    po_vec3 ray_start = ray_at(ray, t_min), ray_end = ray_at(ray, t_max);
    int s = iplane_front(&node->sep, ray_start), e = iplane_front(&node->sep, ray_end);
    if (s && e) return kd_cast(node->front, ray, range, extent, leaf, ctx, hit, id, st);
    if (!s && !e) return kd_cast(node->back, ray, range, extent, leaf, ctx, hit, id, st);
    double plane_t;
    if (!kd_axis_plane_t(&node->sep, ray, range, &plane_t)) {
        /* node.rs:146-147 / :177-178: the reference panics here ("bug: ray should definitely hit
         * infinite plane"). Counted; reported as a miss (quirk Q4). */
        st->kd_plane_miss++;
        return 0;
    }
    const kdnode_t *first = s ? node->front : node->back, *second = s ? node->back : node->front;
    po_range r1 = {range->start, plane_t};
    if (kd_cast(first, ray, &r1, extent, leaf, ctx, hit, id, st)) { *range = r1; return 1; }
    po_range r2 = {plane_t, range->end};
    if (kd_cast(second, ray, &r2, extent, leaf, ctx, hit, id, st)) { *range = r2; return 1; }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * primitive/mesh.rs + kdtree/kdmesh.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t n_verts, n_tris;
    const double *pos, *nrm, *uv; const uint32_t *idx;
    int has_normals, has_uv;
    bbox_t bounds;              /* mesh.rs:69-75 */
    kdnode_t *kd[2];            /* kdmesh.rs:37-58, per Shading (0 flat, 1 smooth) */
    bbox_t kd_bounds[2];        /* root bounds incl. invtrans (kdmesh.rs:26-30) */
} mesh_t;

static inline po_vec3 ld3(const double *p) { return po_v3(p[0], p[1], p[2]); }

static tri_t mesh_triangle(const mesh_t *m, uint32_t i, int smooth) { /* mesh.rs:97-115 */
    tri_t t;
    uint32_t a = m->idx[3 * i], b = m->idx[3 * i + 1], c = m->idx[3 * i + 2];
    t.a = ld3(m->pos + 3 * a); t.b = ld3(m->pos + 3 * b); t.c = ld3(m->pos + 3 * c);
    t.has_n = smooth;
    if (smooth) { t.na = ld3(m->nrm + 3 * a); t.nb = ld3(m->nrm + 3 * b); t.nc = ld3(m->nrm + 3 * c); }
    t.has_uv = m->has_uv; /* mesh.rs:106-110 */
    if (m->has_uv) { t.uv[0] = m->uv[2 * a]; t.uv[1] = m->uv[2 * a + 1]; t.uv[2] = m->uv[2 * b]; t.uv[3] = m->uv[2 * b + 1]; t.uv[4] = m->uv[2 * c]; t.uv[5] = m->uv[2 * c + 1]; }
    return t;
}

static int mesh_hit(const mesh_t *m, int smooth, const ray_t *ray, const po_range *init, hit_t *out, po_stats *st) { /* mesh.rs:146-167 */
    st->n_bbox++;
    if (!bbox_test_hit(&m->bounds, ray, init)) return 0;
    po_range range = *init;
    int found = 0;
    for (uint32_t i = 0; i < m->n_tris; i++) {
        tri_t t = mesh_triangle(m, i, smooth);
        hit_t h;
        st->n_tri++;
        if (triangle_hit(&t, ray, &range, &h)) { range.end = h.t; *out = h; found = 1; }
    }
    return found;
}

typedef struct { const mesh_t *mesh; int smooth; po_stats *st; } kdmesh_ctx_t;
static int kdmesh_leaf(void *vctx, const uint32_t *items, uint32_t n, const ray_t *ray, po_range *range, hit_t *hit, int32_t *id) { /* node.rs:38-49 closure + ray.rs:50-63 */
    kdmesh_ctx_t *c = (kdmesh_ctx_t *)vctx;
    po_range r = *range;
    int found = 0;
    for (uint32_t i = 0; i < n; i++) {
        tri_t t = mesh_triangle(c->mesh, items[i], c->smooth);
        hit_t h;
        c->st->n_tri++;
        if (triangle_hit(&t, ray, &r, &h)) { r.end = h.t; *hit = h; *id = (int32_t)items[i]; found = 1; }
    }
    if (found) range->end = hit->t;
    return found;
}
static void kdmesh_build(mesh_t *m, int smooth, int depth) { /* kdmesh.rs:37-58 */
    if (m->kd[smooth]) return;
    po_vec3 *mins = (po_vec3 *)malloc(sizeof(po_vec3) * (m->n_tris ? m->n_tris : 1)), *maxs = (po_vec3 *)malloc(sizeof(po_vec3) * (m->n_tris ? m->n_tris : 1));
    uint32_t *items = (uint32_t *)malloc(sizeof(uint32_t) * (m->n_tris ? m->n_tris : 1));
    for (uint32_t i = 0; i < m->n_tris; i++) { /* triangle.rs:29-36 */
        tri_t t = mesh_triangle(m, i, 0);
        mins[i] = po_vmin(t.a, po_vmin(t.b, t.c));
        maxs[i] = po_vmax(t.a, po_vmax(t.b, t.c));
        items[i] = i;
    }
    po_vec3 bmin, bmax;
    items_bounds(mins, maxs, items, m->n_tris, &bmin, &bmax);
    kdconf_t conf = {3, 3, 10};
    m->kd[smooth] = kd_partitioned(mins, maxs, items, m->n_tris, bmin, bmax, po_v3(1, 0, 0), depth, conf);
    m->kd_bounds[smooth] = bbox_new(bmin, bmax);
    free(mins); free(maxs);
}
static int kdmesh_hit(const mesh_t *m, int smooth, const ray_t *ray, const po_range *init, hit_t *out, po_stats *st) { /* kdmesh.rs:62-74 + node.rs:33-51 */
    st->n_bbox++;
    if (!bbox_test_hit(&m->kd_bounds[smooth], ray, init)) return 0;
    po_range range = *init;
    kdmesh_ctx_t c = {m, smooth, st};
    int32_t id;
    return kd_cast(m->kd[smooth], ray, &range, bbox_extent(&m->kd_bounds[smooth]), kdmesh_leaf, &c, out, &id, st);
}

/* ---------------------------------------------------------------------------------------------
 * Scene context: scene.rs, flat_scene.rs, kdtree/kdscene.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct { int type, data, flags, material; po_mat4 trans, inv, nrm; po_vec3 bmin, bmax; } flatnode_t; /* flat_scene.rs:50-61 */

typedef struct {
    const po_scene *sc;
    mesh_t *meshes;
    po_mat4 *h_inv, *h_nrm;   /* per hierarchical node: invtrans, normal_trans (scene.rs:201-205) */
    flatnode_t *flat; uint32_t n_flat;
    kdnode_t *kd_root; double kd_extent; po_vec3 kd_bmin, kd_bmax;
    int mode;
} ctx_t;

static inline po_mat4 ldm(const double *p) { po_mat4 m; memcpy(m.m, p, sizeof m.m); return m; }

static int prim_hit(const ctx_t *cx, int type, int data, int flags, const ray_t *ray, const po_range *range, hit_t *out, po_stats *st) { /* primitive.rs:55-62 */
    switch (type) {
    case PO_SPHERE: return sphere_hit(ray, range, out);
    case PO_TRIANGLE: {
        tri_t t;
        const double *v = cx->sc->tri_vertices + 9 * (size_t)data;
        t.a = ld3(v); t.b = ld3(v + 3); t.c = ld3(v + 6);
        t.has_n = cx->sc->tri_has_normals && cx->sc->tri_has_normals[data];
        if (t.has_n) { const double *n = cx->sc->tri_normals + 9 * (size_t)data; t.na = ld3(n); t.nb = ld3(n + 3); t.nc = ld3(n + 6); }
        t.has_uv = cx->sc->tri_has_texcoords && cx->sc->tri_texcoords && cx->sc->tri_has_texcoords[data];
        if (t.has_uv) memcpy(t.uv, cx->sc->tri_texcoords + 6 * (size_t)data, sizeof t.uv);
        st->n_tri++;
        return triangle_hit(&t, ray, range, out);
    }
    case PO_MESH: return mesh_hit(&cx->meshes[data], flags & 1, ray, range, out, st);
    case PO_KDMESH: return kdmesh_hit(&cx->meshes[data], flags & 1, ray, range, out, st);
    case PO_PLANE: return plane_hit(ray, range, out);
    case PO_CUBE: return cube_hit(ray, range, out);
    case PO_CYLINDER: return cylinder_hit(ray, range, out);
    case PO_CONE: return cone_hit(ray, range, out);
    default: return 0;
    }
}

static po_vec3 prim_bounds(const ctx_t *cx, int type, int data, int flags, po_vec3 *omax) { /* Bounds impls of each primitive */
    switch (type) {
    case PO_SPHERE: *omax = po_v3(1, 1, 1); return po_v3(-1, -1, -1);                     /* sphere.rs:18-24 */
    case PO_PLANE: *omax = po_v3(0.5, 0, 0.5); return po_v3(-0.5, 0, -0.5);               /* plane.rs:17-23 */
    case PO_MESH: *omax = cx->meshes[data].bounds.max; return cx->meshes[data].bounds.min; /* mesh.rs:125-129 */
    case PO_KDMESH: *omax = cx->meshes[data].kd_bounds[flags & 1].max; return cx->meshes[data].kd_bounds[flags & 1].min; /* kdmesh.rs:26-30 */
    case PO_TRIANGLE: {                                                                    /* triangle.rs:29-36 */
        const double *v = cx->sc->tri_vertices + 9 * (size_t)data;
        po_vec3 a = ld3(v), b = ld3(v + 3), c = ld3(v + 6);
        *omax = po_vmax(a, po_vmax(b, c));
        return po_vmin(a, po_vmin(b, c));
    }
    default: *omax = po_v3(0.5, 0.5, 0.5); return po_v3(-0.5, -0.5, -0.5);                /* cube.rs:30-36, cylinder.rs:18-24, cone.rs:18-24 */
    }
}

/* flat_scene.rs:71-99 FlatSceneNode::ray_cast */
static int flatnode_cast(const ctx_t *cx, uint32_t i, const ray_t *ray, po_range *range, hit_t *out, po_stats *st) {
    const flatnode_t *fn = &cx->flat[i];
    ray_t local = ray_transformed(ray, &fn->inv);
    hit_t h;
    st->n_analytic++;
    if (!prim_hit(cx, fn->type, fn->data, fn->flags, &local, range, &h, st)) return 0;
    h.p = po_xform_point(&fn->trans, h.p);
    h.n = po_xform_dir(&fn->nrm, h.n);
    range->end = h.t;
    *out = h;
    return 1;
}

typedef struct { const ctx_t *cx; po_stats *st; } kdscene_ctx_t;
static int kdscene_leaf(void *vctx, const uint32_t *items, uint32_t n, const ray_t *ray, po_range *range, hit_t *hit, int32_t *id) { /* node.rs:27-31 + ray.rs:87-99 */
    kdscene_ctx_t *c = (kdscene_ctx_t *)vctx;
    int found = 0;
    for (uint32_t i = 0; i < n; i++) {
        hit_t h;
        if (flatnode_cast(c->cx, items[i], ray, range, &h, c->st)) { range->end = h.t; *hit = h; *id = (int32_t)items[i]; found = 1; }
    }
    return found;
}

/* scene.rs:80-120 SceneNode::ray_cast (hierarchical, recursive) */
static int hier_cast(const ctx_t *cx, uint32_t node, const ray_t *ray, po_range *range, hit_t *out, int32_t *mat, po_stats *st) {
    const po_scene *sc = cx->sc;
    ray_t local = ray_transformed(ray, &cx->h_inv[node]);
    po_mat4 trans = ldm(sc->node_trans + 16 * (size_t)node);
    const po_mat4 *nrm = &cx->h_nrm[node];
    int found = 0;
    if (sc->node_prim_type[node] != PO_NONE) {
        hit_t h;
        st->n_analytic++;
        if (prim_hit(cx, sc->node_prim_type[node], sc->node_prim_data[node], sc->node_prim_flags[node], &local, range, &h, st)) {
            h.p = po_xform_point(&trans, h.p);
            h.n = po_xform_dir(nrm, h.n);
            range->end = h.t;
            *out = h; *mat = sc->node_material[node]; found = 1;
        }
    }
    /* children().ray_cast(&local_ray, t_range): ray.rs:87-99 fold */
    int cfound = 0; hit_t ch; int32_t cm = -1;
    for (uint32_t k = sc->node_child_off[node]; k < sc->node_child_off[node + 1]; k++) {
        hit_t h; int32_t m;
        if (hier_cast(cx, sc->children[k], &local, range, &h, &m, st)) { range->end = h.t; ch = h; cm = m; cfound = 1; }
    }
    if (cfound) {
        ch.p = po_xform_point(&trans, ch.p);
        ch.n = po_xform_dir(nrm, ch.n);
        *out = ch; *mat = cm; found = 1;
    }
    return found;
}

/* scene.root.ray_cast(ray, &mut t_range) in the mode's representation */
static int scene_cast(const ctx_t *cx, const ray_t *ray, po_range *range, hit_t *out, int32_t *mat, int32_t *id, po_stats *st) {
    *id = -1;
    if (cx->mode == PO_MODE_HIER) {
        int f = hier_cast(cx, cx->sc->root, ray, range, out, mat, st);
        if (f) *id = 0;
        return f;
    }
    if (cx->mode == PO_MODE_FLAT) { /* ray.rs:87-99 over Vec<FlatSceneNode> */
        int found = 0;
        for (uint32_t i = 0; i < cx->n_flat; i++) {
            hit_t h;
            if (flatnode_cast(cx, i, ray, range, &h, st)) { range->end = h.t; *out = h; *id = (int32_t)i; found = 1; }
        }
        if (found) *mat = cx->flat[*id].material;
        return found;
    }
    kdscene_ctx_t c = {cx, st};
    int f = kd_cast(cx->kd_root, ray, range, cx->kd_extent, kdscene_leaf, &c, out, id, st); /* node.rs:27-31 */
    if (f) *mat = cx->flat[*id].material;
    return f;
}

/* ---------------------------------------------------------------------------------------------
 * Context construction
 * ------------------------------------------------------------------------------------------- */
static void ctx_free(ctx_t *cx) {
    if (cx->meshes) {
        for (uint32_t i = 0; i < cx->sc->n_meshes; i++) { kd_free(cx->meshes[i].kd[0]); kd_free(cx->meshes[i].kd[1]); }
        free(cx->meshes);
    }
    free(cx->h_inv); free(cx->h_nrm); free(cx->flat);
    kd_free(cx->kd_root);
    memset(cx, 0, sizeof *cx);
}

static int ctx_flatten(ctx_t *cx) { /* flat_scene.rs:18-46: BFS with a queue of (parent_trans, node) */
    const po_scene *sc = cx->sc;
    size_t qcap = 1024, qh = 0, qt = 0, fcap = 256;
    struct qe { po_mat4 parent; uint32_t node; } *q = (struct qe *)malloc(qcap * sizeof *q);
    cx->flat = (flatnode_t *)malloc(fcap * sizeof *cx->flat);
    cx->n_flat = 0;
    q[qt].parent = po_identity(); q[qt].node = sc->root; qt++;
    while (qh < qt) {
        struct qe cur = q[qh++];
        po_mat4 nt = ldm(sc->node_trans + 16 * (size_t)cur.node);
        po_mat4 total = po_matmul(&cur.parent, &nt);
        if (sc->node_prim_type[cur.node] != PO_NONE) {
            if (cx->n_flat == fcap) { fcap *= 2; cx->flat = (flatnode_t *)realloc(cx->flat, fcap * sizeof *cx->flat); }
            flatnode_t *fn = &cx->flat[cx->n_flat++];
            fn->type = sc->node_prim_type[cur.node]; fn->data = sc->node_prim_data[cur.node];
            fn->flags = sc->node_prim_flags[cur.node]; fn->material = sc->node_material[cur.node];
            fn->trans = total;
            fn->inv = po_inverted(&total);          /* flat_scene.rs:103-108 */
            fn->nrm = po_transposed(&fn->inv);
            po_vec3 pmin, pmax;
            pmin = prim_bounds(cx, fn->type, fn->data, fn->flags, &pmax);
            bbox_transform_minmax(&total, pmin, pmax, &fn->bmin, &fn->bmax); /* flat_scene.rs:63-69 */
        }
        for (uint32_t k = sc->node_child_off[cur.node]; k < sc->node_child_off[cur.node + 1]; k++) {
            if (qt == qcap) {
                if (qt > (1u << 26)) { free(q); return -2; } /* cycle guard: the reference would never terminate */
                qcap *= 2; q = (struct qe *)realloc(q, qcap * sizeof *q);
            }
            q[qt].parent = total; q[qt].node = sc->children[k]; qt++;
        }
    }
    free(q);
    return 0;
}

static int ctx_init(ctx_t *cx, const po_scene *sc, int mode, int kd_depth, int kd_mesh_depth) {
    memset(cx, 0, sizeof *cx);
    cx->sc = sc; cx->mode = mode;
    if (kd_depth < 0) kd_depth = 10;
    if (kd_mesh_depth < 0) kd_mesh_depth = 10;
    cx->meshes = (mesh_t *)calloc(sc->n_meshes ? sc->n_meshes : 1, sizeof(mesh_t));
    for (uint32_t i = 0; i < sc->n_meshes; i++) {
        mesh_t *m = &cx->meshes[i];
        m->n_verts = (uint32_t)(sc->mesh_vert_off[i + 1] - sc->mesh_vert_off[i]);
        m->n_tris = (uint32_t)(sc->mesh_tri_off[i + 1] - sc->mesh_tri_off[i]);
        m->pos = sc->mesh_positions + 3 * sc->mesh_vert_off[i];
        m->nrm = sc->mesh_normals ? sc->mesh_normals + 3 * sc->mesh_vert_off[i] : NULL;
        m->idx = sc->mesh_indices + 3 * sc->mesh_tri_off[i];
        m->has_normals = sc->mesh_has_normals ? sc->mesh_has_normals[i] : 0;
        m->has_uv = sc->mesh_has_texcoords && sc->mesh_texcoords && sc->mesh_has_texcoords[i];
        m->uv = m->has_uv ? sc->mesh_texcoords + 2 * sc->mesh_vert_off[i] : NULL;
        if (m->n_verts == 0) return -3; /* mesh.rs:71 assert */
        po_vec3 mn = ld3(m->pos), mx = mn; /* mesh.rs:72-75 */
        for (uint32_t v = 1; v < m->n_verts; v++) { po_vec3 p = ld3(m->pos + 3 * v); mn = po_vmin(mn, p); mx = po_vmax(mx, p); }
        m->bounds = bbox_new(mn, mx);
    }
    /* KDMesh trees are built by the scene script before render (kdmesh.rs:37-58) */
    for (uint32_t n = 0; n < sc->n_nodes; n++)
        if (sc->node_prim_type[n] == PO_KDMESH) {
            int smooth = sc->node_prim_flags[n] & 1;
            if (smooth && !cx->meshes[sc->node_prim_data[n]].has_normals) return -4; /* mesh.rs:135-138 */
            kdmesh_build(&cx->meshes[sc->node_prim_data[n]], smooth, kd_mesh_depth);
        } else if (sc->node_prim_type[n] == PO_MESH) {
            if ((sc->node_prim_flags[n] & 1) && !cx->meshes[sc->node_prim_data[n]].has_normals) return -4;
        }
    cx->h_inv = (po_mat4 *)malloc(sizeof(po_mat4) * sc->n_nodes);
    cx->h_nrm = (po_mat4 *)malloc(sizeof(po_mat4) * sc->n_nodes);
    for (uint32_t n = 0; n < sc->n_nodes; n++) { /* scene.rs:201-205 set_transform */
        po_mat4 t = ldm(sc->node_trans + 16 * (size_t)n);
        cx->h_inv[n] = po_inverted(&t);
        cx->h_nrm[n] = po_transposed(&cx->h_inv[n]);
    }
    int rc = ctx_flatten(cx);
    if (rc) return rc;
    if (mode == PO_MODE_KD) { /* kdscene.rs:19-43 */
        uint32_t n = cx->n_flat;
        po_vec3 *mins = (po_vec3 *)malloc(sizeof(po_vec3) * (n ? n : 1)), *maxs = (po_vec3 *)malloc(sizeof(po_vec3) * (n ? n : 1));
        uint32_t *items = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        for (uint32_t i = 0; i < n; i++) { mins[i] = cx->flat[i].bmin; maxs[i] = cx->flat[i].bmax; items[i] = i; }
        items_bounds(mins, maxs, items, n, &cx->kd_bmin, &cx->kd_bmax);
        kdconf_t conf = {3, 3, 10};
        cx->kd_root = kd_partitioned(mins, maxs, items, n, cx->kd_bmin, cx->kd_bmax, po_v3(1, 0, 0), kd_depth, conf);
        po_vec3 d = po_sub(cx->kd_bmax, cx->kd_bmin);
        cx->kd_extent = po_dot(d, d); /* node.rs:62-64 + bounding_box.rs:95-99 */
        free(mins); free(maxs);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * material.rs + light.rs + ray.rs:139-148
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint64_t seed, pixel; uint32_t sample, draw; int jitter; } rng_t;
static inline double rng_next(rng_t *r) { return po_rng_f64(r->seed, r->pixel, r->sample, r->draw++); }

static po_vec3 ray_color(const ctx_t *cx, const ray_t *ray, po_vec3 bg, uint32_t depth, rng_t *rng, po_stats *st);

static int refracted_direction(po_vec3 ray_dir, po_vec3 normal, double eta, po_vec3 *out) { /* material.rs:27-48 */
    double eta_outside = 1.00; /* material.rs:15 AIR_REFRACTION_INDEX */
    double ray_dot_norm = po_dot(ray_dir, normal);
    double under_sqrt = 1.0 - eta_outside * eta_outside * (1.0 - ray_dot_norm * ray_dot_norm) / (eta * eta);
    if (under_sqrt < 0.0) return 0;
    po_vec3 d1 = po_divs(po_scale(po_sub(ray_dir, po_scale(normal, ray_dot_norm)), eta_outside), eta);
    po_vec3 d2 = po_scale(normal, sqrt(under_sqrt));
    *out = po_sub(d1, d2);
    return 1;
}

static inline double powi5(double x) { /* f64::powi(5) = llvm.powi: x * ((x*x)*(x*x)) */
    double x2 = x * x, x4 = x2 * x2;
    return x * x4;
}

/* texture.rs:96-141 RgbImageBuffer::at: nearest texel, wrap-around; channels as f64 / 255 */
static int near_integer(double f) { /* within PO_TEX_EDGE_ULPS units in the last place of f of an integer (the truncation below changes there) */
    if (f != f || fabs(f) >= 4503599627370496.0) return 0;
    double r = nearbyint(f), ulp = nextafter(fabs(f) > 1.0 ? fabs(f) : 1.0, INFINITY) - (fabs(f) > 1.0 ? fabs(f) : 1.0);
    return fabs(f - r) <= PO_TEX_EDGE_ULPS * ulp;
}
static po_vec3 texel_at(const po_scene *sc, int32_t tex, double u, double v, int from_sphere, po_stats *st) {
    int64_t width = sc->texture_size[2 * tex], height = sc->texture_size[2 * tex + 1];
    double fx = u * (double)(width - 1), fy = v * (double)(height - 1);
    if (from_sphere) { st->tex_sphere_lookups++; if (near_integer(fx) || near_integer(fy)) st->tex_sphere_near_edge++; }
    /* Rust `as i64`: truncation toward zero, saturating, NaN -> 0 */
    int64_t x = fx != fx ? 0 : (fx >= 9.2233720368547758e18 ? INT64_MAX : (fx <= -9.2233720368547758e18 ? INT64_MIN : (int64_t)fx));
    int64_t y = fy != fy ? 0 : (fy >= 9.2233720368547758e18 ? INT64_MAX : (fy <= -9.2233720368547758e18 ? INT64_MIN : (int64_t)fy));
    x %= width; if (x < 0) x += width; /* rem_euclid, texture.rs:99-112 */
    y %= height; if (y < 0) y += height;
    const uint8_t *px = sc->texture_rgb + sc->texture_offset[tex] + 3 * ((size_t)y * (size_t)width + (size_t)x);
    return po_v3((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
}

static po_vec3 hit_color(const ctx_t *cx, int32_t mat_id, po_vec3 bg, po_vec3 ray_dir, po_vec3 hit_point, po_vec3 normal_raw,
                         const hit_t *hit, uint32_t depth, rng_t *rng, po_stats *st) { /* material.rs:91-320 */
    if (depth > 10) return bg; /* material.rs:12, :102-104 */
    st->hits++;
    const po_scene *sc = cx->sc;
    const double *m = sc->materials + 10 * (size_t)mat_id;
    po_vec3 kd = po_v3(m[0], m[1], m[2]), ks = po_v3(m[3], m[4], m[5]);
    double shininess = m[6], reflectivity = m[7], glossy = m[8], ior = m[9];

    po_vec3 view = po_neg(ray_dir);
    int32_t tex = sc->material_texture ? sc->material_texture[mat_id] : -1;
    int32_t nmap = sc->material_normal_map ? sc->material_normal_map[mat_id] : -1;
    double tu = hit->u, tv = hit->v;
    if (hit->has_uv && sc->material_uv_trans) { /* material.rs:113-117: uv_trans * (u, v, 1), w dropped */
        const double *t9 = sc->material_uv_trans + 9 * (size_t)mat_id;
        double nu = (t9[0] * hit->u + t9[1] * hit->v) + t9[2] * 1.0;
        double nv = (t9[3] * hit->u + t9[4] * hit->v) + t9[5] * 1.0;
        tu = nu; tv = nv;
    }
    po_vec3 normal;
    if (nmap < 0) {
        normal = po_normalized(normal_raw); /* material.rs:123-125 */
    } else { /* material.rs:126-136 + texture.rs:192-221; quirk Q12: the TBN is the primitive's model-space one */
        if (!hit->has_uv || !hit->has_tbn) { st->kd_plane_miss += 1u << 20; return bg; } /* the reference panics */
        po_vec3 tn = texel_at(sc, nmap, tu, tv, hit->has_uv == 2, st);
        po_vec3 norm = po_v3(2.0 * tn.x - 1.0, 2.0 * tn.y - 1.0, -(2.0 * tn.z - 1.0));
        po_mat3 to_rh = {{{1, 0, 0}, {0, 0, -1}, {0, -1, 0}}};
        po_vec3 tex_norm = mat3_mul(&to_rh, norm);
        normal = mat3_mul(&hit->tbn, po_normalized(tex_norm));
    }
    if (tex >= 0) { /* material.rs:138-144, texture.rs:162-168: sRGB -> linear with powf(2.2) */
        if (!hit->has_uv) { st->kd_plane_miss += 1u << 20; return bg; } /* the reference panics */
        po_vec3 c = texel_at(sc, tex, tu, tv, hit->has_uv == 2, st);
        kd = po_v3(pow(c.x, PO_GAMMA), pow(c.y, PO_GAMMA), pow(c.z, PO_GAMMA));
    }
    po_vec3 color = po_mul(po_v3(sc->ambient[0], sc->ambient[1], sc->ambient[2]), kd); /* :148 */
    for (uint32_t li = 0; li < sc->n_lights; li++) { /* :149-211 */
        const double *L = sc->lights + 15 * (size_t)li;
        po_vec3 lpos = po_v3(L[0], L[1], L[2]), lcol = po_v3(L[3], L[4], L[5]);
        po_vec3 aa = po_v3(L[9], L[10], L[11]), ab = po_v3(L[12], L[13], L[14]);
        int empty = (aa.x == 0.0 && aa.y == 0.0 && aa.z == 0.0) || (ab.x == 0.0 && ab.y == 0.0 && ab.z == 0.0); /* light.rs:51-53 */
        if (!empty) { /* light.rs:62-70, :87-90 */
            double a_coord = 2.0 * rng_next(rng) - 1.0;
            double b_coord = 2.0 * rng_next(rng) - 1.0;
            lpos = po_add(lpos, po_add(po_scale(aa, a_coord), po_scale(ab, b_coord)));
        }
        po_vec3 hit_to_light = po_sub(lpos, hit_point);
        double light_dist = po_magnitude(hit_to_light);
        po_vec3 light_dir = po_divs(hit_to_light, light_dist);
        double attenuation = L[6] + L[7] * light_dist + L[8] * light_dist * light_dist; /* light.rs:31-33 */
        ray_t shadow = {hit_point, light_dir};
        po_range sr = {PO_EPSILON, INFINITY};
        hit_t sh; int32_t sm, sid;
        st->shadow++;
        if (!scene_cast(cx, &shadow, &sr, &sh, &sm, &sid, st)) { /* :179 unbounded: quirk Q2 */
            double normal_light = fmax(po_dot(normal, light_dir), 0.0);
            po_vec3 diffuse = po_scale(po_mul(kd, lcol), normal_light);
            po_vec3 specular = po_v3(0, 0, 0);
            if (ks.x > PO_EPSILON || ks.y > PO_EPSILON || ks.z > PO_EPSILON) { /* :188 */
                po_vec3 half = po_normalized(po_add(view, light_dir));
                double nhs = pow(fmax(po_dot(normal, half), 0.0), 4.0 * shininess);
                specular = po_scale(po_mul(ks, lcol), nhs);
            }
            color = po_add(color, po_divs(po_add(diffuse, specular), attenuation)); /* :209 */
        }
    }
    if (reflectivity > 0.0) { /* :216 */
        po_vec3 reflect_dir = po_sub(ray_dir, po_scale(po_scale(normal, 2.0), po_dot(ray_dir, normal))); /* :218 */
        if (glossy > 0.0) { /* :221-239 */
            po_vec3 off;
            if (fabs(reflect_dir.x) < PO_EPSILON && fabs(reflect_dir.y) < PO_EPSILON) off = po_add(reflect_dir, po_v3(0.0, 0.1, 0.0));
            else off = po_add(reflect_dir, po_v3(0.0, 0.0, 0.1));
            po_vec3 u_basis = po_cross(reflect_dir, off);
            po_vec3 v_basis = po_cross(reflect_dir, u_basis);
            double u_coord = -glossy / 2.0 + rng_next(rng) * glossy;
            double v_coord = -glossy / 2.0 + rng_next(rng) * glossy;
            reflect_dir = po_add(reflect_dir, po_add(po_scale(u_basis, u_coord), po_scale(v_basis, v_coord)));
        }
        ray_t rr = {hit_point, reflect_dir};
        if (depth + 1 > 10) st->depth11++; else st->reflect++;
        po_vec3 reflected = ray_color(cx, &rr, bg, depth + 1, rng, st); /* :242-243 */
        if (ior > 0.0) { /* :247-310 dielectric */
            po_vec3 refract_dir; double cos_incident; int have = 0;
            if (po_dot(ray_dir, normal) < 0.0) {
                if (!refracted_direction(ray_dir, normal, ior, &refract_dir)) {
                    /* :257-258 expect(): the reference panics; treat as total internal reflection */
                    color = po_add(color, po_scale(reflected, reflectivity));
                } else { cos_incident = po_dot(po_neg(ray_dir), normal); have = 1; }
            } else if (refracted_direction(ray_dir, po_neg(normal), 1.0 / ior, &refract_dir)) {
                cos_incident = po_dot(refract_dir, normal); have = 1;
            } else {
                color = po_add(color, po_scale(reflected, reflectivity)); /* :280 */
            }
            if (have) {
                double r0 = (ior - 1.0) * (ior - 1.0);
                r0 = r0 / ((ior + 1.0) * (ior + 1.0));
                double schlick = r0 + (1.0 - r0) * powi5(1.0 - cos_incident);
                double transmittance = 1.0 - schlick;
                ray_t tr = {hit_point, refract_dir};
                if (depth + 1 > 10) st->depth11++; else st->refract++;
                po_vec3 refracted = ray_color(cx, &tr, bg, depth + 1, rng, st);
                po_vec3 total = po_add(po_scale(reflected, schlick), po_scale(refracted, transmittance));
                color = po_add(color, po_scale(total, reflectivity)); /* :309 */
            }
        } else {
            color = po_add(color, po_scale(reflected, reflectivity)); /* :315 */
        }
    }
    return color;
}

static po_vec3 ray_color(const ctx_t *cx, const ray_t *ray, po_vec3 bg, uint32_t depth, rng_t *rng, po_stats *st) { /* ray.rs:139-148 */
    po_range range = {PO_EPSILON, INFINITY};
    hit_t h; int32_t mat, id;
    if (scene_cast(cx, ray, &range, &h, &mat, &id, st))
        return hit_color(cx, mat, bg, ray->d, h.p, h.n, &h, depth, rng, st);
    return bg;
}

/* ---------------------------------------------------------------------------------------------
 * camera.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct { po_vec3 eye; po_mat4 view_to_world; double fov_factor, aspect, width, height; } camera_t;

static camera_t camera_new(const po_camera_settings *cs, double width, double height) { /* camera.rs:34-45 */
    camera_t c;
    po_vec3 eye = ld3(cs->eye), center = ld3(cs->center), up = ld3(cs->up);
    /* vek Mat4::look_at_rh (SURVEY App.A.2) */
    po_vec3 f = po_normalized(po_sub(center, eye));
    po_vec3 s = po_normalized(po_cross(f, up));
    po_vec3 u = po_cross(s, f);
    po_mat4 v = po_identity();
    v.m[0][0] = s.x; v.m[0][1] = s.y; v.m[0][2] = s.z; v.m[0][3] = -po_dot(s, eye);
    v.m[1][0] = u.x; v.m[1][1] = u.y; v.m[1][2] = u.z; v.m[1][3] = -po_dot(u, eye);
    v.m[2][0] = -f.x; v.m[2][1] = -f.y; v.m[2][2] = -f.z; v.m[2][3] = po_dot(f, eye);
    c.eye = eye;
    c.view_to_world = po_inverted(&v);
    c.fov_factor = tan(cs->fovy_radians / 2.0);
    c.aspect = width / height;
    c.width = width; c.height = height;
    return c;
}
static ray_t camera_ray_at(const camera_t *c, double x, double y) { /* camera.rs:48-84 */
    double ndc_y = y / c->height;
    double view_y = (1.0 - 2.0 * ndc_y) * c->fov_factor;
    double ndc_x = x / c->width;
    double view_x = (2.0 * ndc_x - 1.0) * c->aspect * c->fov_factor;
    po_vec3 pixel_world = po_xform_point(&c->view_to_world, po_v3(view_x, view_y, -1.0));
    ray_t r;
    r.o = c->eye;
    r.d = po_normalized(po_sub(pixel_world, c->eye));
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * render.rs
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const ctx_t *cx; const camera_t *cam; const double *bg; const po_render_params *p;
    uint8_t *rgb; double *linear;
    uint32_t next_row; pthread_mutex_t lock;
    po_stats total;
} job_t;

static inline uint8_t to_u8(double c) { /* render.rs:143-147 `as u8`: saturating, NaN -> 0 */
    double v = c * 255.0;
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

static void stats_add(po_stats *a, const po_stats *b) {
    uint64_t *pa = (uint64_t *)a; const uint64_t *pb = (const uint64_t *)b;
    for (size_t i = 0; i < sizeof(po_stats) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

static void render_pixel(const job_t *j, uint32_t x, uint32_t y, po_stats *st) { /* render.rs:22-51 + :131-147 */
    const po_render_params *p = j->p;
    size_t pix = (size_t)y * p->width + x;
    const double *b = p->background_rows ? j->bg + 3 * (size_t)y : j->bg + 3 * pix; /* render.rs:31-34: one colour per integer pixel */
    po_vec3 bg = ld3(b);
    /* render.rs:36-43 reduces with rayon (no fixed association). The build's summation contract:
     * chunks of PO_SAMPLE_CHUNK samples, each summed in ascending sample order, then the chunk sums
     * added in ascending chunk order. */
    po_vec3 total = po_v3(0, 0, 0), chunk = po_v3(0, 0, 0);
    for (uint32_t s = 0; s < p->samples; s++) {
        rng_t rng = {p->seed, pix, s, 0, p->jitter_mode};
        double jx = 0.5, jy = 0.5;
        if (p->jitter_mode == PO_JITTER_RNG) { jx = rng_next(&rng); jy = rng_next(&rng); }
        else rng.draw = 2;
        ray_t ray = camera_ray_at(j->cam, (double)x + jx, (double)y + jy);
        st->primary++;
        po_vec3 c = ray_color(j->cx, &ray, bg, 0, &rng, st);
        chunk = s % PO_SAMPLE_CHUNK == 0 ? c : po_add(chunk, c);
        if ((s + 1) % PO_SAMPLE_CHUNK == 0 || s + 1 == p->samples) total = s < PO_SAMPLE_CHUNK ? chunk : po_add(total, chunk);
    }
    po_vec3 color = po_divs(total, (double)p->samples); /* render.rs:45 */
    if (j->linear) { double *o = j->linear + 3 * pix; o[0] = color.x; o[1] = color.y; o[2] = color.z; }
    double g = 1.0 / PO_GAMMA;
    double ch[3] = {pow(color.x, g), pow(color.y, g), pow(color.z, g)}; /* render.rs:47 */
    uint8_t *o = j->rgb + 3 * pix;
    for (int k = 0; k < 3; k++) {
        double v = ch[k];
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); /* render.rs:50 clamp01 */
        o[k] = to_u8(v);
    }
}

static void *render_worker(void *arg) {
    job_t *j = (job_t *)arg;
    po_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        pthread_mutex_lock(&j->lock);
        uint32_t y = j->next_row++;
        pthread_mutex_unlock(&j->lock);
        if (y > j->p->y1) break;
        for (uint32_t x = j->p->x0; x <= j->p->x1; x++) render_pixel(j, x, y, &st);
    }
    pthread_mutex_lock(&j->lock);
    stats_add(&j->total, &st);
    pthread_mutex_unlock(&j->lock);
    return NULL;
}

/* Wall time of the last po_render call, split where render.rs splits it: scene conversion (render.rs:115-126: flatten,
 * k-d tree builds) and the pixel loop (render.rs:127-150). For bench.py's CPU baseline, which times the pixel loop. */
static double g_last_prepare_ms = 0.0, g_last_pixels_ms = 0.0;
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
void po_last_render_ms(double out[2]) { out[0] = g_last_prepare_ms; out[1] = g_last_pixels_ms; }

int po_render(const po_scene *scene, const po_camera_settings *cam, const double *background,
              const po_render_params *p, uint8_t *rgb, double *linear, po_stats *stats) {
    if (!scene || !cam || !background || !p || !rgb) return -1;
    if (p->x0 >= p->width || p->y0 >= p->height || p->x1 >= p->width || p->y1 >= p->height) return -5; /* render.rs:79-90 panics */
    if (p->samples == 0) return -6;
    ctx_t cx;
    double t_begin = now_ms();
    int rc = ctx_init(&cx, scene, p->mode, p->kd_depth, p->kd_mesh_depth);
    if (rc) { ctx_free(&cx); return rc; }
    double t_loop = now_ms();
    g_last_prepare_ms = t_loop - t_begin;
    camera_t c = camera_new(cam, (double)p->width, (double)p->height);
    job_t j; memset(&j, 0, sizeof j);
    j.cx = &cx; j.cam = &c; j.bg = background; j.p = p; j.rgb = rgb; j.linear = linear; j.next_row = p->y0;
    pthread_mutex_init(&j.lock, NULL);
    int nt = p->n_threads > 0 ? p->n_threads : (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nt < 1) nt = 1;
    if (nt > 1024) nt = 1024;
    if (p->y1 >= p->y0 && p->x1 >= p->x0) { /* render.rs:60-65: an inverted slice renders nothing */
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
        int started = 0;
        for (int i = 0; i < nt - 1; i++) if (pthread_create(&th[started], NULL, render_worker, &j) == 0) started++;
        render_worker(&j);
        for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.lock);
    g_last_pixels_ms = now_ms() - t_loop;
    if (stats) *stats = j.total;
    ctx_free(&cx);
    return 0;
}

int po_cast_rays(const po_scene *scene, int mode, int kd_depth, int kd_mesh_depth, uint64_t n,
                 const double *origins, const double *directions, double *out_t, int32_t *out_id,
                 double *out_point, double *out_normal) {
    ctx_t cx;
    int rc = ctx_init(&cx, scene, mode, kd_depth, kd_mesh_depth);
    if (rc) { ctx_free(&cx); return rc; }
    po_stats st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        ray_t r = {ld3(origins + 3 * i), ld3(directions + 3 * i)};
        po_range range = {PO_EPSILON, INFINITY};
        hit_t h; int32_t mat, id;
        if (scene_cast(&cx, &r, &range, &h, &mat, &id, &st)) {
            out_t[i] = h.t; if (out_id) out_id[i] = id;
            if (out_point) { out_point[3 * i] = h.p.x; out_point[3 * i + 1] = h.p.y; out_point[3 * i + 2] = h.p.z; }
            if (out_normal) { out_normal[3 * i] = h.n.x; out_normal[3 * i + 1] = h.n.y; out_normal[3 * i + 2] = h.n.z; }
        } else {
            out_t[i] = INFINITY; if (out_id) out_id[i] = -1;
            if (out_point) { out_point[3 * i] = out_point[3 * i + 1] = out_point[3 * i + 2] = 0.0; }
            if (out_normal) { out_normal[3 * i] = out_normal[3 * i + 1] = out_normal[3 * i + 2] = 0.0; }
        }
    }
    ctx_free(&cx);
    return 0;
}

int po_color_rays(const po_scene *scene, int mode, int kd_depth, int kd_mesh_depth, uint64_t n,
                  const double *origins, const double *directions, const double background[3], double *out_rgb) {
    ctx_t cx;
    int rc = ctx_init(&cx, scene, mode, kd_depth, kd_mesh_depth);
    if (rc) { ctx_free(&cx); return rc; }
    po_stats st; memset(&st, 0, sizeof st);
    po_vec3 bg = ld3(background);
    for (uint64_t i = 0; i < n; i++) {
        ray_t r = {ld3(origins + 3 * i), ld3(directions + 3 * i)};
        rng_t rng = {0, i, 0, 2, PO_JITTER_CENTRE};
        po_vec3 c = ray_color(&cx, &r, bg, 0, &rng, &st);
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    }
    ctx_free(&cx);
    return 0;
}

int po_camera_rays(const po_camera_settings *cam, double width, double height, uint64_t n,
                   const double *xy, double *origins, double *directions) {
    camera_t c = camera_new(cam, width, height);
    for (uint64_t i = 0; i < n; i++) {
        ray_t r = camera_ray_at(&c, xy[2 * i], xy[2 * i + 1]);
        origins[3 * i] = r.o.x; origins[3 * i + 1] = r.o.y; origins[3 * i + 2] = r.o.z;
        directions[3 * i] = r.d.x; directions[3 * i + 1] = r.d.y; directions[3 * i + 2] = r.d.z;
    }
    return 0;
}

int po_flatten(const po_scene *scene, uint32_t cap, double *trans, double *invtrans, double *normal_trans,
               int32_t *prim_type, int32_t *prim_data, int32_t *prim_flags, int32_t *material, double *bounds) {
    ctx_t cx;
    int rc = ctx_init(&cx, scene, PO_MODE_FLAT, -1, -1);
    if (rc) { ctx_free(&cx); return rc; }
    uint32_t n = cx.n_flat;
    for (uint32_t i = 0; i < n && i < cap; i++) {
        const flatnode_t *f = &cx.flat[i];
        if (trans) memcpy(trans + 16 * i, f->trans.m, 128);
        if (invtrans) memcpy(invtrans + 16 * i, f->inv.m, 128);
        if (normal_trans) memcpy(normal_trans + 16 * i, f->nrm.m, 128);
        if (prim_type) prim_type[i] = f->type;
        if (prim_data) prim_data[i] = f->data;
        if (prim_flags) prim_flags[i] = f->flags;
        if (material) material[i] = f->material;
        if (bounds) { double *b = bounds + 6 * i; b[0] = f->bmin.x; b[1] = f->bmin.y; b[2] = f->bmin.z; b[3] = f->bmax.x; b[4] = f->bmax.y; b[5] = f->bmax.z; }
    }
    ctx_free(&cx);
    return (int)n;
}

typedef struct {
    uint32_t node_cap, item_cap, n_nodes, n_items;
    int32_t *kind, *axis, *front, *back, *first, *count, *leaf_items; double *plane;
    int overflow;
} dump_t;

static int32_t kd_dump(const kdnode_t *n, dump_t *d) {
    if (d->n_nodes >= d->node_cap) { d->overflow = 1; return -1; }
    int32_t me = (int32_t)d->n_nodes++;
    if (n->is_leaf) {
        d->kind[me] = 1; d->axis[me] = -1; d->plane[me] = 0.0; d->front[me] = d->back[me] = -1;
        d->first[me] = (int32_t)d->n_items; d->count[me] = (int32_t)n->n_items;
        for (uint32_t i = 0; i < n->n_items; i++) {
            if (d->n_items >= d->item_cap) { d->overflow = 1; return me; }
            d->leaf_items[d->n_items++] = (int32_t)n->items[i];
        }
    } else {
        d->kind[me] = 0; d->axis[me] = n->axis;
        d->plane[me] = n->axis == 0 ? n->sep.point.x : (n->axis == 1 ? n->sep.point.y : n->sep.point.z);
        d->first[me] = d->count[me] = 0;
        int32_t f = kd_dump(n->front, d);
        int32_t b = kd_dump(n->back, d);
        d->front[me] = f; d->back[me] = b;
    }
    return me;
}

int po_kd_scene_dump(const po_scene *scene, int kd_depth, uint32_t node_cap, uint32_t item_cap,
                     int32_t *kind, int32_t *axis, double *plane, int32_t *front, int32_t *back,
                     int32_t *first, int32_t *count, int32_t *leaf_items, uint32_t *n_items, double root_bounds[6]) {
    ctx_t cx;
    int rc = ctx_init(&cx, scene, PO_MODE_KD, kd_depth, -1);
    if (rc) { ctx_free(&cx); return rc; }
    dump_t d = {node_cap, item_cap, 0, 0, kind, axis, front, back, first, count, leaf_items, plane, 0};
    kd_dump(cx.kd_root, &d);
    if (n_items) *n_items = d.n_items;
    if (root_bounds) {
        root_bounds[0] = cx.kd_bmin.x; root_bounds[1] = cx.kd_bmin.y; root_bounds[2] = cx.kd_bmin.z;
        root_bounds[3] = cx.kd_bmax.x; root_bounds[4] = cx.kd_bmax.y; root_bounds[5] = cx.kd_bmax.z;
    }
    ctx_free(&cx);
    return d.overflow ? -7 : (int)d.n_nodes;
}

int po_kd_partition_boxes(uint32_t n, const double *mins, const double *maxs, int max_depth,
                          int target_max_nodes, int target_max_merit, int max_tries,
                          uint32_t node_cap, uint32_t item_cap,
                          int32_t *kind, int32_t *axis, double *plane, int32_t *front, int32_t *back,
                          int32_t *first, int32_t *count, int32_t *leaf_items, uint32_t *n_items) {
    po_vec3 *mn = (po_vec3 *)malloc(sizeof(po_vec3) * (n ? n : 1)), *mx = (po_vec3 *)malloc(sizeof(po_vec3) * (n ? n : 1));
    uint32_t *items = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) { mn[i] = ld3(mins + 3 * i); mx[i] = ld3(maxs + 3 * i); items[i] = i; }
    po_vec3 bmin, bmax;
    items_bounds(mn, mx, items, n, &bmin, &bmax);
    kdconf_t conf = {target_max_nodes, target_max_merit, max_tries};
    kdnode_t *root = kd_partitioned(mn, mx, items, n, bmin, bmax, po_v3(1, 0, 0), max_depth, conf);
    dump_t d = {node_cap, item_cap, 0, 0, kind, axis, front, back, first, count, leaf_items, plane, 0};
    kd_dump(root, &d);
    if (n_items) *n_items = d.n_items;
    kd_free(root); free(mn); free(mx);
    return d.overflow ? -7 : (int)d.n_nodes;
}

/* Casts rays through a HAND-BUILT scene k-d tree given in the dump format (node 0 = root), for the
 * reference's node.rs:219-351 edge-case tests, which construct KDTreeNode::Split by hand. */
static kdnode_t *kd_from_dump(int32_t i, const int32_t *kind, const int32_t *axis, const double *plane, const int32_t *front,
                              const int32_t *back, const int32_t *first, const int32_t *count, const int32_t *leaf_items) {
    kdnode_t *n = (kdnode_t *)calloc(1, sizeof *n);
    if (kind[i] == 1) {
        n->is_leaf = 1; n->n_items = (uint32_t)count[i];
        n->items = (uint32_t *)malloc(sizeof(uint32_t) * (n->n_items ? n->n_items : 1));
        for (uint32_t k = 0; k < n->n_items; k++) n->items[k] = (uint32_t)leaf_items[first[i] + k];
    } else {
        n->axis = axis[i];
        n->sep.normal = po_v3(axis[i] == 0, axis[i] == 1, axis[i] == 2);
        n->sep.point = po_v3(axis[i] == 0 ? plane[i] : 0.0, axis[i] == 1 ? plane[i] : 0.0, axis[i] == 2 ? plane[i] : 0.0);
        n->front = kd_from_dump(front[i], kind, axis, plane, front, back, first, count, leaf_items);
        n->back = kd_from_dump(back[i], kind, axis, plane, front, back, first, count, leaf_items);
    }
    return n;
}

int po_kd_cast_custom(const po_scene *scene, const int32_t *kind, const int32_t *axis, const double *plane,
                      const int32_t *front, const int32_t *back, const int32_t *first, const int32_t *count,
                      const int32_t *leaf_items, const double root_bounds[6], uint64_t n,
                      const double *origins, const double *directions, double *out_t, int32_t *out_id) {
    ctx_t cx;
    int rc = ctx_init(&cx, scene, PO_MODE_FLAT, -1, -1);
    if (rc) { ctx_free(&cx); return rc; }
    cx.mode = PO_MODE_KD;
    cx.kd_root = kd_from_dump(0, kind, axis, plane, front, back, first, count, leaf_items);
    po_vec3 d = po_sub(ld3(root_bounds + 3), ld3(root_bounds));
    cx.kd_extent = po_dot(d, d);
    po_stats st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        ray_t r = {ld3(origins + 3 * i), ld3(directions + 3 * i)};
        po_range range = {PO_EPSILON, INFINITY};
        hit_t h; int32_t mat, id;
        if (scene_cast(&cx, &r, &range, &h, &mat, &id, &st)) { out_t[i] = h.t; out_id[i] = id; }
        else { out_t[i] = INFINITY; out_id[i] = -1; }
    }
    ctx_free(&cx);
    return 0;
}

int po_quadratic_solve(double a, double b, double c, double out[2]) { return po_quadratic(a, b, c, out); }

void po_transform_bounds(const double trans[16], const double min_in[3], const double max_in[3], double min_out[3], double max_out[3]) {
    po_mat4 m = ldm(trans);
    po_vec3 mn, mx;
    bbox_transform_minmax(&m, ld3(min_in), ld3(max_in), &mn, &mx);
    min_out[0] = mn.x; min_out[1] = mn.y; min_out[2] = mn.z;
    max_out[0] = mx.x; max_out[1] = mx.y; max_out[2] = mx.z;
}

/* ops: one char per builder call in call order: 's' scaled(x,y,z) 't' translated(x,y,z)
 * 'x' 'y' 'z' rotated_*(radians) (scene.rs:163-199); args are consumed left to right. */
void po_mat4_compose(const char *ops, const double *args, double out[16]) {
    po_mat4 m = po_identity();
    for (const char *o = ops; *o; o++) {
        po_mat4 b;
        switch (*o) {
        case 's': b = po_scaling(po_v3(args[0], args[1], args[2])); args += 3; break;
        case 't': b = po_translation(po_v3(args[0], args[1], args[2])); args += 3; break;
        case 'x': b = po_rotation_x(args[0]); args += 1; break;
        case 'y': b = po_rotation_y(args[0]); args += 1; break;
        case 'z': b = po_rotation_z(args[0]); args += 1; break;
        default: continue;
        }
        m = po_matmul(&b, &m);
    }
    memcpy(out, m.m, 128);
}

void po_mat4_inverse(const double in[16], double out[16]) {
    po_mat4 m = ldm(in), r = po_inverted(&m);
    memcpy(out, r.m, 128);
}

double po_rng_draw(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t draw) { return po_rng_f64(seed, pixel, sample, draw); }
