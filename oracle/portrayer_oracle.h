/* ORACLE — TEST INFRASTRUCTURE ONLY (see po_math.h for the rules and the parity status).
 *
 * CPU restatement (plain C, f64, scalar, no FMA) of portrayer's per-pixel ray-cast/shade path:
 *   src/render.rs, src/camera.rs, src/ray.rs, src/scene.rs, src/flat_scene.rs,
 *   src/bounding_box.rs, src/kdtree/{leaf,node,kdscene,kdmesh}.rs, src/primitive/{sphere,...}.rs,
 *   src/material.rs, src/light.rs, src/math.rs
 * in the reference's three traversal modes (cargo features): hierarchical (default), flat_scene
 * and kdtree (render.rs:121-126).
 *
 * Input is the HIERARCHICAL scene as plain arrays (the scene DAG with each node's LOCAL
 * transform); flattening, inverses, bounding boxes and k-d trees are recomputed here so that they
 * check the product's host code too.
 */
#ifndef PORTRAYER_ORACLE_H
#define PORTRAYER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive.rs:67-81 (enum Primitive), minus the variants that need no tag here */
enum { PO_NONE = -1, PO_SPHERE = 0, PO_TRIANGLE = 1, PO_MESH = 2, PO_KDMESH = 3, PO_PLANE = 4, PO_CUBE = 5, PO_CYLINDER = 6, PO_CONE = 7 };
/* render.rs:121-126 */
enum { PO_MODE_HIER = 0, PO_MODE_FLAT = 1, PO_MODE_KD = 2 };
/* SURVEY App.B.4: centre = (x+0.5, y+0.5); rng = counter-based jitter */
enum { PO_JITTER_CENTRE = 0, PO_JITTER_RNG = 1 };

typedef struct {
    /* scene.rs:36-48 SceneNode, as a DAG (Arc sharing = instancing) */
    uint32_t n_nodes;
    const double *node_trans;        /* n_nodes x 16, row-major LOCAL transform (model->parent) */
    const int32_t *node_prim_type;   /* PO_NONE or PO_* */
    const int32_t *node_prim_data;   /* mesh index (MESH/KDMESH), triangle index (TRIANGLE) */
    const int32_t *node_prim_flags;  /* bit0: Shading::Smooth (mesh.rs:11-18) */
    const int32_t *node_material;    /* index into materials */
    const uint32_t *node_child_off;  /* n_nodes + 1 offsets into children[] */
    const uint32_t *children;        /* child node indices, in `children` Vec order */
    uint32_t root;
    /* mesh.rs:21-34 MeshData */
    uint32_t n_meshes;
    const uint64_t *mesh_vert_off;   /* n_meshes + 1, in vertices */
    const uint64_t *mesh_tri_off;    /* n_meshes + 1, in triangles */
    const double *mesh_positions;    /* total_verts x 3 */
    const double *mesh_normals;      /* total_verts x 3 (rows of meshes without normals unused) */
    const uint8_t *mesh_has_normals; /* n_meshes */
    const uint32_t *mesh_indices;    /* total_tris x 3, indices local to the mesh */
    /* triangle.rs:8-19 stand-alone triangles */
    uint32_t n_triangles;
    const double *tri_vertices;      /* n x 9 (a, b, c) */
    const double *tri_normals;       /* n x 9 or NULL */
    const uint8_t *tri_has_normals;  /* n or NULL */
    /* material.rs:50-86 hot fields: diffuse3, specular3, shininess, reflectivity, glossy_side_length, refraction_index */
    uint32_t n_materials;
    const double *materials;         /* x 10 */
    /* light.rs:74-91: position3, color3, falloff c0 c1 c2, area a3, area b3 */
    uint32_t n_lights;
    const double *lights;            /* x 15 */
    double ambient[3];
    /* texture.rs: image textures / normal maps (all optional: NULL / 0 when the scene has none) */
    const double *mesh_texcoords;    /* total_verts x 2 (mesh.rs:30) */
    const uint8_t *mesh_has_texcoords; /* n_meshes */
    const double *tri_texcoords;     /* n_triangles x 6 (triangle.rs:18) */
    const uint8_t *tri_has_texcoords; /* n_triangles */
    const int32_t *material_texture;    /* n_materials: index into textures, -1 = none (material.rs:75) */
    const int32_t *material_normal_map; /* n_materials: index into textures, -1 = none (material.rs:85) */
    const double *material_uv_trans;    /* n_materials x 9, row-major Mat3 (material.rs:83) */
    uint32_t n_textures;
    const uint32_t *texture_size;    /* n_textures x 2: width, height */
    const uint64_t *texture_offset;  /* n_textures: byte offset of the first texel in texture_rgb */
    const uint8_t *texture_rgb;      /* RGB8 texels, row-major (RgbImageBuffer, texture.rs:74-76) */
} po_scene;

typedef struct { double eye[3], center[3], up[3], fovy_radians; } po_camera_settings; /* camera.rs:5-14 */

typedef struct {
    uint64_t primary, shadow, reflect, refract; /* rays cast at depth <= 10 */
    uint64_t depth11;                           /* rays cast at depth 11 whose result is discarded (ray.rs:139-147 + material.rs:102-104) */
    uint64_t hits;                              /* shaded hits */
    uint64_t n_split, n_leaf;                   /* k-d nodes visited (scene tree + mesh trees) */
    uint64_t n_analytic;                        /* flat-node / scene-node candidate tests (transform + dispatch) */
    uint64_t n_tri;                             /* triangle tests */
    uint64_t n_bbox;                            /* mesh bounding-box tests */
    uint64_t kd_plane_miss;                     /* node.rs:146-147,177-178 would have panicked */
    /* Round 5. A sphere's texture coordinates go through libm's atan2 / acos (sphere.rs:59-96); the device's routines are within 2 / 1 ulp of
     * glibc's, not bit-equal. A texel can only differ when u (w - 1) or v (h - 1) lies that close to an integer: these two counters let a test PROVE that no
     * lookup of a render did - tex_sphere_lookups counts the texel fetches (texture.rs:96-141) whose coordinates came from a sphere hit, tex_sphere_near_edge
     * those of them with a scaled coordinate within PO_TEX_EDGE_ULPS units in the last place of an integer. */
    uint64_t tex_sphere_lookups, tex_sphere_near_edge;
} po_stats;
#define PO_TEX_EDGE_ULPS 4096.0

typedef struct {
    uint32_t width, height;
    uint32_t x0, y0, x1, y1;   /* inclusive slice, render.rs:115-138 */
    uint32_t samples;          /* env SAMPLES, render.rs:107-113 */
    uint64_t seed;
    int32_t jitter_mode;
    int32_t mode;              /* PO_MODE_* */
    int32_t kd_depth;          /* env KD_DEPTH, kdscene.rs:36-38 (default 10) */
    int32_t kd_mesh_depth;     /* env KD_MESH_DEPTH, kdmesh.rs:51-53 (default 10) */
    int32_t background_rows;   /* 1: background is H x 3 (one colour per row); 0: H x W x 3 */
    int32_t n_threads;         /* <= 0: all online cores */
} po_render_params;

/* Renders into rgb (H x W x 3, only the slice is written) and, when not NULL, linear
 * (H x W x 3 f64: the sample mean before gamma, render.rs:45). Returns 0, or <0 on error. */
int po_render(const po_scene *scene, const po_camera_settings *cam, const double *background,
              const po_render_params *params, uint8_t *rgb, double *linear, po_stats *stats);

/* out[0] = ms the last po_render spent converting the scene (render.rs:115-126), out[1] = ms in its pixel loop (render.rs:127-150) */
void po_last_render_ms(double out[2]);

/* Casts explicit world-space rays through the scene in the given mode (ray.rs:139-148 minus
 * shading): out_t = ray parameter or +inf, out_id = flat node index (BFS order) or -1. In HIER
 * mode out_id is -1/0 only. */
int po_cast_rays(const po_scene *scene, int mode, int kd_depth, int kd_mesh_depth, uint64_t n,
                 const double *origins, const double *directions, double *out_t, int32_t *out_id,
                 double *out_point, double *out_normal);

/* ray.color(scene, background, 0) for explicit rays (kdmesh.rs:141-151 mesh_equivalence). */
int po_color_rays(const po_scene *scene, int mode, int kd_depth, int kd_mesh_depth, uint64_t n,
                  const double *origins, const double *directions, const double background[3],
                  double *out_rgb);

/* camera.rs:34-84: primary rays for n (x, y) positions. */
int po_camera_rays(const po_camera_settings *cam, double width, double height, uint64_t n,
                   const double *xy, double *origins, double *directions);

/* Flattening result, for checking the product's host-side flatten (flat_scene.rs:18-46):
 * returns the number of flat nodes; fills up to cap entries of trans/invtrans/normal_trans
 * (16 doubles each, row-major), prim type/data/flags, material, world AABB (6 doubles). */
int po_flatten(const po_scene *scene, uint32_t cap, double *trans, double *invtrans, double *normal_trans,
               int32_t *prim_type, int32_t *prim_data, int32_t *prim_flags, int32_t *material, double *bounds);

/* k-d tree of the flattened scene (kdscene.rs:19-43), linearised in pre-order for inspection:
 * per node: kind (0 split, 1 leaf), axis, plane coordinate, front/back child index, leaf
 * first/count into leaf_items. Returns the number of nodes, or <0 when a cap is too small. */
int po_kd_scene_dump(const po_scene *scene, int kd_depth, uint32_t node_cap, uint32_t item_cap,
                     int32_t *kind, int32_t *axis, double *plane, int32_t *front, int32_t *back,
                     int32_t *first, int32_t *count, int32_t *leaf_items, uint32_t *n_items, double root_bounds[6]);

/* Restated helpers exposed for the reference's unit tests. */
int po_quadratic_solve(double a, double b, double c, double out[2]);              /* math.rs:159-179 */
void po_transform_bounds(const double trans[16], const double min_in[3], const double max_in[3],
                         double min_out[3], double max_out[3]);                      /* bounding_box.rs:123-148 */
void po_mat4_compose(const char *ops, const double *args, double out[16]);           /* scene.rs:163-205 builder ops */
void po_mat4_inverse(const double in[16], double out[16]);
/* leaf.rs:89-231 on a list of AABBs: same dump format as po_kd_scene_dump. */
int po_kd_partition_boxes(uint32_t n, const double *mins, const double *maxs, int max_depth,
                          int target_max_nodes, int target_max_merit, int max_tries,
                          uint32_t node_cap, uint32_t item_cap,
                          int32_t *kind, int32_t *axis, double *plane, int32_t *front, int32_t *back,
                          int32_t *first, int32_t *count, int32_t *leaf_items, uint32_t *n_items);
/* Casts rays through a hand-built scene k-d tree in the dump format (node 0 = root); for
 * node.rs:219-351. out_id = flat node index or -1. */
int po_kd_cast_custom(const po_scene *scene, const int32_t *kind, const int32_t *axis, const double *plane,
                      const int32_t *front, const int32_t *back, const int32_t *first, const int32_t *count,
                      const int32_t *leaf_items, const double root_bounds[6], uint64_t n,
                      const double *origins, const double *directions, double *out_t, int32_t *out_id);
double po_rng_draw(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t draw);

#ifdef __cplusplus
}
#endif
#endif
