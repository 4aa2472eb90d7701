/* portrayer_host.h — C entry points of the C++ host library (portrayer_amd/host/portrayer.hpp) for
 * callers that cannot include C++ (the Python harness in tests/ and bench.py).
 *
 * The host library is the counterpart of the reference crate's public API above the pixel loop:
 * SceneNode builder calls (src/scene.rs:151-205), flattening (src/flat_scene.rs:18-46), bounding
 * boxes (src/bounding_box.rs), the k-d tree build (src/kdtree/leaf.rs:89-231), the camera
 * (src/camera.rs:34-45) and Image::render (src/render.rs:93-126, :216-223). A scene is handed over
 * as a DESCRIPTION — per node the ordered list of builder calls, exactly what a scene script
 * writes — and replayed on the C++ SceneNode API, so this boundary adds no arithmetic of its own.
 * All functions return 0 or a negative code; ph_last_error() gives the message (thread-local).
 */
#ifndef PORTRAYER_HOST_H
#define PORTRAYER_HOST_H

#include <stdint.h>

#include "portrayer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ph_scene ph_scene;       /* a scene::HierScene                                  */
typedef struct ph_renderer ph_renderer; /* a flattened scene resident on one GPU               */

enum { PH_OK = 0, PH_ERR_ARGUMENT = -1, PH_ERR_PANIC = -2 /* where the reference would panic */, PH_ERR_RUNTIME = -3, PH_ERR_SMALL = -4 /* output capacity */ };

typedef struct {
    uint32_t n_nodes;
    /* builder calls of node i, in call order: ops[ops_off[i] .. ops_off[i+1]) with one char per call —
     * 's' scaled(x,y,z)  't' translated(x,y,z)  'x' 'y' 'z' rotated_x/y/z(radians) — and their
     * arguments consumed left to right from args[args_off[i] ..) (scene.rs:163-199). */
    const char *ops; const uint32_t *ops_off;
    const double *args; const uint32_t *args_off;
    const int32_t *prim_type;     /* -1: no geometry, else PT_PRIM_*                                     */
    const int32_t *prim_data;     /* MESH/KDMESH: mesh index; TRIANGLE: triangle index                   */
    const int32_t *prim_flags;    /* bit 0: Shading::Smooth                                              */
    const int32_t *material;
    const uint32_t *child_off;    /* n_nodes + 1                                                         */
    const uint32_t *children;     /* child node indices in `children` Vec order; shared nodes = Arc clones */
    uint32_t root;
    uint32_t n_meshes; const uint64_t *mesh_vert_off, *mesh_tri_off; const double *mesh_positions, *mesh_normals;
    const uint8_t *mesh_has_normals; const uint32_t *mesh_indices;
    uint32_t n_triangles; const double *tri_vertices, *tri_normals; const uint8_t *tri_has_normals;
    uint32_t n_materials; const double *materials; /* x 10, layout of pt_scene.materials                  */
    uint32_t n_lights; const double *lights;       /* x 15, layout of pt_scene.lights                     */
    double ambient[3];
    /* textures (src/texture.rs), same layout as the pt_scene fields of the same names; optional */
    const double *mesh_texcoords; const uint8_t *mesh_has_texcoords;
    const double *tri_texcoords; const uint8_t *tri_has_texcoords;
    const int32_t *material_texture, *material_normal_map; const double *material_uv_trans;
    uint32_t n_textures; const uint32_t *texture_size; const uint64_t *texture_offset; const uint8_t *texture_rgb;
} ph_scene_desc;

const char *ph_last_error(void);

int ph_scene_create(const ph_scene_desc *desc, ph_scene **out);
/* The C++ transliterations of the reference's scene scripts (the .cpp files under examples/): "single-triangle",
 * "primitives-simple", "macho-cows", "entering-the-mirror-dimension", "big-scene" (n = objects per
 * axis, ignored by the others). camera = eye3, center3, up3, fovy (radians); size = width, height. */
int ph_example_scene(const char *name, const char *assets_dir, int n, ph_scene **out, double camera[10], uint32_t size[2]);
void ph_scene_destroy(ph_scene *scene);

/* The scene DAG as arrays: unique nodes in depth-first pre-order from the root, materials and meshes
 * in order of first use. counts = nodes, children, meshes, vertices, mesh triangles, triangles,
 * materials, lights. Buffers may be NULL to skip a field. */
int ph_scene_counts(const ph_scene *scene, uint64_t counts[8]);
int ph_scene_export(const ph_scene *scene, double *node_trans /* x16 */, int32_t *prim_type, int32_t *prim_data, int32_t *prim_flags,
                    int32_t *material, uint32_t *child_off, uint32_t *children, uint32_t *root,
                    uint64_t *mesh_vert_off, uint64_t *mesh_tri_off, double *mesh_positions, double *mesh_normals, uint8_t *mesh_has_normals,
                    uint32_t *mesh_indices, double *tri_vertices, double *tri_normals, uint8_t *tri_has_normals,
                    double *materials, double *lights, double ambient[3]);
/* Textures, normal maps and texture coordinates in ph_scene_export's numbering (materials / meshes / triangles in order of
 * first use). counts = {n_textures, texel bytes}; call once with NULL arrays for the counts. */
int ph_scene_export_textures(const ph_scene *s, uint64_t counts[2], int32_t *material_texture, int32_t *material_normal_map,
                             double *material_uv_trans, uint32_t *texture_size, uint64_t *texture_offset, uint8_t *texture_rgb,
                             double *mesh_texcoords, uint8_t *mesh_has_texcoords, double *tri_texcoords, uint8_t *tri_has_texcoords);

/* FlatScene::from (flat_scene.rs:18-46): returns the number of flat nodes; fills up to cap entries. */
int ph_scene_flatten(const ph_scene *scene, uint32_t cap, double *trans, double *invtrans, double *normal_trans,
                     int32_t *prim_type, int32_t *material, double *bounds /* x6: min, max (flat_scene.rs:63-69) */);
/* The hierarchy as PT_TRAVERSE_HIER takes it (pt_scene's ABI-4 arrays; scene.rs:80-120): per flattened node its path through
 * the distinct SceneNodes (root first) and its depth-first rank; per SceneNode its OWN trans / invtrans / normal_trans.
 * Returns the number of flattened nodes; counts = {chain entries, graph nodes}. chain_off needs node_cap + 1 entries. */
int ph_scene_graph(const ph_scene *scene, uint32_t node_cap, uint32_t chain_cap, uint32_t graph_cap, uint32_t *chain_off, uint32_t *chain,
                   uint32_t *dfs_rank, double *graph_trans, double *graph_invtrans, double *graph_normal_trans, uint32_t counts[2]);
/* KDTreeScene::from (kdscene.rs:19-43), linearised like pt_kdtree. Returns the node count. */
int ph_scene_kdtree(const ph_scene *scene, int kd_depth, uint32_t node_cap, uint32_t item_cap, int32_t *axis, double *plane,
                    int32_t *front, int32_t *back, int32_t *first, int32_t *count, int32_t *leaf_items, uint32_t *n_items,
                    double root_bounds[6], int32_t *max_depth);
/* Camera::new (camera.rs:34-45) */
int ph_camera(const double camera[10], double width, double height, pt_camera *out);
/* MeshData::load_obj (mesh.rs:57-61): counts = vertices, triangles, has_normals; then copy out. */
int ph_obj_load(const char *path, uint64_t counts[3], double *positions, double *normals, uint32_t *indices, uint64_t vert_cap, uint64_t tri_cap);

/* Flatten (+ k-d build for PT_TRAVERSE_KD) and upload to GPU `device`: what render.rs:121-126 does. */
int ph_renderer_create(const ph_scene *scene, int traverse, int kd_depth, int device, ph_renderer **out);
void ph_renderer_destroy(ph_renderer *r);
pt_context *ph_renderer_context(ph_renderer *r);
/* 1, or the number of ranks when PORTRAYER_GPUS / PORTRAYER_DEVICES put the scene on a node (pt_node_*) */
int ph_renderer_ranks(ph_renderer *r);
/* the pt_node behind the renderer in that case (NULL on a single GPU): for callers that drive pt_node_render_resident themselves */
pt_node *ph_renderer_node(ph_renderer *r);
/* where the time before the first pixel went, in ms: flatten (flat_scene.rs:18-46), packing the ABI arrays, context / node
 * creation, the reference's k-d tree build (kdtree feature only), pt_scene_upload (device trees included) */
int ph_renderer_prepare_ms(ph_renderer *r, double out[5]);
/* The pixel loop (render.rs:127-150) on the GPU, host buffers in and out (see pt_render). */
int ph_renderer_render(ph_renderer *r, const double camera[10], const pt_render_params *params, const double *background,
                       uint8_t *rgb, double *linear, pt_stats *stats);

/* Image::new + Image::render + Image::save with the crate's defaults (env SAMPLES, KD_DEPTH) on an
 * example scene: exercises the whole C++ API the way the reference's main() does. */
int ph_example_render_to_png(const char *name, const char *assets_dir, int n, uint32_t width, uint32_t height, const char *png_path);
int ph_png_read(const char *path, uint32_t size[2], uint8_t *rgb, uint64_t cap);
int ph_png_write(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb);
/* Texture files as texture::RgbImageBuffer::open reads them (src/texture.rs:104-141 via the `image` crate): PNG or
 * JPEG (baseline and progressive), decoded to RGB8. size = {width, height}; rgb may be NULL to query the size. */
int ph_image_read(const char *path, uint32_t size[2], uint8_t *rgb, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
