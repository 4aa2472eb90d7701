/* portrayer_hip.h — C ABI of the MI355X (gfx950) ray-cast / shade path.
 *
 * This library replaces the body of the reference's pixel loop and everything it calls:
 *   ImageSliceMut::render           src/render.rs:127-150   (the rayon loop over pixels)
 *   render_single_pixel             src/render.rs:22-51
 *   Camera::ray_at                  src/camera.rs:48-84
 *   Ray::color, RayCast / RayHit    src/ray.rs:39-148
 *   FlatSceneNode::ray_cast         src/flat_scene.rs:71-99
 *   KDTreeNode::ray_cast_impl       src/kdtree/node.rs:66-203
 *   Primitive::ray_hit dispatch     src/primitive.rs:55-62 and src/primitive/{sphere,triangle,mesh,cube,plane,cylinder,cone}.rs
 *   BoundingBox::test_hit           src/bounding_box.rs:104-116
 *   Material::hit_color             src/material.rs:91-320
 * The reference has no FFI of its own (it is one Rust crate); these entry points are what a
 * binding placed where render.rs:127-150 is today would call — INTEGRATION.md shows that binding.
 * Host-side work that the reference does once per render stays with the caller and crosses this
 * boundary as plain arrays: flattening (src/flat_scene.rs:18-46), matrix inverses, bounding boxes
 * (src/bounding_box.rs:55-82, :123-148), the scene k-d tree build (src/kdtree/leaf.rs:89-231,
 * src/kdtree/kdscene.rs:19-43), the camera matrix (src/camera.rs:34-45) and the background closure
 * evaluated per integer pixel (src/render.rs:31-34).
 *
 * Conventions: every array is caller-owned, read-only for the duration of the call and may be
 * freed afterwards; matrices are row-major 4x4 (16 doubles); no callbacks, no unwinding: every
 * function returns 0 or a negative PT_ERR_* code and pt_last_error() describes the failure.
 * One context per GPU; calls on one context must be serialised by the caller.
 */
#ifndef PORTRAYER_HIP_H
#define PORTRAYER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 8

/* enum Primitive, src/primitive.rs:67-81 */
enum { PT_PRIM_SPHERE = 0, PT_PRIM_TRIANGLE = 1, PT_PRIM_MESH = 2, PT_PRIM_KDMESH = 3, PT_PRIM_PLANE = 4, PT_PRIM_CUBE = 5, PT_PRIM_CYLINDER = 6, PT_PRIM_CONE = 7 };
/* cargo features flat_scene / kdtree, src/render.rs:121-126 */
enum { PT_TRAVERSE_FLAT = 1, PT_TRAVERSE_KD = 2, PT_TRAVERSE_HIER = 3 };
/* sample position inside the pixel (the reference draws it from thread_rng, src/render.rs:38-39) */
enum { PT_SAMPLE_CENTRE = 0, PT_SAMPLE_RNG = 1 };

enum {
    PT_OK = 0,
    PT_ERR_ARGUMENT = -1,   /* null pointer, bad index, bad enum                                     */
    PT_ERR_DEVICE = -2,     /* HIP runtime error (text in pt_last_error)                             */
    PT_ERR_NO_SCENE = -3,   /* pt_render before pt_scene_upload                                      */
    PT_ERR_SLICE = -4,      /* slice corner outside the image: ImageSliceMut::new panics, render.rs:79-90 */
    PT_ERR_SCENE = -5,      /* inconsistent scene (mesh without vertices, mesh.rs:71; smooth shading without normals, mesh.rs:135-138;
                               a textured material on a primitive without texture coordinates, material.rs:133,141) */
    PT_ERR_TRAVERSAL = -6   /* a lane ran out of traversal stack (never expected; results invalid)    */
};

typedef struct pt_context pt_context;

/* The flattened scene: Scene<Vec<FlatSceneNode>> (src/flat_scene.rs:16, :50-61), nodes in the
 * breadth-first order FlatScene::from emits them (src/flat_scene.rs:18-46). */
typedef struct {
    uint32_t n_nodes;
    const double *trans;          /* n_nodes x 16  FlatSceneNode::trans                                  */
    const double *invtrans;       /* n_nodes x 16  trans.inverted()                (flat_scene.rs:104)    */
    const double *normal_trans;   /* n_nodes x 16  invtrans.transposed()           (flat_scene.rs:105)    */
    const int32_t *prim_type;     /* PT_PRIM_*                                                           */
    const int32_t *prim_data;     /* MESH/KDMESH: mesh index; TRIANGLE: triangle index; else ignored     */
    const int32_t *prim_flags;    /* bit 0: Shading::Smooth (mesh.rs:11-18) / triangle has vertex normals */
    const int32_t *material;      /* index into materials                                                */
    /* MeshData, src/primitive/mesh.rs:21-34 */
    uint32_t n_meshes;
    const uint64_t *mesh_vert_off;   /* n_meshes + 1, in vertices                                         */
    const uint64_t *mesh_tri_off;    /* n_meshes + 1, in triangles                                        */
    const double *mesh_positions;    /* total vertices x 3                                                */
    const double *mesh_normals;      /* total vertices x 3, or NULL when no mesh has normals              */
    const uint8_t *mesh_has_normals; /* n_meshes, or NULL                                                 */
    const uint32_t *mesh_indices;    /* total triangles x 3, indices local to the mesh                    */
    const double *mesh_bounds_invtrans; /* n_meshes x 16: BoundingBox::invtrans of the mesh AABB (bounding_box.rs:55-82) */
    /* stand-alone Triangle primitives, src/primitive/triangle.rs:8-19 */
    uint32_t n_triangles;
    const double *tri_vertices;      /* n_triangles x 9: a, b, c                                          */
    const double *tri_normals;       /* n_triangles x 9, or NULL                                          */
    /* Material hot fields, src/material.rs:50-86: diffuse rgb, specular rgb, shininess, reflectivity,
     * glossy_side_length, refraction_index */
    uint32_t n_materials;
    const double *materials;         /* n_materials x 10                                                  */
    /* Light, src/light.rs:74-91: position, color, falloff c0 c1 c2, area.a, area.b */
    uint32_t n_lights;
    const double *lights;            /* n_lights x 15                                                     */
    double ambient[3];               /* Scene::ambient, src/scene.rs:17                                   */
    /* Image textures and normal maps, src/texture.rs (ABI 2). All optional: leave NULL / 0 for scenes
     * without textured materials. Texels are the RGB8 pixels as the image decoder returns them
     * (RgbImageBuffer, texture.rs:74-76); sRGB -> linear (texture.rs:162-168) happens at sampling. */
    const double *mesh_texcoords;       /* total vertices x 2 (MeshData::tex_coords, mesh.rs:30), or NULL  */
    const uint8_t *mesh_has_texcoords;  /* n_meshes, or NULL                                               */
    const double *tri_texcoords;        /* n_triangles x 6 (Triangle::tex_coords, triangle.rs:18), or NULL */
    const uint8_t *tri_has_texcoords;   /* n_triangles, or NULL                                            */
    const int32_t *material_texture;    /* n_materials: texture index or -1 (Material::texture, material.rs:75) */
    const int32_t *material_normal_map; /* n_materials: texture index or -1 (Material::normals, material.rs:85) */
    const double *material_uv_trans;    /* n_materials x 9 row-major Mat3 (material.rs:83); NULL = identity  */
    uint32_t n_textures;
    const uint32_t *texture_size;       /* n_textures x 2: width, height                                   */
    const uint64_t *texture_offset;     /* n_textures: byte offset of texel (0,0) in texture_rgb           */
    const uint8_t *texture_rgb;         /* all texels, row-major RGB8                                      */
    /* KDMesh triangle trees (ABI 3): KDMesh::new builds a k-d tree over the mesh's triangles
     * (src/kdtree/kdmesh.rs:37-58, KD_MESH_DEPTH) whose traversal can miss hits a Mesh finds (squared
     * extent, bounding_box.rs:95-99); to reproduce that the host passes the trees, linearised like
     * pt_kdtree with node indices into the shared kdm_* arrays. Optional: a KDMesh whose mesh has
     * mesh_kd_root < 0 (or NULL arrays) is traversed like a Mesh. */
    const int32_t *mesh_kd_root;        /* n_meshes: root node index in kdm_*, or -1                       */
    const int32_t *mesh_kd_depth;       /* n_meshes: Split levels above the deepest leaf                   */
    const double *mesh_kd_bounds;       /* n_meshes x 6: root bounds min, max                              */
    const double *mesh_kd_bounds_invtrans; /* n_meshes x 16: BoundingBox::invtrans of the root bounds      */
    uint32_t n_kdm_nodes;
    const int32_t *kdm_axis; const double *kdm_plane; const int32_t *kdm_front, *kdm_back, *kdm_first, *kdm_count;
    uint32_t n_kdm_items;
    const int32_t *kdm_items;           /* triangle indices local to the mesh, in leaf Vec order            */
    /* The scene GRAPH (ABI 4), for PT_TRAVERSE_HIER: the reference's default traversal (no `flat_scene` /
     * `kdtree` feature) transforms the ray level by level down the hierarchy with each SceneNode's OWN inverse
     * and carries hit point and normal back up level by level (src/scene.rs:80-120). That rounds differently
     * from one composed matrix per flattened node, and where a ray starts on a refractive surface the
     * difference decides hits. NULL / 0 for the other traversals. */
    uint32_t n_graph_nodes;             /* distinct SceneNodes on the paths to the flattened nodes            */
    const double *graph_trans;          /* n_graph_nodes x 16  SceneNode::trans        (scene.rs:150-160)     */
    const double *graph_invtrans;       /* n_graph_nodes x 16  SceneNode::invtrans                            */
    const double *graph_normal_trans;   /* n_graph_nodes x 16  SceneNode::normal_trans                        */
    const uint32_t *node_chain_off;     /* n_nodes + 1: flattened node i's path is node_chain[off[i] .. off[i+1]) */
    const uint32_t *node_chain;         /* graph node indices from the root down to the node itself           */
    const uint32_t *node_dfs_rank;      /* n_nodes: place in depth-first order, a node before its children:   */
                                        /*   the order in which equal hits are resolved (ray.rs:87-99 under scene.rs:95-117) */
} pt_scene;

/* The scene k-d tree the host built (KDTreeScene::from, src/kdtree/kdscene.rs:19-43), linearised;
 * node 0 is the root. Needed for PT_TRAVERSE_KD only.
 * Limits (pt_scene_upload fails with PT_ERR_SCENE beyond them): at most 32 split levels on any path
 * from the root (the reference's KD_DEPTH defaults to 10, kdscene.rs:36; a tree deeper than 32 has more
 * than 2^32 leaves unless it is a degenerate chain), fewer than 2^26 nodes, fewer than 2^27 leaf items. */
typedef struct {
    uint32_t n_nodes;
    const int32_t *axis;      /* 0,1,2: KDTreeNode::Split on that axis; -1: KDTreeNode::Leaf (node.rs:13-25) */
    const double *plane;      /* Split: sep_plane.point on `axis`                                         */
    const int32_t *front;     /* Split: front_nodes                                                       */
    const int32_t *back;      /* Split: back_nodes                                                        */
    const int32_t *first;     /* Leaf: its nodes are leaf_items[first .. first + count), in Vec order      */
    const int32_t *count;
    uint32_t n_items;
    const int32_t *leaf_items; /* flat node indices                                                       */
    double root_min[3], root_max[3]; /* root bounds; extent() = squared diagonal (bounding_box.rs:95-99)   */
    int32_t max_depth;        /* Split levels above the deepest leaf (sizes the traversal stack)          */
} pt_kdtree;

/* Camera, src/camera.rs:17-31, as Camera::new (camera.rs:34-45) computes it */
typedef struct {
    double eye[3];
    double view_to_world[16];
    double fov_factor;    /* tan(fovy / 2)  */
    double aspect_ratio;  /* width / height */
    double width, height;
} pt_camera;

typedef struct { uint32_t x0, y0, x1, y1; } pt_rect; /* inclusive corners like ImageSliceMut, render.rs:56-66 */

typedef struct {
    uint32_t width, height;     /* Image::width / height                                                 */
    pt_rect slice;              /* pixels to render; others are left untouched (render.rs:135-138)        */
    uint32_t samples;           /* env SAMPLES (render.rs:107-113), > 0                                   */
    uint64_t seed;              /* key of the counter-based sample generator                             */
    int32_t sample_mode;        /* PT_SAMPLE_*                                                            */
    int32_t background_rows;    /* 1: background is height x 3 (one colour per row); 0: height x width x 3 */
    uint32_t tile_rank;         /* multi-GPU: render only the 8x8 tiles t of the slice with              */
    uint32_t tile_ranks;        /*   t % tile_ranks == tile_rank (1 GPU: 0 of 1)                          */
    int32_t collect_stats;      /* 1: run the counting build of the kernel and fill ray/test counters     */
} pt_render_params;

typedef struct {
    uint64_t primary, shadow, reflect, refract; /* rays traced                                           */
    uint64_t depth11_skipped;   /* depth-11 rays the reference would trace and discard (material.rs:102-104) */
    uint64_t hits;              /* shaded hits                                                           */
    uint64_t n_inner, n_leaf;   /* tree nodes visited (bounding-volume nodes in FLAT, k-d nodes in KD)    */
    uint64_t n_analytic;        /* flat-node candidate tests (ray transform + primitive dispatch)         */
    uint64_t n_tri;             /* triangle tests                                                         */
    uint64_t n_bbox;            /* mesh bounding-box tests                                                */
    uint64_t kd_plane_miss;     /* places where the reference would panic (node.rs:146-147, :177-178)      */
    uint64_t stack_overflow;    /* must be 0                                                              */
    double kernel_ms;           /* device time of the render kernel (HIP events)                          */
    double total_ms;            /* upload of per-call inputs + kernel + read-back                         */
    uint64_t diag[8];           /* (ABI 5) lane-occupancy diagnostics of -DPT_DIAG builds (profiles/diag.sh); 0 otherwise */
    uint32_t kernel_mode;       /* (ABI 6) which instantiation of the render kernel ran: PT_KERNEL_MODE_* ...                */
    uint32_t kernel_variant;    /* ... and PT_KERNEL_* (tests assert that a timed configuration is the one they checked)    */
} pt_stats;

/* pt_stats.kernel_mode: the walk the kernel was compiled with */
enum { PT_KERNEL_MODE_FLAT = 1, PT_KERNEL_MODE_KD = 2, PT_KERNEL_MODE_FLAT_NOMESH = 3, PT_KERNEL_MODE_FLAT_KDMESH = 4, PT_KERNEL_MODE_HIER = 5,
       PT_KERNEL_MODE_HIER_NOMESH = 6, PT_KERNEL_MODE_KD_NOMESH = 7,
       PT_KERNEL_MODE_HIER_MESH = 8 /* hierarchical, Mesh instances but no KDMesh trees (5 has both compiled in) */,
       PT_KERNEL_MODE_KD_MESH = 9 /* kdtree semantics, Mesh instances but no KDMesh trees (2 has both compiled in) */ };
/* pt_stats.kernel_variant: bit 0-3 waves per SIMD the kernel was compiled for (3 or 4); PT_KERNEL_INTERPRETER: the per-lane
 * interpreter that scenes with reflective materials need (material.rs:216-303), else the straight-line kernel; PT_KERNEL_PARK:
 * a parked recursion frame per lane in LDS; PT_KERNEL_COUNTING: the counting build (collect_stats); PT_KERNEL_TEXTURED */
enum { PT_KERNEL_WAVES_MASK = 15, PT_KERNEL_INTERPRETER = 16, PT_KERNEL_PARK = 32, PT_KERNEL_COUNTING = 64, PT_KERNEL_TEXTURED = 128,
       PT_KERNEL_FORK = 256 /* idle lanes take the refracted subtrees busy lanes offer (LDS queue, ballot / popcount ranks) */,
       PT_KERNEL_CHAIN = 512 /* the straight-line kernel with a loop over the depth: every reflective material of the scene is opaque */ };

int pt_abi_version(void);
int pt_device_count(void);

int pt_context_create(int device, pt_context **out);
void pt_context_destroy(pt_context *ctx);
const char *pt_last_error(const pt_context *ctx);

/* Uploads the scene into HBM and builds the traversal structures. `traverse` = PT_TRAVERSE_*;
 * `kd` must be non-NULL for PT_TRAVERSE_KD. Replaces any scene uploaded before. */
int pt_scene_upload(pt_context *ctx, const pt_scene *scene, int traverse, const pt_kdtree *kd);

/* Renders with host buffers. background: per pt_render_params.background_rows. rgb: height x width
 * x 3 bytes, only pixels of the slice that belong to this tile rank are written. linear (optional):
 * height x width x 3 doubles, the sample mean before gamma (render.rs:45). */
int pt_render(pt_context *ctx, const pt_camera *camera, const double *background, const pt_render_params *params,
              uint8_t *rgb, double *linear, pt_stats *stats);

/* Same, writing into DEVICE memory on `hip_stream` (a hipStream_t, or NULL for the default stream)
 * without synchronising the host: d_rgb is either the full image (compact = 0) or this rank's tiles
 * only, tile-major, 8 x 8 x 3 bytes per tile (compact = 1; size = pt_compact_bytes()). The
 * background must already be resident: d_background is a DEVICE pointer. stats (optional) are
 * filled only if the call is followed by pt_render_finish(). */
int pt_render_device(pt_context *ctx, const pt_camera *camera, const double *d_background, const pt_render_params *params,
                     int compact, void *d_rgb, void *hip_stream);
int pt_render_finish(pt_context *ctx, pt_stats *stats);
/* (ABI 8) Two renders of a context may be in flight at once - pt_render_device, pt_render_device, pt_render_finish (the OLDER one), ... - and since
 * ABI 8 each of the two owns its work buffers, so that they may also run on two streams: pt_context_stream(ctx, k) is the context's own
 * non-blocking stream (a hipStream_t) for the render that takes slot k & 1; pt_context_next_slot() says which slot the next pt_render_device
 * takes (they are taken in turn). The render kernels are persistent: a frame's wavefronts retire one by one over the duration of its longest work
 * items, and a next frame queued on the OTHER stream starts in the places they free instead of waiting for the last of them (the reference
 * renders one image per call, src/render.rs:93-151; consecutive calls are independent, which is what lets two be in flight). For scenes whose
 * kernels park recursion frames in HBM both slots get the SAME stream (two launches at once cost such scenes more than the tail is worth). */
void *pt_context_stream(pt_context *ctx, int slot);
int pt_context_next_slot(const pt_context *ctx);

/* Bytes of one rank's compact tile buffer for a slice split over tile_ranks ranks (equal for all ranks). */
uint64_t pt_compact_bytes(const pt_render_params *params);
/* Scatters the gathered compact buffers (rank-major) into a row-major image on the device. */
int pt_untile_device(pt_context *ctx, const pt_render_params *params, const void *d_gathered, void *d_rgb, void *hip_stream);

/* Host-side view of the tile partition (no GPU needed): pixel of work slot `slot` of rank `rank`
 * (slot = local tile * 64 + position in the 8x8 tile). Returns 1 and fills x, y; 0 for a padding slot
 * (outside the slice / beyond the last tile); < 0 on bad arguments. */
int pt_tile_slot_pixel(const pt_render_params *params, uint32_t rank, uint32_t slot, uint32_t *x, uint32_t *y);
/* Host version of pt_untile_device: gathered = tile_ranks x pt_compact_bytes(), rank-major. */
int pt_untile_host(const pt_render_params *params, const uint8_t *gathered, uint8_t *rgb);

/* ---- One render call over the GPUs of a node (ABI 5). One context per GPU, the scene replicated; the slice's 8x8
 * tiles are dealt round-robin to the ranks, each rank renders its tiles into a compact buffer on its own stream, ONE
 * RCCL gather (ncclGather over xGMI, single process) brings them to rank 0, which untiles them. The slice API of the
 * reference (src/render.rs:56-66, :211-213) is the rectangular special case this generalises. `devices` = n_devices
 * device indices (NULL: 0 .. n_devices - 1); ranks may share a device (then the gather is device-to-device copies:
 * RCCL needs distinct GPUs). RCCL is loaded on first use. */
typedef struct pt_node pt_node;
int pt_node_create(int n_devices, const int *devices, pt_node **out);
void pt_node_destroy(pt_node *node);
const char *pt_node_last_error(const pt_node *node);
int pt_node_ranks(const pt_node *node);
int pt_node_uses_rccl(const pt_node *node);          /* 1: the gather is RCCL; 0: ranks share a device, copies */
pt_context *pt_node_context(pt_node *node, int rank);
int pt_node_scene_upload(pt_node *node, const pt_scene *scene, int traverse, const pt_kdtree *kd);
/* Like pt_render (host buffers; params->tile_rank / tile_ranks must be 0 / 1: the node partitions the tiles itself).
 * stats: counters summed over the ranks, kernel_ms of the slowest rank, total_ms of the whole call. */
int pt_node_render(pt_node *node, const pt_camera *camera, const double *background, const pt_render_params *params,
                   uint8_t *rgb, pt_stats *stats);
/* (ABI 6) The same call in its three parts, for callers that render many frames of one size (and for measuring the frame with its
 * inputs resident in HBM): pt_node_upload_background copies the background to every rank (and, when rgb is not NULL, the
 * caller's image to rank 0, so that pixels outside the slice keep their bytes); pt_node_render_resident renders, gathers and
 * untiles into the image resident on rank 0 and returns when it is complete - no host buffer is touched; pt_node_download_image
 * copies that image out. pt_node_device: the device index of a rank (< 0: no such rank). */
int pt_node_upload_background(pt_node *node, const double *background, const pt_render_params *params, const uint8_t *rgb);
int pt_node_render_resident(pt_node *node, const pt_camera *camera, const pt_render_params *params, pt_stats *stats);
int pt_node_download_image(pt_node *node, const pt_render_params *params, uint8_t *rgb);
int pt_node_device(const pt_node *node, int rank);
/* (ABI 7) Frames in a pipeline. pt_node_frame_begin queues a frame - every rank's render (launched by the rank's own host thread), the ONE
 * gather and the untile - and returns without waiting; pt_node_frame_end closes the OLDEST open frame: it returns when that frame's image is
 * complete on rank 0, with its stats (counters summed over the ranks, kernel_ms of the slowest rank). Up to two frames may be open: their
 * tile buffers are separate and the gather runs on streams of its own, so frame k + 1 renders while frame k is gathered and untiled, and
 * the host's launch work disappears behind the GPUs' (render.rs:93-151 renders one frame per call: pt_node_render_resident = begin + end).
 * The image on rank 0 is a single buffer: pt_node_download_image (and pt_node_upload_background) need every frame closed.
 * pt_node_last_frame_host_ms: host milliseconds of the last frame's calls - [0] begin as a whole, [1] end blocked until the image was
 * complete, [2] end after that, [3] the slowest rank's launch inside begin, [4] the ranks' kernel times added up. */
int pt_node_frame_begin(pt_node *node, const pt_camera *camera, const pt_render_params *params);
int pt_node_frame_end(pt_node *node, pt_stats *stats);
int pt_node_frames_in_flight(const pt_node *node);
int pt_node_last_frame_host_ms(const pt_node *node, double out[5]);
/* (ABI 8) every rank's kernel time of the last frame closed, milliseconds (HIP events around the rank's launches); n_out = pt_node_ranks() */
int pt_node_last_frame_rank_kernel_ms(const pt_node *node, double *out, int n_out);

/* Device-side helpers used by the measurement harness. */
int pt_device_alloc(pt_context *ctx, uint64_t bytes, void **out);
int pt_device_free(pt_context *ctx, void *ptr);
int pt_copy_to_device(pt_context *ctx, void *dst, const void *src, uint64_t bytes);
int pt_copy_from_device(pt_context *ctx, void *dst, const void *src, uint64_t bytes);
int pt_synchronize(pt_context *ctx); /* waits for everything queued on the context's device */
/* Streams `bytes` from src to dst with 16-byte accesses `iters` times and returns the best GB/s
 * (read + write counted), the measured HBM roofline the renderer is compared with. */
int pt_measure_copy_bandwidth(pt_context *ctx, uint64_t bytes, int iters, double *gbps);

/* Self-test entry points for the parity tests: run device arithmetic on explicit inputs. */
int pt_test_cast_rays(pt_context *ctx, uint64_t n, const double *origins, const double *directions, int any_hit,
                      double *out_t, int32_t *out_node, int32_t *out_sub);
int pt_test_math(pt_context *ctx, int op, uint64_t n, const double *a, const double *b, double *out);
/* (ABI 6) No GPU, no context: x[i]^y[i] by the HOST build of the kernels' pow (csrc/pt_pow.h) in `port` and by this machine's
 * libm in `libm` - the pin of the restated glibc algorithm against the library the reference calls. */
int pt_test_pow_host(uint64_t n, const double *x, const double *y, double *port, double *libm);
/* (ABI 7) the host's libm on explicit inputs (2 pow, 4 atan2, 5 acos): what device results are compared with */
int pt_test_libm_host(int op, uint64_t n, const double *a, const double *b, double *out);
/* Host-side replay (no GPU, no context) of how a launch with these parameters lays its work items and their 64 lanes over pixels,
 * chunks and samples - the kernel's own indexing code. Arrays of width x height, zeroed by the caller: samples carried per pixel,
 * the sum of their indices, the sum of the chunk lengths reported by the lanes that add a chunk up; optionally the number of work
 * items and {pixels, chunks, samples} per wavefront. */
int pt_test_work_items(const pt_render_params *params, uint32_t *sample_count, uint64_t *sample_index_sum, uint32_t *chunk_length_sum,
                       uint64_t *n_items, uint32_t *lane_pixels_chunks_samples);

#ifdef __cplusplus
}
#endif
#endif
