// Device-resident scene layout (what pt_scene_upload builds in HBM) as seen by the kernels.
//
// All buffers are structure-of-arrays, read-only during a render, sized for residency (a scene is
// uploaded once per pt_scene_upload and stays in HBM / L2 across renders):
//   inv  : n_nodes x 12 f64  rows 0..2 of invtrans      (flat_scene.rs:50-61; read per candidate test)
//   fwd  : n_nodes x 12 f64  rows 0..2 of trans         (read once per shaded hit)
//   nrm  : n_nodes x  9 f64  upper 3x3 of normal_trans  (read once per shaded hit)
//   info : n_nodes x  4 u32  {prim type, prim data, flags, material}
//   tri_v: n_tris  x  9 f64  a, b, c per triangle, 72-byte records (mesh.rs:97-115 expanded once)
//   tri_n: n_tris  x  9 f64  vertex normals (smooth shading only)
//   materials: x 10 f64, lights: x 15 f64
//   textures: RGB8 texels + a 256-entry sRGB->linear table; per-material map indices and uv transform;
//          per-triangle texture coordinates (6 f64) when a textured material uses the mesh
//   bvh / bvh_items: the build's own acceleration structure for FLAT mode (two levels: one tree over
//          the flattened nodes in world space, one tree per mesh in model space)
//   kd / kd_items: the reference's scene k-d tree, linearised (kdtree/node.rs:13-25), for KD mode
#pragma once

#include "pt_math.h"

enum { PT_MODE_FLAT = 1, PT_MODE_KD = 2, PT_MODE_FLAT_NOMESH = 3, PT_MODE_FLAT_KDMESH = 4, PT_MODE_HIER = 5, PT_MODE_HIER_NOMESH = 6, PT_MODE_KD_NOMESH = 7, PT_MODE_HIER_MESH = 8, PT_MODE_KD_MESH = 9 };  // 9: KD for scenes with Mesh instances but no KDMesh trees (2 has both compiled in)  // 8: HIER for scenes with Mesh instances but no KDMesh trees (5 has both compiled in)  // 7: KD for scenes without mesh instances;  // 6: HIER for scenes without mesh instances  // 3: FLAT without mesh instances; 4: FLAT with KDMesh trees (1 has Mesh only)

// Two children per record so one fetch decides both sides. A child reference is one 32-bit word
// (one traversal-stack slot): bit 31 clear = inner node index; bit 31 set = leaf with
// items[first .. first + count), first in bits 30..3, count - 1 in bits 2..0 (1 <= count <= 8).
#define PT_REF_LEAF 0x80000000u
#define PT_REF_EMPTY 0xFFFFFFFFu
// Boxes are f32, rounded OUTWARD at build time: the tree only decides which candidates are tested
// (every accepted hit is produced by the f64 reference arithmetic of the primitive tests), so its
// own arithmetic may be single precision as long as it never rejects a box the f64 ray touches
// (pt_slab32 in pt_trace.h carries the error bounds).
struct PtBvhNode {
    float lo[3][2], hi[3][2];  // [axis][child]: the two children's planes of one axis side by side, so that a wave-uniform walk
                               // feeds one packed f32 fma with the pair straight from scalar registers (pt_trace.h: pt_slab_pk2)
    uint32_t child0, child1;
    uint32_t pad[2];
};  // 64 bytes = one s_load_dwordx16 / four 16-byte loads
PT_HD void pt_node_get_box(const PtBvhNode& n, int child, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; k++) { lo[k] = n.lo[k][child]; hi[k] = n.hi[k][child]; }
}
PT_HD void pt_node_set_box(PtBvhNode& n, int child, const float lo[3], const float hi[3]) {
    for (int k = 0; k < 3; k++) { n.lo[k][child] = lo[k]; n.hi[k][child] = hi[k]; }
}

// What the walks read: FOUR children per record. Built from the two-child tree by pulling every inner child's
// own two children up one level (pt_collapse4_kernel): node i of this array holds the grandchildren of two-child
// node i - a walk takes half as many DEPENDENT node fetches, and the fetch latency, not the box arithmetic, is
// what a wavefront waits for between steps (profiles/r02/notes.md). Boxes as structure-of-arrays [axis][child] so
// that the four slab tests read whole 16-byte vectors; unused slots have an inverted box (never hit) and
// PT_REF_EMPTY. Child references are encoded as in PtBvhNode (an inner reference indexes THIS array).
struct PtBvh4Node {
    float lo[3][4], hi[3][4];
    uint32_t child[4];
    uint32_t pad[4];
};  // 128 bytes = one cache line

struct PtKdNode {
    double plane;          // coordinate of the separating plane on `axis`
    int32_t axis;          // 0..2, or -1 for a leaf
    int32_t front, back;   // children (node.rs:20-22)
    int32_t first, count;  // leaf: kd_items[first .. first+count)
    int32_t pad;
    float box[6];          // union of the (padded, outward-rounded) boxes of everything below this node: the walk's conservative
    float pad2[2];         //   cull reads it with the node (one 64-byte record, one fetch per step)
};  // 64 bytes

struct PtMeshInfo {
    double bbox_inv[12];   // rows 0..2 of BoundingBox::invtrans (bounding_box.rs:55-82)
    uint32_t tri_first, tri_count;
    uint32_t blas_root;    // encoded like PtBvhNode::child0 (may be a leaf, or PT_REF_EMPTY)
    int32_t kd_root;       // KDMesh: root of the mesh's own k-d tree in mkd[] (kdmesh.rs:37-58), or -1
    double kd_extent;      // squared diagonal of that tree's root bounds (node.rs:62-64)
    double kd_bbox_inv[12];  // BoundingBox::invtrans of the root bounds (kdmesh.rs:66-68)
};

static_assert(offsetof(PtMeshInfo, tri_first) == 96 && offsetof(PtMeshInfo, blas_root) == 104 && offsetof(PtMeshInfo, kd_root) == 108,
              "pt_sload_mat12_x4 (pt_trace.h) reads {tri_first, tri_count, blas_root, kd_root} as the 16 bytes behind bbox_inv");

struct PtTexInfo {  // one RgbImageBuffer (texture.rs:74-76) inside tex_rgb
    uint64_t offset;
    uint32_t width, height;
};

// What the out-of-line texture routine (pt_apply_maps) reads, behind ONE pointer: a non-inlined device function gets its
// arguments in 32 VGPRs and everything beyond that through scratch memory, so its argument list is kept short.
struct PtTexInfo;
struct PtTexView {
    const PtTexInfo* tex;
    const uint8_t* tex_rgb;
    const double* uv_trans;
    const double* tri_v;
    const double* tri_uv;
    const int32_t* mat_maps;
};

struct PtSceneView {
    uint32_t n_nodes, n_lights;
    const double* inv;
    const double* fwd;
    const double* nrm;
    const uint32_t* info;
    const double* tri_v;
    const double* tri_e;   // n_tris x 9 f64: corner a, a - b, a - c (pt_triangle_hit_e): what the wave-uniform walks test triangles from
    const double* own_inv;   // hierarchical semantics: per flattened node the inverse of its OWN (last) path level, 12 f64 - what most leaf tests need, fetched with the path record
    const double* tri_leaf;  // one 80-byte record per slot of bvh_items (round 4): the edge record of the triangle the slot names + its index (dword 18) - a mesh leaf's
                             // triangles side by side, without the look-up through bvh_items (slots that name no triangle: zeros)
    const double* tri_n;
    const PtMeshInfo* meshes;
    const double* materials;
    const double* lights;
    double ambient[3];
    const PtBvhNode* bvh;     // the two-child tree as built (host binned SAH / device PLOC); the walks read bvh4
    const PtBvh4Node* bvh4;   // its four-child form, same node indices
    const uint32_t* bvh_items;
    uint32_t tlas_root;
    uint32_t mesh_oct;     // 1: inside mesh instances the triangle trees are walked with the octant-sorted slab test (pt_trace_packet_mesh; PORTRAYER_MESH_OCT=0 turns it off)
    uint32_t tlas_direct;  // 1: a leaf of the scene-level tree names its flattened node in the reference itself (bits 30..3)
    const PtKdNode* kd;
    const uint32_t* kd_items;
    // PT_MODE_HIER (scene.rs:80-120): each SceneNode's OWN matrices and every flattened node's path through them
    const double* g_inv;        // per graph node: rows 0..2 of its invtrans (12)
    const double* g_fwd;        // per graph node: rows 0..2 of its trans (12)
    const double* g_nrm;        // per graph node: upper 3x3 of its normal_trans (9)
    const uint32_t* chain_off;  // n_nodes + 1
    const uint32_t* chain;      // graph node indices, root first
    const uint32_t* dfs_rank;   // per flattened node: depth-first order, a node before its children (who wins equal hits)
    const uint32_t* hier_rec;   // per flattened node 8 words: levels on its path (bits 0-7; 255 = more than 7: use chain_off / chain) | bit 8 + k set where
                                // level k's three matrices are the identity; then the graph node indices of levels 0..6, root first (pt_trace.h)
    const float* kd_box;     // KD mode: non-null = PtKdNode::box is valid and the walk culls with it (the array itself is a copy kept for tools); else null
    const float* node_box;   // KD mode: per kd_items entry, that node's padded world box as 6 f32 rounded outward (leaf pre-cull in pt_trace_kd); else null
    double kd_extent;  // bounding_box.rs:95-99: squared diagonal of the root bounds
    const uint32_t* kd_ref;  // KD mode: per kd_items entry 8 words {flattened node, 0, its padded world box as 6 f32 rounded outward}: one scalar fetch per leaf reference (pt_trace_packet_kd)
    int32_t kd_levels;       // KD mode: split levels on the deepest path of the scene's k-d tree
    int32_t mode;
    int32_t stack_cap;  // entries per lane in the traversal stack
    // texture.rs (all null when the scene has no textured material)
    const int32_t* mat_maps;   // n_materials x 2: texture index, normal-map index (-1 = none)
    const double* uv_trans;    // n_materials x 9, row-major Mat3 (material.rs:83)
    const PtTexInfo* tex;      // per texture
    const uint8_t* tex_rgb;    // RGB8 texels of all textures
    const double* srgb_lut;    // 256 entries: (k / 255)^2.2 computed on the host (texture.rs:162-168)
    const double* tri_uv;      // n_tris x 6: texture coordinates of a, b, c (mesh.rs:30, triangle.rs:18)
    const PtTexView* texview;  // the same pointers in device memory, for pt_apply_maps
    // KDMesh triangle trees, the reference's structure (kdtree/kdmesh.rs); items are global triangle indices
    const PtKdNode* mkd;
    const uint32_t* mkd_items;
    const float* mkd_box;       // per mkd node: union of the padded triangle boxes below it (6 f32, outward); null = no culling
    const float* mkd_item_box;  // per mkd_items entry: that triangle's padded box
};

struct PtCamera {  // camera.rs:17-31, built on the host (look_at inverse, tan)
    double eye[3];
    double view_to_world[12];  // rows 0..2
    double fov_factor, aspect, width, height;
};

// Per-lane counters, SURVEY §8(d) ray accounting
struct PtCounters {
    unsigned long long primary, shadow, reflect, refract, depth11_skipped, hits;
    unsigned long long n_inner, n_leaf, n_analytic, n_tri, n_bbox, kd_plane_miss, stack_overflow, kd_culled;
    unsigned long long diag[8];  // -DPT_DIAG builds: 0 trace calls (per wave), 1 lanes carrying a ray into them, 2 interpreter calls (per wave),
                                 // 3 lanes active in them, 4 inner-node steps (per wave; per lane = n_inner), 5 leaf steps (per wave; per lane = n_leaf)
};
