/* Replays, from plain C, exactly what the Rust shim (shim/src/hip_pack.rs + shim/overlay/render.rs.append) does with
 * libportrayer_hip.so: the call sequence of render_hip - pt_context_create, pt_scene_upload (PT_TRAVERSE_HIER with the
 * scene-graph arrays: the crate's default feature set), pt_render with a per-pixel background, pt_context_destroy - and
 * hip_pack's array layouts: flattened nodes in FlatScene::from's breadth-first order, row-major 4x4 matrices, materials
 * and meshes numbered by first use, a Mesh as a TRIANGLE LIST (three vertices per triangle, indices 0 1 2 ...),
 * node_chain / node_dfs_rank of an instanced subtree.
 *
 * Every number of the scene is a short dyadic fraction, so every matrix product and inverse below is exact and does not
 * depend on how it is computed; tests/test_shim_replay.py renders the same scene (tests/scene_dsl) with the oracle and
 * compares the bytes. Test infrastructure: built and run by that test on the GPU box.
 *
 *   gcc -O1 -I include tests/shim_replay.c -L portrayer_amd -lportrayer_hip -Wl,-rpath,$PWD/portrayer_amd -lm -o shim_replay
 *   ./shim_replay out.rgb W H SAMPLES
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "portrayer_hip.h"

typedef struct { double m[16]; } M4; /* row-major */

static M4 ident(void) { M4 r; memset(&r, 0, sizeof r); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r; }
static M4 mul(M4 a, M4 b) {
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += a.m[4 * i + k] * b.m[4 * k + j];
        r.m[4 * i + j] = s;
    }
    return r;
}
static M4 scaling(double x, double y, double z) { M4 r = ident(); r.m[0] = x; r.m[5] = y; r.m[10] = z; return r; }
static M4 translation(double x, double y, double z) { M4 r = ident(); r.m[3] = x; r.m[7] = y; r.m[11] = z; return r; }
/* SceneNode::scaled / translated (scene.rs:163-174): trans = Op * trans */
typedef struct { M4 trans, inv; } Xf;
static Xf xf_new(void) { Xf x = {ident(), ident()}; return x; }
static Xf xf_scaled(Xf x, double s) { x.trans = mul(scaling(s, s, s), x.trans); x.inv = mul(x.inv, scaling(1.0 / s, 1.0 / s, 1.0 / s)); return x; }
static Xf xf_scaled3(Xf x, double a, double b, double c) { x.trans = mul(scaling(a, b, c), x.trans); x.inv = mul(x.inv, scaling(1.0 / a, 1.0 / b, 1.0 / c)); return x; }
static Xf xf_translated(Xf x, double a, double b, double c) { x.trans = mul(translation(a, b, c), x.trans); x.inv = mul(x.inv, translation(-a, -b, -c)); return x; }
static M4 transposed(M4 a) { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[4 * i + j] = a.m[4 * j + i]; return r; }
static M4 clean(M4 a) { for (int i = 0; i < 16; i++) if (a.m[i] == 0.0) a.m[i] = 0.0; return a; } /* -0.0 -> +0.0 */

#define MAXN 16
static double g_trans[MAXN * 16], g_inv[MAXN * 16], g_nrm[MAXN * 16];      /* flattened nodes */
static double gr_trans[MAXN * 16], gr_inv[MAXN * 16], gr_nrm[MAXN * 16];   /* graph (per SceneNode) */
static int32_t prim_type[MAXN], prim_data[MAXN], prim_flags[MAXN], material[MAXN];
static uint32_t chain_off[MAXN + 1], chain[MAXN * 4], dfs_rank[MAXN];
static uint32_t n_nodes = 0, n_graph = 0, n_chain = 0;

static uint32_t graph_node(Xf own) { /* one entry per distinct SceneNode: its OWN matrices */
    memcpy(gr_trans + 16 * n_graph, clean(own.trans).m, 128);
    memcpy(gr_inv + 16 * n_graph, clean(own.inv).m, 128);
    memcpy(gr_nrm + 16 * n_graph, clean(transposed(own.inv)).m, 128);
    return n_graph++;
}
static void flat_node(M4 total, M4 total_inv, int type, int data, int flags, int mat, const uint32_t* path, int path_len, uint32_t rank) {
    memcpy(g_trans + 16 * n_nodes, clean(total).m, 128);
    memcpy(g_inv + 16 * n_nodes, clean(total_inv).m, 128);
    memcpy(g_nrm + 16 * n_nodes, clean(transposed(total_inv)).m, 128);
    prim_type[n_nodes] = type; prim_data[n_nodes] = data; prim_flags[n_nodes] = flags; material[n_nodes] = mat;
    for (int k = 0; k < path_len; k++) chain[n_chain++] = path[k];
    chain_off[n_nodes + 1] = n_chain;
    dfs_rank[n_nodes] = rank;
    n_nodes++;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s out.rgb W H SAMPLES\n", argv[0]); return 2; }
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), S = (uint32_t)atoi(argv[4]);

    /* ---- the scene (the same numbers as tests/test_shim_replay.py) -------------------------------------------------
     * root R .translated(0, -0.5, 0)
     *   A  Sphere  mat0  .scaled(2).translated(-2.5, 1.5, 0)
     *   B  group   .scaled(0.5).translated(2, 0, -1)
     *        B0 Cube mat1 .scaled(2)
     *        B1 Mesh(tent, flat) mat2 .translated(0, 2, 0)
     *   C  group   .translated(-1, 3, -2)  with child B (the same B: instancing)
     *   T  Triangle mat0 (stand-alone)
     *   P  Plane   mat3 (mirror) .scaled(16).translated(0, -1, 0)
     * breadth-first order of the flattened nodes (flat_scene.rs:18-46): A, T, P, B0, B1, (C/B)0, (C/B)1            */
    Xf R = xf_translated(xf_new(), 0, -0.5, 0);
    Xf A = xf_translated(xf_scaled(xf_new(), 2), -2.5, 1.5, 0);
    Xf B = xf_translated(xf_scaled(xf_new(), 0.5), 2, 0, -1);
    Xf B0 = xf_scaled(xf_new(), 2);
    Xf B1 = xf_translated(xf_new(), 0, 2, 0);
    Xf Cn = xf_translated(xf_new(), -1, 3, -2);
    Xf T = xf_new();
    Xf P = xf_translated(xf_scaled(xf_new(), 16), 0, -1, 0);
    /* graph ids in hip_pack::pack_graph's order (first visit of the breadth-first walk): R A B C T P B0 B1 */
    uint32_t gR = graph_node(R), gA = graph_node(A), gB = graph_node(B), gC = graph_node(Cn), gT = graph_node(T), gP = graph_node(P);
    uint32_t gB0 = graph_node(B0), gB1 = graph_node(B1);
    M4 tR = R.trans, iR = R.inv;
    M4 tRB = mul(tR, B.trans), iRB = mul(B.inv, iR);
    M4 tRCB = mul(mul(tR, Cn.trans), B.trans), iRCB = mul(B.inv, mul(Cn.inv, iR));
    /* depth-first ranks (a node before its children; children in order A, B, C, T, P): A 0, B0 1, B1 2, (C/B)0 3, (C/B)1 4, T 5, P 6 */
    { uint32_t p[] = {gR, gA}; flat_node(mul(tR, A.trans), mul(A.inv, iR), PT_PRIM_SPHERE, 0, 0, 0, p, 2, 0); }
    { uint32_t p[] = {gR, gT}; flat_node(mul(tR, T.trans), mul(T.inv, iR), PT_PRIM_TRIANGLE, 0, 0, 0, p, 2, 5); }
    { uint32_t p[] = {gR, gP}; flat_node(mul(tR, P.trans), mul(P.inv, iR), PT_PRIM_PLANE, 0, 0, 1, p, 2, 6); }
    { uint32_t p[] = {gR, gB, gB0}; flat_node(mul(tRB, B0.trans), mul(B0.inv, iRB), PT_PRIM_CUBE, 0, 0, 2, p, 3, 1); }
    { uint32_t p[] = {gR, gB, gB1}; flat_node(mul(tRB, B1.trans), mul(B1.inv, iRB), PT_PRIM_MESH, 0, 0, 3, p, 3, 2); }
    { uint32_t p[] = {gR, gC, gB, gB0}; flat_node(mul(tRCB, B0.trans), mul(B0.inv, iRCB), PT_PRIM_CUBE, 0, 0, 2, p, 4, 3); }
    { uint32_t p[] = {gR, gC, gB, gB1}; flat_node(mul(tRCB, B1.trans), mul(B1.inv, iRCB), PT_PRIM_MESH, 0, 0, 3, p, 4, 4); }
    /* materials by first use in that order: mat0 (sphere, triangle), mat3 = index 1 (plane), mat1 = index 2 (cube), mat2 = index 3 (mesh) */
    const double materials[4 * 10] = {
        0.75, 0.25, 0.125, 0.5, 0.5, 0.5, 32.0, 0.0, 0.0, 0.0,     /* sphere / triangle */
        0.125, 0.125, 0.125, 0.5, 0.5, 0.5, 64.0, 0.5, 0.0, 0.0,   /* mirror plane */
        0.25, 0.5, 0.75, 0.25, 0.25, 0.25, 16.0, 0.0, 0.0, 0.0,    /* cube */
        0.5, 0.75, 0.25, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,        /* mesh */
    };
    /* the tent mesh: 2 triangles over 4 positions (-1,0,-1) (1,0,-1) (0,2,1) (0,0,3); hip_pack emits it as a triangle
     * list: 6 vertices, indices 0..5 */
    const double pos4[4][3] = {{-1, 0, -1}, {1, 0, -1}, {0, 2, 1}, {0, 0, 3}};
    const int tri_idx[2][3] = {{0, 1, 2}, {1, 3, 2}};
    double mesh_positions[6 * 3], mesh_normals[6 * 3];
    uint32_t mesh_indices[6];
    for (int t = 0; t < 2; t++) for (int c = 0; c < 3; c++) {
        memcpy(mesh_positions + 3 * (3 * t + c), pos4[tri_idx[t][c]], 24);
        mesh_normals[3 * (3 * t + c)] = mesh_normals[3 * (3 * t + c) + 1] = mesh_normals[3 * (3 * t + c) + 2] = 0.0;
        mesh_indices[3 * t + c] = (uint32_t)(3 * t + c);
    }
    const uint64_t mesh_vert_off[2] = {0, 6}, mesh_tri_off[2] = {0, 2};
    const uint8_t mesh_has_normals[1] = {0};
    /* BoundingBox::new(min, max) of the mesh (bounding_box.rs:55-82): min (-1,0,-1), max (1,2,3): size (2,2,4), centre (0,1,1) */
    Xf bb = xf_translated(xf_scaled3(xf_new(), 2, 2, 4), 0, 1, 1);
    M4 mesh_bounds_inv = clean(bb.inv);
    const double tri_vertices[9] = {-4, -0.5, -3, -2, -0.5, -3, -3, 2.5, -3};
    const double lights[2 * 15] = {
        -4, 8, 6, 0.75, 0.75, 0.75, 1, 0, 0, 0, 0, 0, 0, 0, 0,
        6, 4, 8, 0.5, 0.25, 0.5, 1, 0, 0, 0, 0, 0, 0, 0, 0,
    };

    pt_scene s;
    memset(&s, 0, sizeof s);
    s.n_nodes = n_nodes; s.trans = g_trans; s.invtrans = g_inv; s.normal_trans = g_nrm;
    s.prim_type = prim_type; s.prim_data = prim_data; s.prim_flags = prim_flags; s.material = material;
    s.n_meshes = 1; s.mesh_vert_off = mesh_vert_off; s.mesh_tri_off = mesh_tri_off; s.mesh_positions = mesh_positions;
    s.mesh_normals = mesh_normals; s.mesh_has_normals = mesh_has_normals; s.mesh_indices = mesh_indices; s.mesh_bounds_invtrans = mesh_bounds_inv.m;
    s.n_triangles = 1; s.tri_vertices = tri_vertices; s.tri_normals = NULL;
    s.n_materials = 4; s.materials = materials; s.n_lights = 2; s.lights = lights;
    s.ambient[0] = s.ambient[1] = s.ambient[2] = 0.25;
    s.n_graph_nodes = n_graph; s.graph_trans = gr_trans; s.graph_invtrans = gr_inv; s.graph_normal_trans = gr_nrm;
    s.node_chain_off = chain_off; s.node_chain = chain; s.node_dfs_rank = dfs_rank;

    /* Camera::new (camera.rs:34-45): eye (0, 1, 16) looking down -z: look_at_rh is a pure translation, its inverse too */
    pt_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.eye[0] = 0; cam.eye[1] = 1; cam.eye[2] = 16;
    M4 v2w = translation(0, 1, 16);
    memcpy(cam.view_to_world, v2w.m, 128);
    cam.fov_factor = tan((32.0 * (3.14159265358979323846 / 180.0)) / 2.0); /* Radians::from_degrees(32.0): f64::to_radians */
    cam.aspect_ratio = (double)W / (double)H; cam.width = W; cam.height = H;

    /* render.rs:31-34: the background closure evaluated per INTEGER pixel: the scripts' sky gradient */
    double* bg = (double*)malloc(sizeof(double) * 3 * W * H);
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
        double v = (double)y / (double)H;
        double* o = bg + 3 * ((size_t)y * W + x);
        o[0] = 0.2 * (1.0 - v) + 0.0 * v; o[1] = 0.4 * (1.0 - v) + 0.0 * v; o[2] = 0.6 * (1.0 - v) + 1.0 * v;
    }
    uint8_t* rgb = (uint8_t*)calloc(3, (size_t)W * H);

    pt_render_params p;
    memset(&p, 0, sizeof p);
    p.width = W; p.height = H; p.slice.x0 = 0; p.slice.y0 = 0; p.slice.x1 = W - 1; p.slice.y1 = H - 1;
    p.samples = S; p.seed = 0; p.sample_mode = PT_SAMPLE_RNG; p.background_rows = 0; p.tile_rank = 0; p.tile_ranks = 1; p.collect_stats = 1;

    pt_context* ctx = NULL;
    pt_stats st;
    if (pt_abi_version() != PT_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (pt_context_create(0, &ctx) != PT_OK) { fprintf(stderr, "pt_context_create failed\n"); return 1; }
    if (pt_scene_upload(ctx, &s, PT_TRAVERSE_HIER, NULL) != PT_OK) { fprintf(stderr, "pt_scene_upload: %s\n", pt_last_error(ctx)); return 1; }
    if (pt_render(ctx, &cam, bg, &p, rgb, NULL, &st) != PT_OK) { fprintf(stderr, "pt_render: %s\n", pt_last_error(ctx)); return 1; }
    pt_context_destroy(ctx);

    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(rgb, 3, (size_t)W * H, f) != (size_t)W * H) { fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
    fclose(f);
    printf("primary %llu shadow %llu reflect %llu hits %llu\n", (unsigned long long)st.primary, (unsigned long long)st.shadow,
           (unsigned long long)st.reflect, (unsigned long long)st.hits);
    free(bg); free(rgb);
    return 0;
}
