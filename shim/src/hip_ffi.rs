//! Declarations of the C ABI in include/portrayer_hip.h (PT_ABI_VERSION 8), field for field.
//! tests/test_integration_doc.py of the MI355X repository checks these structs against the header.
#![allow(dead_code)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct PtContext { _private: [u8; 0] }
#[repr(C)] pub struct PtNode { _private: [u8; 0] }

pub const PT_ABI_VERSION: c_int = 8;

// enum Primitive, src/primitive.rs:67-81
pub const PT_PRIM_SPHERE: i32 = 0;
pub const PT_PRIM_TRIANGLE: i32 = 1;
pub const PT_PRIM_MESH: i32 = 2;
pub const PT_PRIM_KDMESH: i32 = 3;
pub const PT_PRIM_PLANE: i32 = 4;
pub const PT_PRIM_CUBE: i32 = 5;
pub const PT_PRIM_CYLINDER: i32 = 6;
pub const PT_PRIM_CONE: i32 = 7;

// cargo features flat_scene / kdtree / neither, src/render.rs:121-126
pub const PT_TRAVERSE_FLAT: c_int = 1;
pub const PT_TRAVERSE_KD: c_int = 2;
pub const PT_TRAVERSE_HIER: c_int = 3;

pub const PT_SAMPLE_CENTRE: i32 = 0;
pub const PT_SAMPLE_RNG: i32 = 1;

pub const PT_OK: c_int = 0;
pub const PT_ERR_SLICE: c_int = -4;
pub const PT_ERR_SCENE: c_int = -5;

#[repr(C)]
pub struct PtScene {                       // pt_scene
    pub n_nodes: u32,
    pub trans: *const f64, pub invtrans: *const f64, pub normal_trans: *const f64,   // n x 16, row-major
    pub prim_type: *const i32, pub prim_data: *const i32, pub prim_flags: *const i32, pub material: *const i32,
    pub n_meshes: u32,
    pub mesh_vert_off: *const u64, pub mesh_tri_off: *const u64,
    pub mesh_positions: *const f64, pub mesh_normals: *const f64, pub mesh_has_normals: *const u8,
    pub mesh_indices: *const u32, pub mesh_bounds_invtrans: *const f64,
    pub n_triangles: u32, pub tri_vertices: *const f64, pub tri_normals: *const f64,
    pub n_materials: u32, pub materials: *const f64,      // x 10
    pub n_lights: u32, pub lights: *const f64,            // x 15
    pub ambient: [f64; 3],
    // ABI 2: textures (src/texture.rs); null / 0 when the scene has none
    pub mesh_texcoords: *const f64, pub mesh_has_texcoords: *const u8,
    pub tri_texcoords: *const f64, pub tri_has_texcoords: *const u8,
    pub material_texture: *const i32, pub material_normal_map: *const i32, pub material_uv_trans: *const f64,
    pub n_textures: u32, pub texture_size: *const u32, pub texture_offset: *const u64, pub texture_rgb: *const u8,
    // ABI 3: KDMesh triangle trees (src/kdtree/kdmesh.rs), linearised like PtKdTree; null when unused
    pub mesh_kd_root: *const i32, pub mesh_kd_depth: *const i32, pub mesh_kd_bounds: *const f64, pub mesh_kd_bounds_invtrans: *const f64,
    pub n_kdm_nodes: u32, pub kdm_axis: *const i32, pub kdm_plane: *const f64, pub kdm_front: *const i32, pub kdm_back: *const i32,
    pub kdm_first: *const i32, pub kdm_count: *const i32, pub n_kdm_items: u32, pub kdm_items: *const i32,
    // ABI 4: the scene graph for PT_TRAVERSE_HIER = the crate's DEFAULT traversal (src/scene.rs:80-120); null / 0 otherwise
    pub n_graph_nodes: u32, pub graph_trans: *const f64, pub graph_invtrans: *const f64, pub graph_normal_trans: *const f64,
    pub node_chain_off: *const u32, pub node_chain: *const u32, pub node_dfs_rank: *const u32,
}

#[repr(C)]
pub struct PtKdTree {                      // pt_kdtree
    pub n_nodes: u32,
    pub axis: *const i32, pub plane: *const f64, pub front: *const i32, pub back: *const i32,
    pub first: *const i32, pub count: *const i32,
    pub n_items: u32, pub leaf_items: *const i32,
    pub root_min: [f64; 3], pub root_max: [f64; 3], pub max_depth: i32,
}

#[repr(C)] pub struct PtCamera { pub eye: [f64; 3], pub view_to_world: [f64; 16], pub fov_factor: f64, pub aspect_ratio: f64, pub width: f64, pub height: f64 }
#[repr(C)] pub struct PtRect { pub x0: u32, pub y0: u32, pub x1: u32, pub y1: u32 }

#[repr(C)]
pub struct PtRenderParams {
    pub width: u32, pub height: u32, pub slice: PtRect, pub samples: u32, pub seed: u64,
    pub sample_mode: i32, pub background_rows: i32, pub tile_rank: u32, pub tile_ranks: u32, pub collect_stats: i32,
}

#[repr(C)] #[derive(Default)]
pub struct PtStats {
    pub primary: u64, pub shadow: u64, pub reflect: u64, pub refract: u64, pub depth11_skipped: u64, pub hits: u64,
    pub n_inner: u64, pub n_leaf: u64, pub n_analytic: u64, pub n_tri: u64, pub n_bbox: u64,
    pub kd_plane_miss: u64, pub stack_overflow: u64, pub kernel_ms: f64, pub total_ms: f64,
    pub diag: [u64; 8],
    pub kernel_mode: u32, pub kernel_variant: u32,
}

extern "C" {
    pub fn pt_abi_version() -> c_int;
    pub fn pt_device_count() -> c_int;
    pub fn pt_context_create(device: c_int, out: *mut *mut PtContext) -> c_int;
    pub fn pt_context_destroy(ctx: *mut PtContext);
    pub fn pt_last_error(ctx: *const PtContext) -> *const c_char;
    pub fn pt_scene_upload(ctx: *mut PtContext, scene: *const PtScene, traverse: c_int, kd: *const PtKdTree) -> c_int;
    pub fn pt_render(ctx: *mut PtContext, camera: *const PtCamera, background: *const f64, params: *const PtRenderParams,
                     rgb: *mut u8, linear: *mut f64, stats: *mut PtStats) -> c_int;
    pub fn pt_render_device(ctx: *mut PtContext, camera: *const PtCamera, d_background: *const f64, params: *const PtRenderParams,
                            compact: c_int, d_rgb: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn pt_render_finish(ctx: *mut PtContext, stats: *mut PtStats) -> c_int;
    pub fn pt_context_stream(ctx: *mut PtContext, slot: c_int) -> *mut c_void;
    pub fn pt_context_next_slot(ctx: *const PtContext) -> c_int;
    // one render call over the GPUs of a node (one context per GPU, one RCCL gather)
    pub fn pt_node_create(n_devices: c_int, devices: *const c_int, out: *mut *mut PtNode) -> c_int;
    pub fn pt_node_destroy(node: *mut PtNode);
    pub fn pt_node_last_error(node: *const PtNode) -> *const c_char;
    pub fn pt_node_scene_upload(node: *mut PtNode, scene: *const PtScene, traverse: c_int, kd: *const PtKdTree) -> c_int;
    pub fn pt_node_render(node: *mut PtNode, camera: *const PtCamera, background: *const f64, params: *const PtRenderParams,
                          rgb: *mut u8, stats: *mut PtStats) -> c_int;
}
