//! Packs the crate's scene types into the plain arrays of include/portrayer_hip.h.
//!
//! Lives INSIDE the portrayer crate (it reads crate-private types: FlatScene, KDTreeNode, BoundingBox); the few
//! private fields it needs are reached through the `hip_*` accessors that shim/apply.py appends to the owning
//! modules (shim/overlay/*.rs). The array layouts are those of the C++ host layer of the MI355X repository
//! (portrayer_amd/host/portrayer.cpp, `Renderer::Renderer` and `pack_graph`), which the GPU parity tests exercise;
//! tests/shim_replay.c replays this file's layout - meshes as triangle lists, three vertices per triangle - through
//! libportrayer_hip.so and compares the image with the CPU oracle.
//!
//! NOT COMPILED in the environment this was written in (no rustc / cargo there).
#![allow(dead_code)]

use std::collections::HashMap;
use std::collections::VecDeque;
use std::sync::Arc;

use crate::bounding_box::BoundingBox;
use crate::flat_scene::FlatSceneNode;
use crate::hip_ffi::*;
use crate::kdtree::{KDLeaf, KDTreeNode, NodeBounds};
use crate::light::Light;
use crate::material::Material;
use crate::math::{Mat4, Rgb};
use crate::primitive::{MeshData, Primitive, Shading, Triangle};
use crate::scene::{HierScene, SceneNode};
use crate::texture::Texture;

fn push_mat4(dst: &mut Vec<f64>, m: Mat4) {
    dst.extend_from_slice(&m.into_row_array()); // the ABI is row-major, vek stores columns
}

/// A linearised k-d tree (pt_kdtree, or the kdm_* arrays of a KDMesh): pre-order, node 0 is the root.
#[derive(Default)]
pub struct LinearTree {
    pub axis: Vec<i32>, pub plane: Vec<f64>, pub front: Vec<i32>, pub back: Vec<i32>, pub first: Vec<i32>, pub count: Vec<i32>,
    pub items: Vec<i32>,
    pub root_min: [f64; 3], pub root_max: [f64; 3],
    pub depth: i32,
}

/// Pre-order walk of a KDTreeNode (kdtree/node.rs:13-25). `item_index` numbers a leaf entry (by the address of its
/// shared Arc<NodeBounds<T>>: a node sits in every leaf its box touches, leaf.rs:200-214).
fn linearise<T>(root: &KDTreeNode<T>, item_index: &mut dyn FnMut(&Arc<NodeBounds<T>>) -> i32) -> LinearTree {
    fn walk<T>(n: &KDTreeNode<T>, t: &mut LinearTree, level: i32, item_index: &mut dyn FnMut(&Arc<NodeBounds<T>>) -> i32) -> i32 {
        let me = t.axis.len() as i32;
        match n {
            KDTreeNode::Leaf(KDLeaf {nodes, ..}) => {
                t.axis.push(-1); t.plane.push(0.0); t.front.push(-1); t.back.push(-1);
                t.first.push(t.items.len() as i32); t.count.push(nodes.len() as i32);
                for nb in nodes { let i = item_index(nb); t.items.push(i); }
            },
            KDTreeNode::Split {sep_plane, front_nodes, back_nodes, ..} => {
                let axis = if sep_plane.normal.x != 0.0 { 0 } else if sep_plane.normal.y != 0.0 { 1 } else { 2 };
                t.axis.push(axis);
                t.plane.push(match axis { 0 => sep_plane.point.x, 1 => sep_plane.point.y, _ => sep_plane.point.z });
                t.front.push(-1); t.back.push(-1); t.first.push(0); t.count.push(0);
                t.depth = t.depth.max(level + 1);
                let f = walk(front_nodes, t, level + 1, item_index);
                let b = walk(back_nodes, t, level + 1, item_index);
                t.front[me as usize] = f; t.back[me as usize] = b;
            },
        }
        me
    }
    let mut t = LinearTree::default();
    walk(root, &mut t, 0, item_index);
    let b = root.hip_bounds();
    t.root_min = b.min().into_array(); t.root_max = b.max().into_array();
    t
}

impl LinearTree {
    pub fn as_abi(&self) -> PtKdTree {
        PtKdTree {
            n_nodes: self.axis.len() as u32,
            axis: self.axis.as_ptr(), plane: self.plane.as_ptr(), front: self.front.as_ptr(), back: self.back.as_ptr(),
            first: self.first.as_ptr(), count: self.count.as_ptr(),
            n_items: self.items.len() as u32, leaf_items: self.items.as_ptr(),
            root_min: self.root_min, root_max: self.root_max, max_depth: self.depth,
        }
    }
}

/// Everything pt_scene points to. Field names follow the header.
#[derive(Default)]
pub struct Packed {
    trans: Vec<f64>, invtrans: Vec<f64>, normal_trans: Vec<f64>,
    prim_type: Vec<i32>, prim_data: Vec<i32>, prim_flags: Vec<i32>, material: Vec<i32>,
    mesh_vert_off: Vec<u64>, mesh_tri_off: Vec<u64>, mesh_positions: Vec<f64>, mesh_normals: Vec<f64>, mesh_has_normals: Vec<u8>,
    mesh_indices: Vec<u32>, mesh_bounds_invtrans: Vec<f64>, mesh_texcoords: Vec<f64>, mesh_has_texcoords: Vec<u8>,
    tri_vertices: Vec<f64>, tri_normals: Vec<f64>, tri_texcoords: Vec<f64>, tri_has_texcoords: Vec<u8>, any_tri_normals: bool,
    materials: Vec<f64>, material_texture: Vec<i32>, material_normal_map: Vec<i32>, material_uv_trans: Vec<f64>,
    texture_size: Vec<u32>, texture_offset: Vec<u64>, texture_rgb: Vec<u8>,
    lights: Vec<f64>, ambient: [f64; 3],
    mesh_kd_root: Vec<i32>, mesh_kd_depth: Vec<i32>, mesh_kd_bounds: Vec<f64>, mesh_kd_bounds_invtrans: Vec<f64>,
    kdm: LinearTree,
    graph_trans: Vec<f64>, graph_invtrans: Vec<f64>, graph_normal_trans: Vec<f64>,
    node_chain_off: Vec<u32>, node_chain: Vec<u32>, node_dfs_rank: Vec<u32>,
    // identity maps while packing
    mat_id: HashMap<*const Material, i32>,
    mesh_id: HashMap<*const (), i32>,
    tex_id: HashMap<*const u8, i32>,
}

impl Packed {
    fn texture_index(&mut self, rgb: &[u8], width: u32, height: u32) -> i32 {
        let key = rgb.as_ptr();
        if let Some(&i) = self.tex_id.get(&key) { return i; }
        let i = (self.texture_size.len() / 2) as i32;
        self.texture_size.push(width); self.texture_size.push(height);
        self.texture_offset.push(self.texture_rgb.len() as u64);
        self.texture_rgb.extend_from_slice(rgb); // RgbImage::into_raw() order: row-major RGB8 (texture.rs:74-76)
        self.tex_id.insert(key, i);
        i
    }

    fn material_index(&mut self, m: &Arc<Material>) -> i32 {
        let key = Arc::as_ptr(m);
        if let Some(&i) = self.mat_id.get(&key) { return i; }
        let i = (self.materials.len() / 10) as i32;
        self.materials.extend_from_slice(&[m.diffuse.r, m.diffuse.g, m.diffuse.b, m.specular.r, m.specular.g, m.specular.b,
                                           m.shininess, m.reflectivity, m.glossy_side_length, m.refraction_index]);
        let tex = match m.texture.as_ref().map(|t| &**t) {
            Some(Texture::Image(img)) => { let (rgb, w, h) = img.hip_rgb(); self.texture_index(rgb, w, h) },
            Some(Texture::FnTex(_)) => panic!("function textures cannot run on the GPU: bake them into an image (texture.rs:25)"),
            None => -1,
        };
        let nmap = match m.normals.as_ref() {
            Some(n) => { let (rgb, w, h) = n.hip_rgb(); self.texture_index(rgb, w, h) },
            None => -1,
        };
        self.material_texture.push(tex); self.material_normal_map.push(nmap);
        self.material_uv_trans.extend_from_slice(&m.uv_trans.into_row_array());
        self.mat_id.insert(key, i);
        i
    }

    /// One mesh of the ABI from a list of triangles: vertices 3 per triangle, indices 0, 1, 2, ... (the reference
    /// rebuilds Triangle values from index triples on every test, mesh.rs:95-115; the values are what matters).
    fn push_mesh(&mut self, key: *const (), tris: &[Triangle], bounds: &BoundingBox) -> i32 {
        if let Some(&i) = self.mesh_id.get(&key) { return i; }
        let i = self.mesh_has_normals.len() as i32;
        if self.mesh_vert_off.is_empty() { self.mesh_vert_off.push(0); self.mesh_tri_off.push(0); }
        let v0 = *self.mesh_vert_off.last().unwrap();
        let smooth = tris.first().map_or(false, |t| t.normals.is_some());
        let textured = tris.first().map_or(false, |t| t.tex_coords.is_some());
        for (k, t) in tris.iter().enumerate() {
            for (c, p) in [t.a, t.b, t.c].iter().enumerate() {
                self.mesh_positions.extend_from_slice(&[p.x, p.y, p.z]);
                let n = t.normals.map(|(a, b, cc)| [a, b, cc][c]);
                self.mesh_normals.extend_from_slice(&n.map_or([0.0; 3], |n| [n.x, n.y, n.z]));
                let uv = t.tex_coords.map(|(a, b, cc)| [a, b, cc][c]);
                self.mesh_texcoords.extend_from_slice(&uv.map_or([0.0; 2], |uv| [uv.u, uv.v]));
            }
            let base = 3 * k as u32;
            self.mesh_indices.extend_from_slice(&[base, base + 1, base + 2]);
        }
        self.mesh_has_normals.push(smooth as u8); self.mesh_has_texcoords.push(textured as u8);
        self.mesh_vert_off.push(v0 + 3 * tris.len() as u64);
        let t0 = *self.mesh_tri_off.last().unwrap();
        self.mesh_tri_off.push(t0 + tris.len() as u64);
        push_mat4(&mut self.mesh_bounds_invtrans, bounds.hip_invtrans()); // BoundingBox::new, bounding_box.rs:55-82
        self.mesh_kd_root.push(-1); self.mesh_kd_depth.push(0);
        self.mesh_kd_bounds.extend_from_slice(&[0.0; 6]);
        self.mesh_kd_bounds_invtrans.extend_from_slice(&[0.0; 16]);
        self.mesh_id.insert(key, i);
        i
    }

    /// flat_scene.rs:50-61 for one flattened node (in the order the caller numbers them).
    fn push_node(&mut self, node: &FlatSceneNode) {
        push_mat4(&mut self.trans, node.trans());
        push_mat4(&mut self.invtrans, node.inverse_trans());
        push_mat4(&mut self.normal_trans, node.normal_trans());
        let geometry = node.geometry();
        let (ty, data, flags) = match &geometry.primitive {
            Primitive::Sphere(_) => (PT_PRIM_SPHERE, 0, 0),
            Primitive::Plane(_) => (PT_PRIM_PLANE, 0, 0),
            Primitive::Cube(_) => (PT_PRIM_CUBE, 0, 0),
            Primitive::Cylinder(_) => (PT_PRIM_CYLINDER, 0, 0),
            Primitive::Cone(_) => (PT_PRIM_CONE, 0, 0),
            Primitive::Triangle(t) => {
                let i = (self.tri_vertices.len() / 9) as i32;
                for p in &[t.a, t.b, t.c] { self.tri_vertices.extend_from_slice(&[p.x, p.y, p.z]); }
                let n = t.normals.map_or([[0.0; 3]; 3], |(a, b, c)| [[a.x, a.y, a.z], [b.x, b.y, b.z], [c.x, c.y, c.z]]);
                for v in &n { self.tri_normals.extend_from_slice(v); }
                let uv = t.tex_coords.map_or([[0.0; 2]; 3], |(a, b, c)| [[a.u, a.v], [b.u, b.v], [c.u, c.v]]);
                for v in &uv { self.tri_texcoords.extend_from_slice(v); }
                self.tri_has_texcoords.push(t.tex_coords.is_some() as u8);
                self.any_tri_normals |= t.normals.is_some();
                (PT_PRIM_TRIANGLE, i, t.normals.is_some() as i32)
            },
            Primitive::Mesh(m) => {
                let data: &Arc<MeshData> = m.hip_data();
                let smooth = match m.hip_shading() { Shading::Smooth => true, Shading::Flat => false };
                let tris: Vec<Triangle> = data.triangles(m.hip_shading()).collect();
                // one ABI mesh per (MeshData, shading): the vertex normals travel with the triangles
                let key = (Arc::as_ptr(data) as usize + smooth as usize) as *const ();
                let i = self.push_mesh(key, &tris, data.hip_bounds());
                (PT_PRIM_MESH, i, smooth as i32)
            },
            Primitive::KDMesh(m) => {
                // KDMesh keeps no MeshData, only its tree of triangles (kdmesh.rs:19-24): the mesh of the ABI is the list of
                // the tree's distinct triangles in pre-order of first appearance, and kdm_items index into that list
                let tree = m.hip_tree();
                let key = tree as *const _ as *const ();
                if let Some(&i) = self.mesh_id.get(&key) {
                    (PT_PRIM_KDMESH, i, (self.mesh_has_normals[i as usize] != 0) as i32)
                } else {
                    let mut tris: Vec<Triangle> = Vec::new();
                    let mut seen: HashMap<*const NodeBounds<Triangle>, i32> = HashMap::new();
                    let lin = linearise(tree, &mut |nb| *seen.entry(Arc::as_ptr(nb)).or_insert_with(|| { tris.push(nb.node.clone()); tris.len() as i32 - 1 }));
                    let i = self.push_mesh(key, &tris, tree.hip_bounds()) as usize;
                    // append this tree to the shared kdm_* arrays, node and item indices shifted
                    let (nb, ib) = (self.kdm.axis.len() as i32, self.kdm.items.len() as i32);
                    for k in 0..lin.axis.len() {
                        let split = lin.axis[k] >= 0;
                        self.kdm.axis.push(lin.axis[k]); self.kdm.plane.push(lin.plane[k]);
                        self.kdm.front.push(if split { lin.front[k] + nb } else { -1 }); self.kdm.back.push(if split { lin.back[k] + nb } else { -1 });
                        self.kdm.first.push(if split { 0 } else { lin.first[k] + ib }); self.kdm.count.push(lin.count[k]);
                    }
                    self.kdm.items.extend_from_slice(&lin.items);
                    self.mesh_kd_root[i] = nb; self.mesh_kd_depth[i] = lin.depth;
                    self.mesh_kd_bounds[6 * i..6 * i + 3].copy_from_slice(&lin.root_min);
                    self.mesh_kd_bounds[6 * i + 3..6 * i + 6].copy_from_slice(&lin.root_max);
                    self.mesh_kd_bounds_invtrans[16 * i..16 * i + 16].copy_from_slice(&tree.hip_bounds().hip_invtrans().into_row_array());
                    (PT_PRIM_KDMESH, i as i32, (self.mesh_has_normals[i] != 0) as i32)
                }
            },
        };
        self.prim_type.push(ty); self.prim_data.push(data); self.prim_flags.push(flags);
        let mat = self.material_index(&geometry.material);
        self.material.push(mat);
    }

    fn push_lights(&mut self, lights: &[Light], ambient: Rgb) {
        for l in lights {
            self.lights.extend_from_slice(&[l.position.x, l.position.y, l.position.z, l.color.r, l.color.g, l.color.b,
                                            l.falloff.c0, l.falloff.c1, l.falloff.c2,
                                            l.area.a.x, l.area.a.y, l.area.a.z, l.area.b.x, l.area.b.y, l.area.b.z]);
        }
        self.ambient = [ambient.r, ambient.g, ambient.b];
    }

    pub fn as_abi(&self) -> PtScene {
        let opt = |v: &Vec<f64>| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        let textured = !self.texture_size.is_empty();
        PtScene {
            n_nodes: self.prim_type.len() as u32,
            trans: self.trans.as_ptr(), invtrans: self.invtrans.as_ptr(), normal_trans: self.normal_trans.as_ptr(),
            prim_type: self.prim_type.as_ptr(), prim_data: self.prim_data.as_ptr(), prim_flags: self.prim_flags.as_ptr(), material: self.material.as_ptr(),
            n_meshes: self.mesh_has_normals.len() as u32,
            mesh_vert_off: self.mesh_vert_off.as_ptr(), mesh_tri_off: self.mesh_tri_off.as_ptr(),
            mesh_positions: self.mesh_positions.as_ptr(), mesh_normals: self.mesh_normals.as_ptr(), mesh_has_normals: self.mesh_has_normals.as_ptr(),
            mesh_indices: self.mesh_indices.as_ptr(), mesh_bounds_invtrans: self.mesh_bounds_invtrans.as_ptr(),
            n_triangles: (self.tri_vertices.len() / 9) as u32, tri_vertices: self.tri_vertices.as_ptr(),
            tri_normals: if self.any_tri_normals { self.tri_normals.as_ptr() } else { std::ptr::null() },
            n_materials: (self.materials.len() / 10) as u32, materials: self.materials.as_ptr(),
            n_lights: (self.lights.len() / 15) as u32, lights: self.lights.as_ptr(),
            ambient: self.ambient,
            mesh_texcoords: if textured { self.mesh_texcoords.as_ptr() } else { std::ptr::null() },
            mesh_has_texcoords: if textured { self.mesh_has_texcoords.as_ptr() } else { std::ptr::null() },
            tri_texcoords: if textured { self.tri_texcoords.as_ptr() } else { std::ptr::null() },
            tri_has_texcoords: if textured { self.tri_has_texcoords.as_ptr() } else { std::ptr::null() },
            material_texture: if textured { self.material_texture.as_ptr() } else { std::ptr::null() },
            material_normal_map: if textured { self.material_normal_map.as_ptr() } else { std::ptr::null() },
            material_uv_trans: if textured { self.material_uv_trans.as_ptr() } else { std::ptr::null() },
            n_textures: (self.texture_size.len() / 2) as u32,
            texture_size: self.texture_size.as_ptr(), texture_offset: self.texture_offset.as_ptr(), texture_rgb: self.texture_rgb.as_ptr(),
            mesh_kd_root: if self.kdm.axis.is_empty() { std::ptr::null() } else { self.mesh_kd_root.as_ptr() },
            mesh_kd_depth: self.mesh_kd_depth.as_ptr(), mesh_kd_bounds: self.mesh_kd_bounds.as_ptr(), mesh_kd_bounds_invtrans: self.mesh_kd_bounds_invtrans.as_ptr(),
            n_kdm_nodes: self.kdm.axis.len() as u32,
            kdm_axis: self.kdm.axis.as_ptr(), kdm_plane: self.kdm.plane.as_ptr(), kdm_front: self.kdm.front.as_ptr(), kdm_back: self.kdm.back.as_ptr(),
            kdm_first: self.kdm.first.as_ptr(), kdm_count: self.kdm.count.as_ptr(),
            n_kdm_items: self.kdm.items.len() as u32, kdm_items: self.kdm.items.as_ptr(),
            n_graph_nodes: (self.graph_trans.len() / 16) as u32,
            graph_trans: opt(&self.graph_trans), graph_invtrans: opt(&self.graph_invtrans), graph_normal_trans: opt(&self.graph_normal_trans),
            node_chain_off: if self.node_chain_off.is_empty() { std::ptr::null() } else { self.node_chain_off.as_ptr() },
            node_chain: self.node_chain.as_ptr(), node_dfs_rank: self.node_dfs_rank.as_ptr(),
        }
    }
}

/// `--features flat_scene` and the default (hierarchical) feature set: the flattened nodes in FlatScene::from's
/// breadth-first order (flat_scene.rs:18-46).
pub fn pack_flat(nodes: &[FlatSceneNode], lights: &[Light], ambient: Rgb) -> Packed {
    let mut p = Packed::default();
    for n in nodes { p.push_node(n); }
    p.push_lights(lights, ambient);
    p
}

/// The default feature set (SceneNode::ray_cast on the hierarchy, scene.rs:80-120): every SceneNode's OWN matrices and
/// each flattened node's path through them. Walks the hierarchy in the same breadth-first order as FlatScene::from,
/// so entry i describes flattened node i; `dfs_rank` = the order in which the recursive fold meets the nodes (a node
/// before its children), which decides exact ties in t (ray.rs:87-99).
pub fn pack_graph(scene: &HierScene, p: &mut Packed) {
    let mut graph_id: HashMap<*const SceneNode, u32> = HashMap::new();
    let mut paths: Vec<Vec<u32>> = Vec::new(); // child indices from the root, per flattened node
    let mut remaining: VecDeque<(Vec<u32>, Vec<u32>, Arc<SceneNode>)> = VecDeque::new(); // (chain of graph ids above, child-index path, node)
    remaining.push_back((Vec::new(), Vec::new(), scene.root.clone()));
    p.node_chain_off.push(0);
    while let Some((above, path, node)) = remaining.pop_front() {
        let next = graph_id.len() as u32;
        let id = *graph_id.entry(Arc::as_ptr(&node)).or_insert_with(|| {
            push_mat4(&mut p.graph_trans, node.trans());
            push_mat4(&mut p.graph_invtrans, node.inverse_trans());
            push_mat4(&mut p.graph_normal_trans, node.normal_trans());
            next
        });
        let mut chain = above.clone();
        chain.push(id);
        if node.geometry().is_some() {
            p.node_chain.extend_from_slice(&chain);
            p.node_chain_off.push(p.node_chain.len() as u32);
            paths.push(path.clone());
        }
        for (k, child) in node.children().iter().enumerate() {
            let mut child_path = path.clone();
            child_path.push(k as u32);
            remaining.push_back((chain.clone(), child_path, child.clone()));
        }
    }
    // depth-first order with a node before its children = lexicographic order of the child-index paths, a prefix first
    let mut order: Vec<usize> = (0..paths.len()).collect();
    order.sort_by(|&a, &b| paths[a].cmp(&paths[b]));
    p.node_dfs_rank = vec![0; paths.len()];
    for (rank, &i) in order.iter().enumerate() { p.node_dfs_rank[i] = rank as u32; }
}

/// `--features kdtree`: the scene tree KDTreeScene::from built (kdscene.rs:19-43). Its leaves own the FlatSceneNodes, so
/// the flattened nodes are numbered in pre-order of first appearance in the tree (the k-d walk never compares node
/// indices: within a leaf the order of the leaf's Vec decides, ray.rs:87-99).
pub fn pack_kd(root: &KDTreeNode<FlatSceneNode>, lights: &[Light], ambient: Rgb) -> (Packed, LinearTree) {
    let mut p = Packed::default();
    let mut seen: HashMap<*const NodeBounds<FlatSceneNode>, i32> = HashMap::new();
    let tree = linearise(root, &mut |nb| {
        let next = seen.len() as i32;
        *seen.entry(Arc::as_ptr(nb)).or_insert_with(|| { p.push_node(&nb.node); next })
    });
    p.push_lights(lights, ambient);
    (p, tree)
}
