"""Turns a test-DSL scene into the product host library's scene DESCRIPTION (ph_scene_desc): per
node the ordered builder calls, replayed by the C++ SceneNode API. Test infrastructure."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
from scene_dsl import Camera, Scene


def description(scene: Scene) -> dict:
    a, lin = O.arrays_from_dsl(scene)
    ops = "".join("".join(o for o, _ in node_ops) for node_ops in lin.ops)
    ops_off, args_off, args = [0], [0], []
    for node_ops in lin.ops:
        ops_off.append(ops_off[-1] + len(node_ops))
        for _, arg in node_ops:
            args.extend(arg)
        args_off.append(len(args))
    d = {k: a[k] for k in ("prim_type", "prim_data", "prim_flags", "material", "child_off", "children", "root", "mesh_vert_off", "mesh_tri_off",
                            "mesh_positions", "mesh_normals", "mesh_has_normals", "mesh_indices", "n_triangles", "tri_vertices", "tri_normals",
                            "tri_has_normals", "n_materials", "materials", "n_lights", "lights", "ambient", "mesh_texcoords", "mesh_has_texcoords",
                            "tri_texcoords", "tri_has_texcoords", "material_texture", "material_normal_map", "material_uv_trans", "n_textures",
                            "texture_size", "texture_offset", "texture_rgb")}
    d["ops"] = ops.encode()
    d["ops_off"] = np.array(ops_off, dtype=np.uint32)
    d["args_off"] = np.array(args_off, dtype=np.uint32)
    d["args"] = np.array(args + [0.0], dtype=np.float64)
    return d


def host_scene(scene: Scene):
    from portrayer_amd import host
    return host.Scene.from_description(description(scene))


def cam10(cam: Camera) -> np.ndarray:
    return np.array([*map(float, cam.eye), *map(float, cam.center), *map(float, cam.up), cam.fovy_radians], dtype=np.float64)
