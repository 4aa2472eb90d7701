"""Feeds the device library (include/portrayer_hip.h) straight from a test-DSL scene, bypassing the
product's C++ host layer: flattening, inverses, boxes and the k-d tree come from the ORACLE, so a
mismatch seen through this path is a kernel bug, not a host bug. Test infrastructure."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

import oracle_lib as O
from portrayer_amd import _hip as H
from scene_dsl import Camera, default_background

_dp, _ip, _up, _u64p, _u8p = H._dp, H._ip, H._up, H._u64p, H._u8p


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def bbox_invtrans(mn, mx):
    """BoundingBox::new (bounding_box.rs:55-82) through the oracle's matrix code."""
    size = np.maximum(np.asarray(mx) - np.asarray(mn), 0.00001)
    center = (np.asarray(mn) + np.asarray(mx)) / 2.0
    trans = O.compose([("s", tuple(size)), ("t", tuple(center))])
    return O.mat4_inverse(trans)


class DeviceScene:
    def __init__(self, scene, traverse=H.TRAVERSE_FLAT, kd_depth=10):
        self.ps = O.pack(scene)
        a = self.ps.arrays
        flat = O.flatten(self.ps)
        self.flat = flat
        self.keep = k = {}
        k["trans"] = np.ascontiguousarray(flat["trans"].reshape(-1, 16))
        k["inv"] = np.ascontiguousarray(flat["invtrans"].reshape(-1, 16))
        k["nrm"] = np.ascontiguousarray(flat["normal_trans"].reshape(-1, 16))
        k["type"], k["data"], k["flags"], k["mat"] = flat["prim_type"], flat["prim_data"], flat["prim_flags"].copy(), flat["material"]
        lin = self.ps.lin
        # stand-alone triangles with normals are "smooth" for the device (triangle.rs:82-86)
        for i, t in enumerate(k["type"]):
            if t == 1 and lin.triangles[k["data"][i]].tri_normals is not None:
                k["flags"][i] |= 1
        nm = len(lin.meshes)
        k["mbi"] = np.zeros((max(nm, 1), 16))
        for i, m in enumerate(lin.meshes):
            k["mbi"][i] = bbox_invtrans(m.positions.min(axis=0), m.positions.max(axis=0)).reshape(16)
        s = H.PtScene()
        s.n_nodes = len(k["type"])
        s.trans, s.invtrans, s.normal_trans = _p(k["trans"], _dp), _p(k["inv"], _dp), _p(k["nrm"], _dp)
        s.prim_type, s.prim_data, s.prim_flags, s.material = _p(k["type"], _ip), _p(k["data"], _ip), _p(k["flags"], _ip), _p(k["mat"], _ip)
        s.n_meshes = nm
        s.mesh_vert_off, s.mesh_tri_off = _p(a["mesh_vert_off"], _u64p), _p(a["mesh_tri_off"], _u64p)
        s.mesh_positions, s.mesh_normals = _p(a["mesh_positions"], _dp), _p(a["mesh_normals"], _dp)
        s.mesh_has_normals, s.mesh_indices = _p(a["mesh_has_normals"], _u8p), _p(a["mesh_indices"], _up)
        s.mesh_bounds_invtrans = _p(k["mbi"], _dp)
        s.n_triangles = len(lin.triangles)
        s.tri_vertices, s.tri_normals = _p(a["tri_vertices"], _dp), _p(a["tri_normals"], _dp)
        s.n_materials, s.materials = len(lin.materials), _p(a["materials"], _dp)
        s.n_lights, s.lights = self.ps.struct.n_lights, _p(a["lights"], _dp)
        s.ambient = self.ps.struct.ambient
        self.struct = s
        self.traverse = traverse
        self.kd = None
        if traverse == H.TRAVERSE_KD:
            tree = O.kd_scene_dump(self.ps, kd_depth)
            self.tree = tree
            kd = H.PtKdTree()
            kd.n_nodes = len(tree["kind"])
            k["kd_axis"] = np.ascontiguousarray(tree["axis"]); k["kd_plane"] = np.ascontiguousarray(tree["plane"])
            k["kd_front"] = np.ascontiguousarray(tree["front"]); k["kd_back"] = np.ascontiguousarray(tree["back"])
            k["kd_first"] = np.ascontiguousarray(tree["first"]); k["kd_count"] = np.ascontiguousarray(tree["count"])
            k["kd_items"] = np.ascontiguousarray(np.concatenate([tree["items"], np.zeros(1, dtype=np.int32)]))
            kd.axis, kd.plane, kd.front, kd.back = _p(k["kd_axis"], _ip), _p(k["kd_plane"], _dp), _p(k["kd_front"], _ip), _p(k["kd_back"], _ip)
            kd.first, kd.count = _p(k["kd_first"], _ip), _p(k["kd_count"], _ip)
            kd.n_items, kd.leaf_items = len(tree["items"]), _p(k["kd_items"], _ip)
            kd.root_min = (C.c_double * 3)(*tree["root_bounds"][:3]); kd.root_max = (C.c_double * 3)(*tree["root_bounds"][3:])
            kd.max_depth = kd_depth
            self.kd = kd

    def upload(self, ctx: H.Context):
        ctx.upload(self.struct, self.traverse, self.kd)


def _sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def _dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
def _cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def _norm(a):
    m = math.sqrt(_dot(a, a))
    return (a[0] / m, a[1] / m, a[2] / m)


def camera_struct(cam: Camera, width: int, height: int) -> H.PtCamera:
    """Camera::new (camera.rs:34-45), same operation order as the oracle's camera_new."""
    eye, center, up = tuple(map(float, cam.eye)), tuple(map(float, cam.center)), tuple(map(float, cam.up))
    f = _norm(_sub(center, eye)); s = _norm(_cross(f, up)); u = _cross(s, f)
    v = np.array([[s[0], s[1], s[2], -_dot(s, eye)], [u[0], u[1], u[2], -_dot(u, eye)],
                  [-f[0], -f[1], -f[2], _dot(f, eye)], [0.0, 0.0, 0.0, 1.0]])
    v2w = O.mat4_inverse(v)
    c = H.PtCamera()
    c.eye = (C.c_double * 3)(*eye)
    c.view_to_world = (C.c_double * 16)(*v2w.reshape(16))
    c.fov_factor = math.tan(cam.fovy_radians / 2.0)
    c.aspect_ratio = float(width) / float(height)
    c.width, c.height = float(width), float(height)
    return c


def render(ctx: H.Context, cam: Camera, width: int, height: int, samples=1, seed=0, sample_mode=H.SAMPLE_CENTRE, rect=None,
           background=None, stats=False, tile_rank=0, tile_ranks=1, into=None):
    bg = default_background(width, height) if background is None else np.ascontiguousarray(background, dtype=np.float64)
    rows = 1 if bg.shape == (height, 3) else 0
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, width - 1, height - 1)
    p = H.PtRenderParams(width, height, H.PtRect(x0, y0, x1, y1), samples, seed, sample_mode, rows, tile_rank, tile_ranks, 1 if stats else 0)
    rgb = into if into is not None else np.zeros((height, width, 3), dtype=np.uint8)
    linear = np.zeros((height, width, 3), dtype=np.float64)
    st = ctx.render(camera_struct(cam, width, height), bg, p, rgb, linear)
    return rgb, linear, st
