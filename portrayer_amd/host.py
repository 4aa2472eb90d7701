"""ctypes binding of libportrayer_host.so (include/portrayer_host.h): the C++ host layer that mirrors
the portrayer crate's API above the pixel loop (scene graph, flattening, k-d build, camera,
Image::render) and drives the gfx950 kernels through the C ABI. No CPU fallback exists."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _hip as H

LIB_PATH = os.path.join(H.PKG_DIR, "libportrayer_host.so")
REPO_ROOT = os.path.dirname(H.PKG_DIR)
DEFAULT_ASSETS = os.path.join(REPO_ROOT, "tests", "golden", "assets")

_dp, _ip, _up, _u64p, _u8p = H._dp, H._ip, H._up, H._u64p, H._u8p

EXPORTS = ["ph_last_error", "ph_scene_create", "ph_example_scene", "ph_scene_destroy", "ph_scene_counts", "ph_scene_export", "ph_scene_export_textures",
           "ph_scene_flatten", "ph_scene_kdtree", "ph_camera", "ph_obj_load", "ph_renderer_create", "ph_renderer_destroy",
           "ph_renderer_context", "ph_renderer_ranks", "ph_renderer_node", "ph_renderer_prepare_ms", "ph_renderer_render", "ph_example_render_to_png", "ph_png_read", "ph_png_write", "ph_image_read", "ph_scene_graph"]


class PortrayerHostError(RuntimeError):
    pass


class PortrayerPanic(PortrayerHostError):
    """Raised where the reference would panic (e.g. ImageSliceMut::new, render.rs:79-90)."""


class PhSceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("ops", C.c_char_p), ("ops_off", _up), ("args", _dp), ("args_off", _up),
        ("prim_type", _ip), ("prim_data", _ip), ("prim_flags", _ip), ("material", _ip), ("child_off", _up), ("children", _up),
        ("root", C.c_uint32),
        ("n_meshes", C.c_uint32), ("mesh_vert_off", _u64p), ("mesh_tri_off", _u64p), ("mesh_positions", _dp), ("mesh_normals", _dp),
        ("mesh_has_normals", _u8p), ("mesh_indices", _up),
        ("n_triangles", C.c_uint32), ("tri_vertices", _dp), ("tri_normals", _dp), ("tri_has_normals", _u8p),
        ("n_materials", C.c_uint32), ("materials", _dp), ("n_lights", C.c_uint32), ("lights", _dp), ("ambient", C.c_double * 3),
        ("mesh_texcoords", _dp), ("mesh_has_texcoords", _u8p), ("tri_texcoords", _dp), ("tri_has_texcoords", _u8p),
        ("material_texture", _ip), ("material_normal_map", _ip), ("material_uv_trans", _dp),
        ("n_textures", C.c_uint32), ("texture_size", _up), ("texture_offset", _u64p), ("texture_rgb", _u8p),
    ]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        H.lib()
        if not os.path.exists(LIB_PATH):
            raise PortrayerHostError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build())")
        l = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        l.ph_last_error.restype = C.c_char_p
        l.ph_scene_create.restype = C.c_int; l.ph_scene_create.argtypes = [C.POINTER(PhSceneDesc), C.POINTER(vp)]
        l.ph_example_scene.restype = C.c_int; l.ph_example_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp), _dp, _up]
        l.ph_scene_destroy.restype = None; l.ph_scene_destroy.argtypes = [vp]
        l.ph_scene_counts.restype = C.c_int; l.ph_scene_counts.argtypes = [vp, _u64p]
        l.ph_scene_export.restype = C.c_int
        l.ph_scene_export.argtypes = [vp, _dp, _ip, _ip, _ip, _ip, _up, _up, _up, _u64p, _u64p, _dp, _dp, _u8p, _up, _dp, _dp, _u8p, _dp, _dp, _dp]
        l.ph_scene_flatten.restype = C.c_int; l.ph_scene_flatten.argtypes = [vp, C.c_uint32, _dp, _dp, _dp, _ip, _ip, _dp]
        l.ph_scene_kdtree.restype = C.c_int
        l.ph_scene_kdtree.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, _ip, _dp, _ip, _ip, _ip, _ip, _ip, _up, _dp, _ip]
        l.ph_camera.restype = C.c_int; l.ph_camera.argtypes = [_dp, C.c_double, C.c_double, C.POINTER(H.PtCamera)]
        l.ph_obj_load.restype = C.c_int; l.ph_obj_load.argtypes = [C.c_char_p, _u64p, _dp, _dp, _up, C.c_uint64, C.c_uint64]
        l.ph_renderer_create.restype = C.c_int; l.ph_renderer_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        l.ph_renderer_destroy.restype = None; l.ph_renderer_destroy.argtypes = [vp]
        l.ph_renderer_context.restype = vp; l.ph_renderer_context.argtypes = [vp]
        l.ph_renderer_ranks.restype = C.c_int; l.ph_renderer_ranks.argtypes = [vp]
        l.ph_renderer_node.restype = vp; l.ph_renderer_node.argtypes = [vp]
        l.ph_renderer_prepare_ms.restype = C.c_int; l.ph_renderer_prepare_ms.argtypes = [vp, _dp]
        l.ph_renderer_render.restype = C.c_int
        l.ph_renderer_render.argtypes = [vp, _dp, C.POINTER(H.PtRenderParams), _dp, _u8p, _dp, C.POINTER(H.PtStats)]
        l.ph_scene_export_textures.restype = C.c_int
        l.ph_scene_export_textures.argtypes = [vp, _u64p, _ip, _ip, _dp, _up, _u64p, _u8p, _dp, _u8p, _dp, _u8p]
        l.ph_example_render_to_png.restype = C.c_int
        l.ph_example_render_to_png.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p]
        l.ph_png_read.restype = C.c_int; l.ph_png_read.argtypes = [C.c_char_p, _up, _u8p, C.c_uint64]
        l.ph_scene_graph.restype = C.c_int
        l.ph_scene_graph.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, _up, _up, _up, _dp, _dp, _dp, _up]
        l.ph_image_read.restype = C.c_int; l.ph_image_read.argtypes = [C.c_char_p, _up, _u8p, C.c_uint64]
        l.ph_png_write.restype = C.c_int; l.ph_png_write.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _u8p]
        _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc < 0:
        msg = lib().ph_last_error().decode()
        raise (PortrayerPanic if rc == -2 else PortrayerHostError)(f"{what} failed with {rc}: {msg}")
    return rc


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class Scene:
    """A scene::HierScene owned by the C++ host library."""

    def __init__(self, handle, camera=None, size=None):
        self._h = handle
        self.camera = camera  # 10 doubles: eye, center, up, fovy (radians), for example scenes
        self.size = size

    @staticmethod
    def from_description(d: dict) -> "Scene":
        """d: arrays named like ph_scene_desc's fields (see tests/host_glue.py for a producer)."""
        keep = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        s = PhSceneDesc()
        s.n_nodes = len(keep["prim_type"])
        s.ops = keep["ops"]; s.ops_off = _p(keep["ops_off"], _up); s.args = _p(keep["args"], _dp); s.args_off = _p(keep["args_off"], _up)
        s.prim_type = _p(keep["prim_type"], _ip); s.prim_data = _p(keep["prim_data"], _ip); s.prim_flags = _p(keep["prim_flags"], _ip)
        s.material = _p(keep["material"], _ip); s.child_off = _p(keep["child_off"], _up); s.children = _p(keep["children"], _up)
        s.root = int(keep["root"])
        s.n_meshes = len(keep["mesh_vert_off"]) - 1
        s.mesh_vert_off = _p(keep["mesh_vert_off"], _u64p); s.mesh_tri_off = _p(keep["mesh_tri_off"], _u64p)
        s.mesh_positions = _p(keep["mesh_positions"], _dp); s.mesh_normals = _p(keep["mesh_normals"], _dp)
        s.mesh_has_normals = _p(keep["mesh_has_normals"], _u8p); s.mesh_indices = _p(keep["mesh_indices"], _up)
        s.n_triangles = int(keep["n_triangles"])
        s.tri_vertices = _p(keep["tri_vertices"], _dp); s.tri_normals = _p(keep["tri_normals"], _dp); s.tri_has_normals = _p(keep["tri_has_normals"], _u8p)
        s.n_materials = int(keep["n_materials"]); s.materials = _p(keep["materials"], _dp)
        s.n_lights = int(keep["n_lights"]); s.lights = _p(keep["lights"], _dp)
        s.ambient = (C.c_double * 3)(*map(float, keep["ambient"]))
        if keep.get("n_textures"):
            s.mesh_texcoords = _p(keep["mesh_texcoords"], _dp); s.mesh_has_texcoords = _p(keep["mesh_has_texcoords"], _u8p)
            s.tri_texcoords = _p(keep["tri_texcoords"], _dp); s.tri_has_texcoords = _p(keep["tri_has_texcoords"], _u8p)
            s.material_texture = _p(keep["material_texture"], _ip); s.material_normal_map = _p(keep["material_normal_map"], _ip)
            s.material_uv_trans = _p(keep["material_uv_trans"], _dp)
            s.n_textures = int(keep["n_textures"]); s.texture_size = _p(keep["texture_size"], _up)
            s.texture_offset = _p(keep["texture_offset"], _u64p); s.texture_rgb = _p(keep["texture_rgb"], _u8p)
        h = C.c_void_p()
        _check(lib().ph_scene_create(C.byref(s), C.byref(h)), "ph_scene_create")
        return Scene(h)

    @staticmethod
    def example(name: str, n: int = 10, assets: str = DEFAULT_ASSETS) -> "Scene":
        h = C.c_void_p()
        cam = np.zeros(10); size = np.zeros(2, dtype=np.uint32)
        _check(lib().ph_example_scene(name.encode(), assets.encode(), n, C.byref(h), _p(cam, _dp), _p(size, _up)), "ph_example_scene")
        return Scene(h, cam, (int(size[0]), int(size[1])))

    def close(self):
        if self._h:
            lib().ph_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export(self) -> dict:
        """The scene DAG as arrays (layout of the oracle's po_scene)."""
        c = np.zeros(8, dtype=np.uint64)
        _check(lib().ph_scene_counts(self._h, _p(c, _u64p)), "ph_scene_counts")
        n, nch, nm, nv, nmt, nt, nmat, nl = map(int, c)
        a = dict(node_trans=np.zeros((n, 16)), prim_type=np.zeros(n, dtype=np.int32), prim_data=np.zeros(n, dtype=np.int32),
                 prim_flags=np.zeros(n, dtype=np.int32), material=np.zeros(n, dtype=np.int32), child_off=np.zeros(n + 1, dtype=np.uint32),
                 children=np.zeros(nch + 1, dtype=np.uint32), mesh_vert_off=np.zeros(nm + 1, dtype=np.uint64),
                 mesh_tri_off=np.zeros(nm + 1, dtype=np.uint64), mesh_positions=np.zeros((max(nv, 1), 3)), mesh_normals=np.zeros((max(nv, 1), 3)),
                 mesh_has_normals=np.zeros(nm + 1, dtype=np.uint8), mesh_indices=np.zeros((max(nmt, 1), 3), dtype=np.uint32),
                 tri_vertices=np.zeros((max(nt, 1), 9)), tri_normals=np.zeros((max(nt, 1), 9)), tri_has_normals=np.zeros(nt + 1, dtype=np.uint8),
                 materials=np.zeros((max(nmat, 1), 10)), lights=np.zeros((max(nl, 1), 15)), ambient=np.zeros(3))
        root = C.c_uint32(0)
        _check(lib().ph_scene_export(self._h, _p(a["node_trans"], _dp), _p(a["prim_type"], _ip), _p(a["prim_data"], _ip), _p(a["prim_flags"], _ip),
                                     _p(a["material"], _ip), _p(a["child_off"], _up), _p(a["children"], _up), C.byref(root),
                                     _p(a["mesh_vert_off"], _u64p), _p(a["mesh_tri_off"], _u64p), _p(a["mesh_positions"], _dp),
                                     _p(a["mesh_normals"], _dp), _p(a["mesh_has_normals"], _u8p), _p(a["mesh_indices"], _up),
                                     _p(a["tri_vertices"], _dp), _p(a["tri_normals"], _dp), _p(a["tri_has_normals"], _u8p),
                                     _p(a["materials"], _dp), _p(a["lights"], _dp), _p(a["ambient"], _dp)), "ph_scene_export")
        a.update(root=root.value, n_meshes=nm, n_triangles=nt, n_materials=nmat, n_lights=nl)
        tc = np.zeros(2, dtype=np.uint64)
        _check(lib().ph_scene_export_textures(self._h, _p(tc, _u64p), None, None, None, None, None, None, None, None, None, None), "ph_scene_export_textures")
        ntex, nbytes = int(tc[0]), int(tc[1])
        if ntex:
            t = dict(material_texture=np.zeros(max(nmat, 1), dtype=np.int32), material_normal_map=np.zeros(max(nmat, 1), dtype=np.int32),
                     material_uv_trans=np.zeros((max(nmat, 1), 9)), texture_size=np.zeros((ntex, 2), dtype=np.uint32), texture_offset=np.zeros(ntex, dtype=np.uint64),
                     texture_rgb=np.zeros(max(nbytes, 1), dtype=np.uint8), mesh_texcoords=np.zeros((max(nv, 1), 2)), mesh_has_texcoords=np.zeros(nm + 1, dtype=np.uint8),
                     tri_texcoords=np.zeros((max(nt, 1), 6)), tri_has_texcoords=np.zeros(nt + 1, dtype=np.uint8))
            _check(lib().ph_scene_export_textures(self._h, _p(tc, _u64p), _p(t["material_texture"], _ip), _p(t["material_normal_map"], _ip),
                                                  _p(t["material_uv_trans"], _dp), _p(t["texture_size"], _up), _p(t["texture_offset"], _u64p),
                                                  _p(t["texture_rgb"], _u8p), _p(t["mesh_texcoords"], _dp), _p(t["mesh_has_texcoords"], _u8p),
                                                  _p(t["tri_texcoords"], _dp), _p(t["tri_has_texcoords"], _u8p)), "ph_scene_export_textures")
            a.update(t, n_textures=ntex)
        return a

    def flatten(self) -> dict:
        n = _check(lib().ph_scene_flatten(self._h, 0, None, None, None, None, None, None), "ph_scene_flatten")
        tr, inv, nrm = np.zeros((n, 16)), np.zeros((n, 16)), np.zeros((n, 16))
        pt, mat, b = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros((n, 6))
        _check(lib().ph_scene_flatten(self._h, n, _p(tr, _dp), _p(inv, _dp), _p(nrm, _dp), _p(pt, _ip), _p(mat, _ip), _p(b, _dp)), "ph_scene_flatten")
        return dict(trans=tr.reshape(n, 4, 4), invtrans=inv.reshape(n, 4, 4), normal_trans=nrm.reshape(n, 4, 4), prim_type=pt, material=mat, bounds=b)

    def graph(self) -> dict:
        """The ABI-4 scene-graph arrays PT_TRAVERSE_HIER takes (host logic, no GPU needed)."""
        counts = np.zeros(2, dtype=np.uint32)
        n = _check(lib().ph_scene_graph(self._h, 0, 0, 0, None, None, None, None, None, None, _p(counts, _up)), "ph_scene_graph")
        nc, ng = int(counts[0]), int(counts[1])
        off, chain, rank = np.zeros(n + 1, dtype=np.uint32), np.zeros(max(nc, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint32)
        tr, inv, nrm = np.zeros((ng, 16)), np.zeros((ng, 16)), np.zeros((ng, 16))
        _check(lib().ph_scene_graph(self._h, n, nc, ng, _p(off, _up), _p(chain, _up), _p(rank, _up), _p(tr, _dp), _p(inv, _dp), _p(nrm, _dp), _p(counts, _up)), "ph_scene_graph")
        return dict(chain_off=off, chain=chain[:nc], dfs_rank=rank[:n], trans=tr.reshape(ng, 4, 4), invtrans=inv.reshape(ng, 4, 4), normal_trans=nrm.reshape(ng, 4, 4))

    def kdtree(self, kd_depth: int = 10, node_cap: int = 1 << 16, item_cap: int = 1 << 20) -> dict:
        axis = np.zeros(node_cap, dtype=np.int32); plane = np.zeros(node_cap)
        front, back, first, count = (np.zeros(node_cap, dtype=np.int32) for _ in range(4))
        items = np.zeros(item_cap, dtype=np.int32)
        n_items = C.c_uint32(0); rb = np.zeros(6); md = C.c_int32(0)
        n = _check(lib().ph_scene_kdtree(self._h, kd_depth, node_cap, item_cap, _p(axis, _ip), _p(plane, _dp), _p(front, _ip), _p(back, _ip),
                                         _p(first, _ip), _p(count, _ip), _p(items, _ip), C.byref(n_items), _p(rb, _dp), C.byref(md)), "ph_scene_kdtree")
        return dict(axis=axis[:n], plane=plane[:n], front=front[:n], back=back[:n], first=first[:n], count=count[:n],
                    items=items[:n_items.value], root_bounds=rb, max_depth=md.value)


def camera(cam10, width: float, height: float) -> H.PtCamera:
    out = H.PtCamera()
    c = np.ascontiguousarray(cam10, dtype=np.float64)
    _check(lib().ph_camera(_p(c, _dp), float(width), float(height), C.byref(out)), "ph_camera")
    return out


def load_obj(path: str):
    c = np.zeros(3, dtype=np.uint64)
    _check(lib().ph_obj_load(path.encode(), _p(c, _u64p), None, None, None, 0, 0), "ph_obj_load")
    nv, nt, hn = map(int, c)
    pos, nrm, idx = np.zeros((nv, 3)), np.zeros((nv, 3)), np.zeros((nt, 3), dtype=np.uint32)
    _check(lib().ph_obj_load(path.encode(), _p(c, _u64p), _p(pos, _dp), _p(nrm, _dp), _p(idx, _up), nv, nt), "ph_obj_load")
    return pos, (nrm if hn else None), idx


class Renderer:
    """A flattened scene resident on one MI355X (what render.rs:121-126 prepares, kept across renders)."""

    def __init__(self, scene: Scene, traverse: int = H.TRAVERSE_FLAT, kd_depth: int = 10, device: int = 0):
        self._h = C.c_void_p()
        self.scene = scene
        _check(lib().ph_renderer_create(scene._h, traverse, kd_depth, device, C.byref(self._h)), "ph_renderer_create")

    def close(self):
        if self._h:
            lib().ph_renderer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def context(self):
        return lib().ph_renderer_context(self._h)

    @property
    def ranks(self) -> int:
        return int(lib().ph_renderer_ranks(self._h))

    @property
    def node(self):
        """The pt_node behind this renderer when PORTRAYER_GPUS / PORTRAYER_DEVICES spread it over several ranks, else None."""
        return lib().ph_renderer_node(self._h)

    def prepare_ms(self) -> dict:
        out = np.zeros(5)
        _check(lib().ph_renderer_prepare_ms(self._h, _p(out, _dp)), "ph_renderer_prepare_ms")
        return dict(zip(("flatten", "pack_arrays", "context", "kd_build", "upload_and_device_trees"), map(float, out)))

    def render(self, cam10, width: int, height: int, background: np.ndarray, samples: int = 1, seed: int = 0,
               sample_mode: int = H.SAMPLE_CENTRE, rect=None, stats: bool = False, into: Optional[np.ndarray] = None, want_linear: bool = True):
        bg = np.ascontiguousarray(background, dtype=np.float64)
        rows = 1 if bg.shape == (height, 3) else 0
        if not rows and bg.shape != (height, width, 3):
            raise ValueError("background must be (H, 3) or (H, W, 3)")
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, width - 1, height - 1)
        p = H.PtRenderParams(width, height, H.PtRect(x0, y0, x1, y1), samples, seed, sample_mode, rows, 0, 1, 1 if stats else 0)
        rgb = into if into is not None else np.zeros((height, width, 3), dtype=np.uint8)
        linear = np.zeros((height, width, 3), dtype=np.float64) if want_linear else None
        st = H.PtStats()
        c = np.ascontiguousarray(cam10, dtype=np.float64)
        _check(lib().ph_renderer_render(self._h, _p(c, _dp), C.byref(p), _p(bg, _dp), _p(rgb, _u8p), _p(linear, _dp), C.byref(st)), "ph_renderer_render")
        return rgb, linear, st.as_dict()
