"""ctypes binding of the CPU oracle (oracle/libportrayer_oracle.so). Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
portrayer_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

from scene_dsl import Camera, Linearised, Scene, linearise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libportrayer_oracle.so")

MODE_HIER, MODE_FLAT, MODE_KD = 0, 1, 2
JITTER_CENTRE, JITTER_RNG = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class PoScene(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("node_trans", _dp), ("node_prim_type", _ip), ("node_prim_data", _ip),
        ("node_prim_flags", _ip), ("node_material", _ip), ("node_child_off", _up), ("children", _up),
        ("root", C.c_uint32),
        ("n_meshes", C.c_uint32), ("mesh_vert_off", _u64p), ("mesh_tri_off", _u64p), ("mesh_positions", _dp),
        ("mesh_normals", _dp), ("mesh_has_normals", _u8p), ("mesh_indices", _up),
        ("n_triangles", C.c_uint32), ("tri_vertices", _dp), ("tri_normals", _dp), ("tri_has_normals", _u8p),
        ("n_materials", C.c_uint32), ("materials", _dp),
        ("n_lights", C.c_uint32), ("lights", _dp),
        ("ambient", C.c_double * 3),
        ("mesh_texcoords", _dp), ("mesh_has_texcoords", _u8p), ("tri_texcoords", _dp), ("tri_has_texcoords", _u8p),
        ("material_texture", _ip), ("material_normal_map", _ip), ("material_uv_trans", _dp),
        ("n_textures", C.c_uint32), ("texture_size", _up), ("texture_offset", _u64p), ("texture_rgb", _u8p),
    ]


class PoCamera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("center", C.c_double * 3), ("up", C.c_double * 3), ("fovy_radians", C.c_double)]


class PoStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary", "shadow", "reflect", "refract", "depth11", "hits", "n_split",
                                          "n_leaf", "n_analytic", "n_tri", "n_bbox", "kd_plane_miss", "tex_sphere_lookups", "tex_sphere_near_edge")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PoRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("x0", C.c_uint32), ("y0", C.c_uint32),
                ("x1", C.c_uint32), ("y1", C.c_uint32), ("samples", C.c_uint32), ("seed", C.c_uint64),
                ("jitter_mode", C.c_int32), ("mode", C.c_int32), ("kd_depth", C.c_int32), ("kd_mesh_depth", C.c_int32),
                ("background_rows", C.c_int32), ("n_threads", C.c_int32)]


_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    """Compile the oracle with the recipe committed under oracle/ (gcc, -ffp-contract=off)."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("portrayer_oracle.c", "portrayer_oracle.h", "po_math.h", "Makefile")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.po_render.restype = C.c_int
        _lib.po_render.argtypes = [C.POINTER(PoScene), C.POINTER(PoCamera), _dp, C.POINTER(PoRenderParams), _u8p, _dp, C.POINTER(PoStats)]
        _lib.po_last_render_ms.restype = None
        _lib.po_last_render_ms.argtypes = [_dp]
        _lib.po_cast_rays.restype = C.c_int
        _lib.po_cast_rays.argtypes = [C.POINTER(PoScene), C.c_int, C.c_int, C.c_int, C.c_uint64, _dp, _dp, _dp, _ip, _dp, _dp]
        _lib.po_color_rays.restype = C.c_int
        _lib.po_color_rays.argtypes = [C.POINTER(PoScene), C.c_int, C.c_int, C.c_int, C.c_uint64, _dp, _dp, _dp, _dp]
        _lib.po_camera_rays.restype = C.c_int
        _lib.po_camera_rays.argtypes = [C.POINTER(PoCamera), C.c_double, C.c_double, C.c_uint64, _dp, _dp, _dp]
        _lib.po_flatten.restype = C.c_int
        _lib.po_flatten.argtypes = [C.POINTER(PoScene), C.c_uint32, _dp, _dp, _dp, _ip, _ip, _ip, _ip, _dp]
        _lib.po_kd_scene_dump.restype = C.c_int
        _lib.po_kd_scene_dump.argtypes = [C.POINTER(PoScene), C.c_int, C.c_uint32, C.c_uint32, _ip, _ip, _dp, _ip, _ip, _ip, _ip, _ip, _up, _dp]
        _lib.po_kd_partition_boxes.restype = C.c_int
        _lib.po_kd_partition_boxes.argtypes = [C.c_uint32, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                               _ip, _ip, _dp, _ip, _ip, _ip, _ip, _ip, _up]
        _lib.po_kd_cast_custom.restype = C.c_int
        _lib.po_kd_cast_custom.argtypes = [C.POINTER(PoScene), _ip, _ip, _dp, _ip, _ip, _ip, _ip, _ip, _dp, C.c_uint64, _dp, _dp, _dp, _ip]
        _lib.po_quadratic_solve.restype = C.c_int
        _lib.po_quadratic_solve.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
        _lib.po_transform_bounds.restype = None
        _lib.po_transform_bounds.argtypes = [_dp, _dp, _dp, _dp, _dp]
        _lib.po_mat4_compose.restype = None
        _lib.po_mat4_compose.argtypes = [C.c_char_p, _dp, _dp]
        _lib.po_mat4_inverse.restype = None
        _lib.po_mat4_inverse.argtypes = [_dp, _dp]
        _lib.po_rng_draw.restype = C.c_double
        _lib.po_rng_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    return _lib


def _p(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


def compose(ops) -> np.ndarray:
    """Matrix of a node's builder calls through the oracle's vek restatement (scene.rs:163-205)."""
    s = "".join(o for o, _ in ops).encode()
    args = np.array([x for _, a in ops for x in a] + [0.0], dtype=np.float64)
    out = np.zeros(16, dtype=np.float64)
    lib().po_mat4_compose(s, _p(args, _dp), _p(out, _dp))
    return out.reshape(4, 4)


def mat4_inverse(m: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
    out = np.zeros(16, dtype=np.float64)
    lib().po_mat4_inverse(_p(a, _dp), _p(out, _dp))
    return out.reshape(4, 4)


@dataclass
class PackedScene:
    """Owns the numpy buffers a PoScene points into."""
    struct: PoScene
    arrays: dict
    lin: Linearised

    @property
    def ref(self):
        return C.byref(self.struct)


def arrays_from_dsl(scene: Scene):
    """Linearises a DSL scene into the po_scene arrays; node matrices come from the oracle's own
    builder-call composition (po_mat4_compose)."""
    lin = linearise(scene)
    n = len(lin.nodes)
    a = {}
    a["node_trans"] = np.stack([compose(o) for o in lin.ops]).reshape(n, 16).astype(np.float64)
    a["prim_type"] = np.array(lin.prim_type, dtype=np.int32)
    a["prim_data"] = np.array(lin.prim_data, dtype=np.int32)
    a["prim_flags"] = np.array(lin.prim_flags, dtype=np.int32)
    a["material"] = np.array(lin.material, dtype=np.int32)
    a["child_off"] = np.array(lin.child_off, dtype=np.uint32)
    a["children"] = np.array(lin.children + [0], dtype=np.uint32)
    vo, to = [0], [0]
    for m in lin.meshes:
        vo.append(vo[-1] + len(m.positions)); to.append(to[-1] + len(m.triangles))
    a["mesh_vert_off"] = np.array(vo, dtype=np.uint64)
    a["mesh_tri_off"] = np.array(to, dtype=np.uint64)
    a["mesh_positions"] = (np.concatenate([m.positions for m in lin.meshes]) if lin.meshes else np.zeros((1, 3))).astype(np.float64)
    a["mesh_normals"] = (np.concatenate([m.normals if m.normals is not None else np.zeros_like(m.positions) for m in lin.meshes])
                         if lin.meshes else np.zeros((1, 3))).astype(np.float64)
    a["mesh_has_normals"] = np.array([1 if m.normals is not None else 0 for m in lin.meshes] + [0], dtype=np.uint8)
    a["mesh_indices"] = (np.concatenate([m.triangles for m in lin.meshes]) if lin.meshes else np.zeros((1, 3))).astype(np.uint32)
    nt = len(lin.triangles)
    a["tri_vertices"] = (np.stack([t.tri.reshape(9) for t in lin.triangles]) if nt else np.zeros((1, 9))).astype(np.float64)
    a["tri_normals"] = (np.stack([(t.tri_normals if t.tri_normals is not None else np.zeros((3, 3))).reshape(9) for t in lin.triangles])
                        if nt else np.zeros((1, 9))).astype(np.float64)
    a["tri_has_normals"] = np.array([1 if t.tri_normals is not None else 0 for t in lin.triangles] + [0], dtype=np.uint8)
    a["materials"] = np.array([m.row() for m in lin.materials] or [[0.0] * 10], dtype=np.float64)
    a["lights"] = np.array([l.row() for l in scene.lights] or [[0.0] * 15], dtype=np.float64)
    a["ambient"] = np.array(list(map(float, scene.ambient)))
    # textures (texture.rs), texture coordinates (mesh.rs:30, triangle.rs:18), per-material maps
    a["mesh_texcoords"] = (np.concatenate([m.tex_coords if m.tex_coords is not None else np.zeros((len(m.positions), 2)) for m in lin.meshes])
                           if lin.meshes else np.zeros((1, 2))).astype(np.float64)
    a["mesh_has_texcoords"] = np.array([1 if m.tex_coords is not None else 0 for m in lin.meshes] + [0], dtype=np.uint8)
    a["tri_texcoords"] = (np.stack([(t.tri_tex_coords if t.tri_tex_coords is not None else np.zeros((3, 2))).reshape(6) for t in lin.triangles])
                          if nt else np.zeros((1, 6))).astype(np.float64)
    a["tri_has_texcoords"] = np.array([1 if t.tri_tex_coords is not None else 0 for t in lin.triangles] + [0], dtype=np.uint8)
    tid = lin.texture_index
    a["material_texture"] = np.array([tid[id(m.texture)] if m.texture is not None else -1 for m in lin.materials] + [-1], dtype=np.int32)
    a["material_normal_map"] = np.array([tid[id(m.normals)] if m.normals is not None else -1 for m in lin.materials] + [-1], dtype=np.int32)
    a["material_uv_trans"] = np.array([list(map(float, m.uv_trans)) for m in lin.materials] or [[1, 0, 0, 0, 1, 0, 0, 0, 1]], dtype=np.float64)
    a["texture_size"] = np.array([[t.pixels.shape[1], t.pixels.shape[0]] for t in lin.textures] or [[0, 0]], dtype=np.uint32)
    offs, blob = [], []
    for t in lin.textures:
        offs.append(sum(len(b) for b in blob))
        blob.append(np.ascontiguousarray(t.pixels, dtype=np.uint8).reshape(-1))
    a["texture_offset"] = np.array(offs + [0], dtype=np.uint64)
    a["texture_rgb"] = np.concatenate(blob) if blob else np.zeros(4, dtype=np.uint8)
    a["n_textures"] = len(lin.textures)
    a.update(root=lin.root, n_meshes=len(lin.meshes), n_triangles=nt, n_materials=len(lin.materials), n_lights=len(scene.lights))
    return a, lin


def pack_arrays(a: dict, lin=None) -> PackedScene:
    """po_scene over a dict of arrays (from arrays_from_dsl, or exported by the product's host
    library: portrayer_amd.host.Scene.export())."""
    a = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in a.items()}
    s = PoScene()
    s.n_nodes = len(a["prim_type"])
    s.node_trans = _p(a["node_trans"], _dp); s.node_prim_type = _p(a["prim_type"], _ip)
    s.node_prim_data = _p(a["prim_data"], _ip); s.node_prim_flags = _p(a["prim_flags"], _ip)
    s.node_material = _p(a["material"], _ip); s.node_child_off = _p(a["child_off"], _up)
    s.children = _p(a["children"], _up); s.root = int(a["root"])
    s.n_meshes = int(a["n_meshes"])
    s.mesh_vert_off = _p(a["mesh_vert_off"], _u64p); s.mesh_tri_off = _p(a["mesh_tri_off"], _u64p)
    s.mesh_positions = _p(a["mesh_positions"], _dp); s.mesh_normals = _p(a["mesh_normals"], _dp)
    s.mesh_has_normals = _p(a["mesh_has_normals"], _u8p); s.mesh_indices = _p(a["mesh_indices"], _up)
    s.n_triangles = int(a["n_triangles"])
    s.tri_vertices = _p(a["tri_vertices"], _dp); s.tri_normals = _p(a["tri_normals"], _dp)
    s.tri_has_normals = _p(a["tri_has_normals"], _u8p)
    s.n_materials = int(a["n_materials"]); s.materials = _p(a["materials"], _dp)
    s.n_lights = int(a["n_lights"]); s.lights = _p(a["lights"], _dp)
    s.ambient = (C.c_double * 3)(*map(float, a["ambient"]))
    if "texture_rgb" in a:
        s.mesh_texcoords = _p(a["mesh_texcoords"], _dp); s.mesh_has_texcoords = _p(a["mesh_has_texcoords"], _u8p)
        s.tri_texcoords = _p(a["tri_texcoords"], _dp); s.tri_has_texcoords = _p(a["tri_has_texcoords"], _u8p)
        s.material_texture = _p(a["material_texture"], _ip); s.material_normal_map = _p(a["material_normal_map"], _ip)
        s.material_uv_trans = _p(a["material_uv_trans"], _dp)
        s.n_textures = int(a["n_textures"]); s.texture_size = _p(a["texture_size"], _up)
        s.texture_offset = _p(a["texture_offset"], _u64p); s.texture_rgb = _p(a["texture_rgb"], _u8p)
    return PackedScene(s, a, lin)


def pack(scene: Scene) -> PackedScene:
    a, lin = arrays_from_dsl(scene)
    return pack_arrays(a, lin)


def camera_struct(cam) -> PoCamera:
    """cam: a scene_dsl.Camera, or 10 doubles (eye, center, up, fovy in radians)."""
    c = PoCamera()
    if not isinstance(cam, Camera):
        v = [float(x) for x in cam]
        c.eye = (C.c_double * 3)(*v[0:3]); c.center = (C.c_double * 3)(*v[3:6]); c.up = (C.c_double * 3)(*v[6:9]); c.fovy_radians = v[9]
        return c
    c.eye = (C.c_double * 3)(*map(float, cam.eye)); c.center = (C.c_double * 3)(*map(float, cam.center))
    c.up = (C.c_double * 3)(*map(float, cam.up)); c.fovy_radians = cam.fovy_radians
    return c


@dataclass
class RenderResult:
    rgb: np.ndarray      # (H, W, 3) u8
    linear: np.ndarray   # (H, W, 3) f64, sample mean before gamma
    stats: dict


def render(scene, cam: Camera, width: int, height: int, background: Optional[np.ndarray] = None, samples: int = 1,
           seed: int = 0, jitter: int = JITTER_CENTRE, mode: int = MODE_FLAT, kd_depth: int = -1, kd_mesh_depth: int = -1,
           rect=None, threads: int = 0, into: Optional[np.ndarray] = None) -> RenderResult:
    from scene_dsl import default_background
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    bg = default_background(width, height) if background is None else np.ascontiguousarray(background, dtype=np.float64)
    rows = 1 if bg.shape == (height, 3) else 0
    if not rows:
        assert bg.shape == (height, width, 3)
    bg = np.ascontiguousarray(bg)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, width - 1, height - 1)
    p = PoRenderParams(width, height, x0, y0, x1, y1, samples, seed, jitter, mode, kd_depth, kd_mesh_depth, rows, threads)
    rgb = into if into is not None else np.zeros((height, width, 3), dtype=np.uint8)
    linear = np.zeros((height, width, 3), dtype=np.float64)
    st = PoStats()
    c = camera_struct(cam)
    rc = lib().po_render(ps.ref, C.byref(c), _p(bg, _dp), C.byref(p), _p(rgb, _u8p), _p(linear, _dp), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"po_render failed: {rc}")
    return RenderResult(rgb, linear, st.as_dict())


def last_render_ms():
    """(scene conversion ms, pixel loop ms) of the last render() call: render.rs:115-126 / :127-150."""
    out = np.zeros(2)
    lib().po_last_render_ms(_p(out, _dp))
    return float(out[0]), float(out[1])


def cast_rays(scene, origins, directions, mode=MODE_FLAT, kd_depth=-1, kd_mesh_depth=-1):
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    n = len(o)
    t = np.zeros(n); ids = np.zeros(n, dtype=np.int32); pt = np.zeros((n, 3)); nr = np.zeros((n, 3))
    rc = lib().po_cast_rays(ps.ref, mode, kd_depth, kd_mesh_depth, n, _p(o, _dp), _p(d, _dp), _p(t, _dp), _p(ids, _ip), _p(pt, _dp), _p(nr, _dp))
    if rc != 0:
        raise RuntimeError(f"po_cast_rays failed: {rc}")
    return t, ids, pt, nr


def color_rays(scene, origins, directions, background=(0.0, 0.0, 0.0), mode=MODE_FLAT, kd_depth=-1, kd_mesh_depth=-1):
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    bg = np.array(background, dtype=np.float64)
    out = np.zeros((len(o), 3))
    rc = lib().po_color_rays(ps.ref, mode, kd_depth, kd_mesh_depth, len(o), _p(o, _dp), _p(d, _dp), _p(bg, _dp), _p(out, _dp))
    if rc != 0:
        raise RuntimeError(f"po_color_rays failed: {rc}")
    return out


def camera_rays(cam: Camera, width: float, height: float, xy):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    o = np.zeros((len(xy), 3)); d = np.zeros((len(xy), 3))
    c = camera_struct(cam)
    lib().po_camera_rays(C.byref(c), float(width), float(height), len(xy), _p(xy, _dp), _p(o, _dp), _p(d, _dp))
    return o, d


def flatten(scene):
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    n = lib().po_flatten(ps.ref, 0, None, None, None, None, None, None, None, None)
    if n < 0:
        raise RuntimeError(f"po_flatten failed: {n}")
    tr = np.zeros((n, 16)); inv = np.zeros((n, 16)); nrm = np.zeros((n, 16))
    pt = np.zeros(n, dtype=np.int32); pd = np.zeros(n, dtype=np.int32); pf = np.zeros(n, dtype=np.int32); mat = np.zeros(n, dtype=np.int32)
    b = np.zeros((n, 6))
    lib().po_flatten(ps.ref, n, _p(tr, _dp), _p(inv, _dp), _p(nrm, _dp), _p(pt, _ip), _p(pd, _ip), _p(pf, _ip), _p(mat, _ip), _p(b, _dp))
    return dict(trans=tr.reshape(n, 4, 4), invtrans=inv.reshape(n, 4, 4), normal_trans=nrm.reshape(n, 4, 4), prim_type=pt,
                prim_data=pd, prim_flags=pf, material=mat, bounds=b)


def _dump_buffers(node_cap, item_cap):
    return dict(kind=np.zeros(node_cap, dtype=np.int32), axis=np.zeros(node_cap, dtype=np.int32), plane=np.zeros(node_cap),
                front=np.zeros(node_cap, dtype=np.int32), back=np.zeros(node_cap, dtype=np.int32),
                first=np.zeros(node_cap, dtype=np.int32), count=np.zeros(node_cap, dtype=np.int32),
                items=np.zeros(item_cap, dtype=np.int32))


def _trim(b, n, n_items):
    out = {k: v[:n] for k, v in b.items() if k != "items"}
    out["items"] = b["items"][:n_items]
    return out


def kd_scene_dump(scene, kd_depth=-1, node_cap=1 << 16, item_cap=1 << 20):
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    b = _dump_buffers(node_cap, item_cap)
    n_items = C.c_uint32(0)
    rb = np.zeros(6)
    n = lib().po_kd_scene_dump(ps.ref, kd_depth, node_cap, item_cap, _p(b["kind"], _ip), _p(b["axis"], _ip), _p(b["plane"], _dp),
                               _p(b["front"], _ip), _p(b["back"], _ip), _p(b["first"], _ip), _p(b["count"], _ip),
                               _p(b["items"], _ip), C.byref(n_items), _p(rb, _dp))
    if n < 0:
        raise RuntimeError(f"po_kd_scene_dump failed: {n}")
    out = _trim(b, n, n_items.value)
    out["root_bounds"] = rb
    return out


def kd_partition_boxes(mins, maxs, max_depth, target_max_nodes=3, target_max_merit=3, max_tries=10):
    mins = np.ascontiguousarray(mins, dtype=np.float64).reshape(-1, 3)
    maxs = np.ascontiguousarray(maxs, dtype=np.float64).reshape(-1, 3)
    node_cap, item_cap = 1 << 16, 1 << 20
    b = _dump_buffers(node_cap, item_cap)
    n_items = C.c_uint32(0)
    n = lib().po_kd_partition_boxes(len(mins), _p(mins, _dp), _p(maxs, _dp), max_depth, target_max_nodes, target_max_merit, max_tries,
                                    node_cap, item_cap, _p(b["kind"], _ip), _p(b["axis"], _ip), _p(b["plane"], _dp),
                                    _p(b["front"], _ip), _p(b["back"], _ip), _p(b["first"], _ip), _p(b["count"], _ip),
                                    _p(b["items"], _ip), C.byref(n_items))
    if n < 0:
        raise RuntimeError(f"po_kd_partition_boxes failed: {n}")
    return _trim(b, n, n_items.value)


def quadratic(a, b, c):
    out = np.zeros(2)
    n = lib().po_quadratic_solve(a, b, c, _p(out, _dp))
    return list(out[:n])


def transform_bounds(trans, mn, mx):
    t = np.ascontiguousarray(trans, dtype=np.float64).reshape(16)
    a = np.array(mn, dtype=np.float64); b = np.array(mx, dtype=np.float64)
    omin = np.zeros(3); omax = np.zeros(3)
    lib().po_transform_bounds(_p(t, _dp), _p(a, _dp), _p(b, _dp), _p(omin, _dp), _p(omax, _dp))
    return omin, omax


def rng_draw(seed, pixel, sample, draw):
    return lib().po_rng_draw(seed, pixel, sample, draw)


def kd_cast_custom(scene, tree: dict, root_bounds, origins, directions):
    """tree: dict(kind, axis, plane, front, back, first, count, items) in the dump format."""
    ps = scene if isinstance(scene, PackedScene) else pack(scene)
    i32 = lambda k: np.ascontiguousarray(tree[k], dtype=np.int32)
    kind, axis, front, back, first, count, items = map(i32, ("kind", "axis", "front", "back", "first", "count", "items"))
    plane = np.ascontiguousarray(tree["plane"], dtype=np.float64)
    rb = np.ascontiguousarray(root_bounds, dtype=np.float64).reshape(6)
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    t = np.zeros(len(o)); ids = np.zeros(len(o), dtype=np.int32)
    rc = lib().po_kd_cast_custom(ps.ref, _p(kind, _ip), _p(axis, _ip), _p(plane, _dp), _p(front, _ip), _p(back, _ip), _p(first, _ip),
                                 _p(count, _ip), _p(items, _ip), _p(rb, _dp), len(o), _p(o, _dp), _p(d, _dp), _p(t, _dp), _p(ids, _ip))
    if rc != 0:
        raise RuntimeError(f"po_kd_cast_custom failed: {rc}")
    return t, ids
