"""ctypes binding of libportrayer_hip.so (include/portrayer_hip.h): the gfx950 kernels' C ABI.

There is no CPU fallback: if the shared library is missing or no MI355X is visible the calls fail
loudly (PortrayerHipError)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libportrayer_hip.so")

TRAVERSE_FLAT, TRAVERSE_KD, TRAVERSE_HIER = 1, 2, 3
SAMPLE_CENTRE, SAMPLE_RNG = 0, 1
# the header's error codes (include/portrayer_hip.h: every entry point returns 0 or one of these)
OK, ERR_ARGUMENT, ERR_DEVICE, ERR_NO_SCENE, ERR_SLICE, ERR_SCENE, ERR_TRAVERSAL = 0, -1, -2, -3, -4, -5, -6
# pt_stats.kernel_variant (ABI 6): low four bits = waves per SIMD
KERNEL_WAVES_MASK, KERNEL_INTERPRETER, KERNEL_PARK, KERNEL_COUNTING, KERNEL_TEXTURED, KERNEL_FORK, KERNEL_CHAIN = 15, 16, 32, 64, 128, 256, 512

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class PortrayerHipError(RuntimeError):
    pass


class PtScene(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("trans", _dp), ("invtrans", _dp), ("normal_trans", _dp),
        ("prim_type", _ip), ("prim_data", _ip), ("prim_flags", _ip), ("material", _ip),
        ("n_meshes", C.c_uint32), ("mesh_vert_off", _u64p), ("mesh_tri_off", _u64p), ("mesh_positions", _dp),
        ("mesh_normals", _dp), ("mesh_has_normals", _u8p), ("mesh_indices", _up), ("mesh_bounds_invtrans", _dp),
        ("n_triangles", C.c_uint32), ("tri_vertices", _dp), ("tri_normals", _dp),
        ("n_materials", C.c_uint32), ("materials", _dp),
        ("n_lights", C.c_uint32), ("lights", _dp),
        ("ambient", C.c_double * 3),
        ("mesh_texcoords", _dp), ("mesh_has_texcoords", _u8p), ("tri_texcoords", _dp), ("tri_has_texcoords", _u8p),
        ("material_texture", _ip), ("material_normal_map", _ip), ("material_uv_trans", _dp),
        ("n_textures", C.c_uint32), ("texture_size", _up), ("texture_offset", _u64p), ("texture_rgb", _u8p),
        ("mesh_kd_root", _ip), ("mesh_kd_depth", _ip), ("mesh_kd_bounds", _dp), ("mesh_kd_bounds_invtrans", _dp),
        ("n_kdm_nodes", C.c_uint32), ("kdm_axis", _ip), ("kdm_plane", _dp), ("kdm_front", _ip), ("kdm_back", _ip), ("kdm_first", _ip), ("kdm_count", _ip),
        ("n_kdm_items", C.c_uint32), ("kdm_items", _ip),
        # ABI 4: the scene graph, for TRAVERSE_HIER (the reference's default traversal, scene.rs:80-120)
        ("n_graph_nodes", C.c_uint32), ("graph_trans", _dp), ("graph_invtrans", _dp), ("graph_normal_trans", _dp),
        ("node_chain_off", _up), ("node_chain", _up), ("node_dfs_rank", _up),
    ]


class PtKdTree(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32), ("axis", _ip), ("plane", _dp), ("front", _ip), ("back", _ip), ("first", _ip), ("count", _ip),
        ("n_items", C.c_uint32), ("leaf_items", _ip),
        ("root_min", C.c_double * 3), ("root_max", C.c_double * 3), ("max_depth", C.c_int32),
    ]


class PtCamera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("view_to_world", C.c_double * 16), ("fov_factor", C.c_double),
                ("aspect_ratio", C.c_double), ("width", C.c_double), ("height", C.c_double)]


class PtRect(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class PtRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("slice", PtRect), ("samples", C.c_uint32), ("seed", C.c_uint64),
                ("sample_mode", C.c_int32), ("background_rows", C.c_int32), ("tile_rank", C.c_uint32), ("tile_ranks", C.c_uint32),
                ("collect_stats", C.c_int32)]


class PtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary", "shadow", "reflect", "refract", "depth11_skipped", "hits", "n_inner", "n_leaf",
                                          "n_analytic", "n_tri", "n_bbox", "kd_plane_miss", "stack_overflow")] + \
               [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("diag", C.c_uint64 * 8), ("kernel_mode", C.c_uint32), ("kernel_variant", C.c_uint32)]

    def as_dict(self):
        d = {n: (float if n.endswith("_ms") else int)(getattr(self, n)) for n, _ in self._fields_ if n != "diag"}
        d["diag"] = [int(x) for x in self.diag]
        return d


_lib: Optional[C.CDLL] = None

EXPORTS = ["pt_abi_version", "pt_device_count", "pt_context_create", "pt_context_destroy", "pt_last_error", "pt_scene_upload",
           "pt_render", "pt_render_device", "pt_render_finish", "pt_context_stream", "pt_context_next_slot", "pt_compact_bytes", "pt_untile_device", "pt_tile_slot_pixel", "pt_untile_host", "pt_device_alloc",
           "pt_device_free", "pt_copy_to_device", "pt_copy_from_device", "pt_synchronize", "pt_measure_copy_bandwidth", "pt_test_cast_rays",
           "pt_test_math", "pt_test_work_items", "pt_node_create", "pt_node_destroy", "pt_node_last_error", "pt_node_ranks", "pt_node_uses_rccl", "pt_node_context",
           "pt_node_scene_upload", "pt_node_render", "pt_node_upload_background", "pt_node_render_resident", "pt_node_download_image",
           "pt_node_device", "pt_node_frame_begin", "pt_node_frame_end", "pt_node_frames_in_flight", "pt_node_last_frame_host_ms", "pt_node_last_frame_rank_kernel_ms", "pt_test_pow_host", "pt_test_libm_host"]


def header_functions():
    """Names of the functions include/portrayer_hip.h declares (every `pt_*(` outside comments)."""
    import re
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "portrayer_hip.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def missing_symbols():
    """Functions the header declares that the loaded library does not export (must be empty)."""
    l = lib()
    return [n for n in header_functions() if not hasattr(l, n)]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PortrayerHipError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        l.pt_abi_version.restype = C.c_int
        l.pt_device_count.restype = C.c_int
        l.pt_context_create.restype = C.c_int
        l.pt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.pt_context_destroy.restype = None
        l.pt_context_destroy.argtypes = [C.c_void_p]
        l.pt_last_error.restype = C.c_char_p
        l.pt_last_error.argtypes = [C.c_void_p]
        l.pt_scene_upload.restype = C.c_int
        l.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(PtScene), C.c_int, C.POINTER(PtKdTree)]
        l.pt_render.restype = C.c_int
        l.pt_render.argtypes = [C.c_void_p, C.POINTER(PtCamera), _dp, C.POINTER(PtRenderParams), _u8p, _dp, C.POINTER(PtStats)]
        l.pt_render_device.restype = C.c_int
        l.pt_render_device.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.c_void_p, C.POINTER(PtRenderParams), C.c_int, C.c_void_p, C.c_void_p]
        l.pt_render_finish.restype = C.c_int
        l.pt_render_finish.argtypes = [C.c_void_p, C.POINTER(PtStats)]
        l.pt_context_stream.restype = C.c_void_p
        l.pt_context_stream.argtypes = [C.c_void_p, C.c_int]
        l.pt_context_next_slot.restype = C.c_int
        l.pt_context_next_slot.argtypes = [C.c_void_p]
        l.pt_compact_bytes.restype = C.c_uint64
        l.pt_compact_bytes.argtypes = [C.POINTER(PtRenderParams)]
        l.pt_untile_device.restype = C.c_int
        l.pt_untile_device.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
        l.pt_tile_slot_pixel.restype = C.c_int
        l.pt_tile_slot_pixel.argtypes = [C.POINTER(PtRenderParams), C.c_uint32, C.c_uint32, _up, _up]
        l.pt_untile_host.restype = C.c_int
        l.pt_untile_host.argtypes = [C.POINTER(PtRenderParams), _u8p, _u8p]
        l.pt_device_alloc.restype = C.c_int
        l.pt_device_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        l.pt_device_free.restype = C.c_int
        l.pt_device_free.argtypes = [C.c_void_p, C.c_void_p]
        l.pt_copy_to_device.restype = C.c_int
        l.pt_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        l.pt_copy_from_device.restype = C.c_int
        l.pt_copy_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        l.pt_synchronize.restype = C.c_int
        l.pt_synchronize.argtypes = [C.c_void_p]
        l.pt_measure_copy_bandwidth.restype = C.c_int
        l.pt_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _dp]
        l.pt_test_cast_rays.restype = C.c_int
        l.pt_test_cast_rays.argtypes = [C.c_void_p, C.c_uint64, _dp, _dp, C.c_int, _dp, _ip, _ip]
        l.pt_test_work_items.restype = C.c_int
        l.pt_test_work_items.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        l.pt_test_math.restype = C.c_int
        l.pt_test_math.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _dp, _dp, _dp]
        l.pt_node_create.restype = C.c_int
        l.pt_node_create.argtypes = [C.c_int, _ip, C.POINTER(C.c_void_p)]
        l.pt_node_destroy.restype = None
        l.pt_node_destroy.argtypes = [C.c_void_p]
        l.pt_node_last_error.restype = C.c_char_p
        l.pt_node_last_error.argtypes = [C.c_void_p]
        l.pt_node_ranks.restype = C.c_int
        l.pt_node_ranks.argtypes = [C.c_void_p]
        l.pt_node_uses_rccl.restype = C.c_int
        l.pt_node_uses_rccl.argtypes = [C.c_void_p]
        l.pt_node_context.restype = C.c_void_p
        l.pt_node_context.argtypes = [C.c_void_p, C.c_int]
        l.pt_node_scene_upload.restype = C.c_int
        l.pt_node_scene_upload.argtypes = [C.c_void_p, C.POINTER(PtScene), C.c_int, C.POINTER(PtKdTree)]
        l.pt_node_render.restype = C.c_int
        l.pt_node_render.argtypes = [C.c_void_p, C.POINTER(PtCamera), _dp, C.POINTER(PtRenderParams), _u8p, C.POINTER(PtStats)]
        l.pt_node_upload_background.restype = C.c_int
        l.pt_node_upload_background.argtypes = [C.c_void_p, _dp, C.POINTER(PtRenderParams), _u8p]
        l.pt_node_render_resident.restype = C.c_int
        l.pt_node_render_resident.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtRenderParams), C.POINTER(PtStats)]
        l.pt_node_download_image.restype = C.c_int
        l.pt_node_download_image.argtypes = [C.c_void_p, C.POINTER(PtRenderParams), _u8p]
        l.pt_node_device.restype = C.c_int
        l.pt_node_device.argtypes = [C.c_void_p, C.c_int]
        l.pt_node_frame_begin.restype = C.c_int
        l.pt_node_frame_begin.argtypes = [C.c_void_p, C.POINTER(PtCamera), C.POINTER(PtRenderParams)]
        l.pt_node_frame_end.restype = C.c_int
        l.pt_node_frame_end.argtypes = [C.c_void_p, C.POINTER(PtStats)]
        l.pt_node_frames_in_flight.restype = C.c_int
        l.pt_node_frames_in_flight.argtypes = [C.c_void_p]
        l.pt_node_last_frame_host_ms.restype = C.c_int
        l.pt_node_last_frame_host_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double * 5)]
        l.pt_node_last_frame_rank_kernel_ms.restype = C.c_int
        l.pt_node_last_frame_rank_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        l.pt_test_pow_host.restype = C.c_int
        l.pt_test_pow_host.argtypes = [C.c_uint64, _dp, _dp, _dp, _dp]
        l.pt_test_libm_host.restype = C.c_int
        l.pt_test_libm_host.argtypes = [C.c_int, C.c_uint64, _dp, _dp, _dp]
        _lib = l
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class Context:
    """One pt_context (one GPU)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().pt_context_create(device, C.byref(self._h))
        if rc != 0:
            raise PortrayerHipError(f"pt_context_create(device={device}) failed with {rc}: no usable MI355X (there is no CPU fallback)")

    def close(self):
        if self._h:
            lib().pt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != 0:
            raise PortrayerHipError(f"{what} failed with {rc}: {lib().pt_last_error(self._h).decode()}")

    @property
    def handle(self):
        return self._h

    def upload(self, scene: PtScene, traverse: int, kd: Optional[PtKdTree] = None):
        self.check(lib().pt_scene_upload(self._h, C.byref(scene), traverse, C.byref(kd) if kd is not None else None), "pt_scene_upload")

    def render(self, cam: PtCamera, background: np.ndarray, params: PtRenderParams, rgb: np.ndarray, linear: Optional[np.ndarray] = None) -> dict:
        st = PtStats()
        bg = np.ascontiguousarray(background, dtype=np.float64)
        self.check(lib().pt_render(self._h, C.byref(cam), _p(bg, _dp), C.byref(params), _p(rgb, _u8p), _p(linear, _dp), C.byref(st)), "pt_render")
        return st.as_dict()

    def cast_rays(self, origins, directions, any_hit=False):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = len(o)
        t = np.zeros(n); node = np.zeros(n, dtype=np.int32); sub = np.zeros(n, dtype=np.int32)
        self.check(lib().pt_test_cast_rays(self._h, n, _p(o, _dp), _p(d, _dp), int(any_hit), _p(t, _dp), _p(node, _ip), _p(sub, _ip)), "pt_test_cast_rays")
        return t, node, sub

    def math(self, op: int, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros_like(a)
        self.check(lib().pt_test_math(self._h, op, a.size, _p(a, _dp), _p(b, _dp), _p(out, _dp)), "pt_test_math")
        return out

    def copy_bandwidth(self, nbytes: int = 1 << 30, iters: int = 5) -> float:
        g = C.c_double(0.0)
        self.check(lib().pt_measure_copy_bandwidth(self._h, nbytes, iters, C.byref(g)), "pt_measure_copy_bandwidth")
        return g.value
