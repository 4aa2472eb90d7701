"""Scene-description DSL for the tests (test infrastructure, not product code).

A neutral, Python-side mirror of the reference's builder API (src/scene.rs:36-206,
src/material.rs:50-86, src/light.rs:74-91, src/camera.rs:5-14, src/primitive/mesh.rs:21-75): a
node records the ORDER of its builder calls instead of a matrix, so the same description can be
  * packed for the oracle (matrices composed by the oracle's own vek restatement), and
  * replayed call by call on the product's host API (portrayer_amd), whose matrices are then
    compared bit for bit with the oracle's.

The OBJ reader follows tobj 0.1.7 as used by mesh.rs:36-61 (f32 parse widened to f64, fan
triangulation, vertices de-duplicated per (v, vt, vn) tuple, first model only; SURVEY App.B.6).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ASSETS = os.path.join(GOLDEN, "assets")

# primitive.rs:67-81 tags shared by the oracle (PO_*) and the product (PT_*)
NONE, SPHERE, TRIANGLE, MESH, KDMESH, PLANE, CUBE, CYLINDER, CONE = -1, 0, 1, 2, 3, 4, 5, 6, 7


def to_radians(deg: float) -> float:
    """f64::to_radians: self * (PI / 180.0) (math.rs:59-61)."""
    return deg * (math.pi / 180.0)


def _vec3(v) -> Tuple[float, float, float]:
    if isinstance(v, (int, float)):
        return (float(v), float(v), float(v))  # Vec3::from(f64) broadcasts
    x, y, z = v
    return (float(x), float(y), float(z))


@dataclass
class Texture:  # texture.rs:74-76 RgbImageBuffer: RGB8 pixels as decoded from the image file
    pixels: np.ndarray  # (H, W, 3) uint8

    @staticmethod
    def open(path: str) -> "Texture":
        from PIL import Image
        return Texture(np.ascontiguousarray(np.array(Image.open(path).convert("RGB"), dtype=np.uint8)))


@dataclass
class Material:  # material.rs:50-86
    diffuse: Sequence[float] = (0.0, 0.0, 0.0)
    specular: Sequence[float] = (0.0, 0.0, 0.0)
    shininess: float = 0.0
    reflectivity: float = 0.0
    glossy_side_length: float = 0.0
    refraction_index: float = 0.0
    texture: Optional["Texture"] = None      # ImageTexture (sRGB -> linear on sampling, texture.rs:162-168)
    normals: Optional["Texture"] = None      # NormalMap (texture.rs:175-221)
    uv_trans: Sequence[float] = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)  # row-major Mat3 (material.rs:83)

    def row(self) -> List[float]:
        return [*map(float, self.diffuse), *map(float, self.specular), float(self.shininess),
                float(self.reflectivity), float(self.glossy_side_length), float(self.refraction_index)]


@dataclass
class Light:  # light.rs:74-91
    position: Sequence[float] = (0.0, 0.0, 0.0)
    color: Sequence[float] = (0.0, 0.0, 0.0)
    falloff: Sequence[float] = (1.0, 0.0, 0.0)  # light.rs:19-28 default
    area_a: Sequence[float] = (0.0, 0.0, 0.0)
    area_b: Sequence[float] = (0.0, 0.0, 0.0)

    def row(self) -> List[float]:
        return [*map(float, self.position), *map(float, self.color), *map(float, self.falloff),
                *map(float, self.area_a), *map(float, self.area_b)]


@dataclass
class MeshData:  # mesh.rs:21-34
    positions: np.ndarray  # (n, 3) f64
    triangles: np.ndarray  # (m, 3) u32
    normals: Optional[np.ndarray] = None  # (n, 3) f64
    name: str = ""
    tex_coords: Optional[np.ndarray] = None  # (n, 2) f64

    @staticmethod
    def load_obj(path: str) -> "MeshData":
        pos, nrm, tex = [], [], []
        out_pos, out_nrm, out_tex, tris = [], [], [], []
        index_map = {}
        seen_faces = False
        with open(path, "r") as fh:
            for line in fh:
                parts = line.split()
                if not parts:
                    continue
                tag = parts[0]
                if tag == "v":
                    pos.append([float(np.float32(p)) for p in parts[1:4]])
                elif tag == "vn":
                    nrm.append([float(np.float32(p)) for p in parts[1:4]])
                elif tag == "vt":
                    tex.append([float(np.float32(p)) for p in parts[1:3]])
                elif tag in ("o", "g"):
                    if seen_faces:
                        break  # models[0] only (mesh.rs:60)
                elif tag == "f":
                    seen_faces = True
                    corner = []
                    for tok in parts[1:]:
                        f = tok.split("/")
                        v = int(f[0])
                        vt = int(f[1]) if len(f) > 1 and f[1] else 0
                        vn = int(f[2]) if len(f) > 2 and f[2] else 0
                        v = v - 1 if v > 0 else len(pos) + v
                        vt = vt - 1 if vt > 0 else (len(tex) + vt if vt < 0 else -1)
                        vn = vn - 1 if vn > 0 else (len(nrm) + vn if vn < 0 else -1)
                        key = (v, vt, vn)
                        if key not in index_map:
                            index_map[key] = len(out_pos)
                            out_pos.append(pos[v])
                            if vn >= 0:
                                out_nrm.append(nrm[vn])
                            if vt >= 0:
                                out_tex.append(tex[vt])
                        corner.append(index_map[key])
                    for k in range(1, len(corner) - 1):  # fan
                        tris.append([corner[0], corner[k], corner[k + 1]])
        normals = np.array(out_nrm, dtype=np.float64) if len(out_nrm) == len(out_pos) and out_nrm else None
        tex_coords = np.array(out_tex, dtype=np.float64) if len(out_tex) == len(out_pos) and out_tex else None
        return MeshData(np.array(out_pos, dtype=np.float64).reshape(-1, 3),
                        np.array(tris, dtype=np.uint32).reshape(-1, 3), normals, os.path.basename(path), tex_coords)


@dataclass
class Prim:
    kind: int
    mesh: Optional[MeshData] = None
    smooth: bool = False
    tri: Optional[np.ndarray] = None  # (3, 3) a, b, c
    tri_normals: Optional[np.ndarray] = None
    tri_tex_coords: Optional[np.ndarray] = None  # (3, 2)


Sphere = lambda: Prim(SPHERE)
Cube = lambda: Prim(CUBE)
Plane = lambda: Prim(PLANE)
Cylinder = lambda: Prim(CYLINDER)
Cone = lambda: Prim(CONE)


def Triangle(a, b, c, normals=None, tex_coords=None) -> Prim:  # triangle.rs:8-26
    return Prim(TRIANGLE, tri=np.array([a, b, c], dtype=np.float64),
                tri_normals=None if normals is None else np.array(normals, dtype=np.float64),
                tri_tex_coords=None if tex_coords is None else np.array(tex_coords, dtype=np.float64))


def Mesh(data: MeshData, smooth: bool = False) -> Prim:  # mesh.rs:131-144
    return Prim(MESH, mesh=data, smooth=smooth)


def KDMesh(data: MeshData, smooth: bool = False) -> Prim:  # kdmesh.rs:32-58
    return Prim(KDMESH, mesh=data, smooth=smooth)


class Node:
    """scene.rs:36-206 SceneNode. Builder calls return self (Rust moves self through)."""

    def __init__(self, geometry: Optional[Tuple[Prim, Material]] = None, children: Optional[List["Node"]] = None):
        self.geometry = geometry
        self.children: List[Node] = list(children or [])
        self.ops: List[Tuple[str, Tuple[float, ...]]] = []

    @staticmethod
    def geo(prim: Prim, material: Material) -> "Node":  # From<Geometry>, scene.rs:51-58
        return Node(geometry=(prim, material))

    @staticmethod
    def group(children: Sequence["Node"]) -> "Node":  # From<Vec<Arc<SceneNode>>>, scene.rs:61-68
        return Node(children=list(children))

    def with_child(self, c: "Node") -> "Node":
        self.children.append(c)
        return self

    def with_children(self, cs: Sequence["Node"]) -> "Node":
        self.children.extend(cs)
        return self

    def scaled(self, v) -> "Node":
        self.ops.append(("s", _vec3(v)))
        return self

    def translated(self, v) -> "Node":
        self.ops.append(("t", _vec3(v)))
        return self

    def rotated_x(self, radians: float) -> "Node":
        self.ops.append(("x", (float(radians),)))
        return self

    def rotated_y(self, radians: float) -> "Node":
        self.ops.append(("y", (float(radians),)))
        return self

    def rotated_z(self, radians: float) -> "Node":
        self.ops.append(("z", (float(radians),)))
        return self

    def rotated_xzy(self, angles) -> "Node":  # scene.rs:177-180: x, then z, then y
        x, y, z = _vec3(angles)
        return self.rotated_x(x).rotated_z(z).rotated_y(y)


@dataclass
class Scene:  # scene.rs:14-18
    root: Node
    lights: List[Light]
    ambient: Sequence[float]


@dataclass
class Camera:  # camera.rs:5-14
    eye: Sequence[float]
    center: Sequence[float]
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fovy_degrees: float = 50.0

    @property
    def fovy_radians(self) -> float:
        return to_radians(self.fovy_degrees)


def default_background(width: int, height: int) -> np.ndarray:
    """The closure every example passes (e.g. examples/single-triangle.rs:56-57), evaluated at
    Uv{x/W, y/H} (render.rs:31-34). It only depends on v, so one colour per row: (H, 3)."""
    v = np.arange(height, dtype=np.float64) / float(height)
    a = np.array([0.2, 0.4, 0.6])
    b = np.array([0.0, 0.0, 1.0])  # Rgb::blue()
    return a[None, :] * (1.0 - v)[:, None] + b[None, :] * v[:, None]


@dataclass
class Linearised:
    """The scene DAG as flat arrays: the oracle's po_scene layout (oracle/portrayer_oracle.h)."""
    nodes: List[Node] = field(default_factory=list)
    ops: List[List[Tuple[str, Tuple[float, ...]]]] = field(default_factory=list)
    prim_type: List[int] = field(default_factory=list)
    prim_data: List[int] = field(default_factory=list)
    prim_flags: List[int] = field(default_factory=list)
    material: List[int] = field(default_factory=list)
    child_off: List[int] = field(default_factory=list)
    children: List[int] = field(default_factory=list)
    meshes: List[MeshData] = field(default_factory=list)
    triangles: List[Prim] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    textures: List[Texture] = field(default_factory=list)
    root: int = 0


def linearise(scene: Scene) -> Linearised:
    lin = Linearised()
    node_id, mesh_id, mat_id, tex_id = {}, {}, {}, {}
    order: List[Node] = []

    def visit(n: Node):
        if id(n) in node_id:
            return
        node_id[id(n)] = len(order)
        order.append(n)
        for c in n.children:
            visit(c)

    visit(scene.root)
    lin.nodes = order
    lin.root = node_id[id(scene.root)]
    lin.child_off = [0]
    for n in order:
        lin.ops.append(list(n.ops))
        if n.geometry is None:
            lin.prim_type.append(NONE); lin.prim_data.append(0); lin.prim_flags.append(0); lin.material.append(0)
        else:
            prim, mat = n.geometry
            if id(mat) not in mat_id:
                mat_id[id(mat)] = len(lin.materials)
                lin.materials.append(mat)
                for t in (mat.texture, mat.normals):
                    if t is not None and id(t) not in tex_id:
                        tex_id[id(t)] = len(lin.textures)
                        lin.textures.append(t)
            data = 0
            if prim.kind in (MESH, KDMESH):
                if id(prim.mesh) not in mesh_id:
                    mesh_id[id(prim.mesh)] = len(lin.meshes)
                    lin.meshes.append(prim.mesh)
                data = mesh_id[id(prim.mesh)]
            elif prim.kind == TRIANGLE:
                data = len(lin.triangles)
                lin.triangles.append(prim)
            lin.prim_type.append(prim.kind); lin.prim_data.append(data)
            lin.prim_flags.append(1 if prim.smooth else 0); lin.material.append(mat_id[id(mat)])
        for c in n.children:
            lin.children.append(node_id[id(c)])
        lin.child_off.append(len(lin.children))
    lin.texture_index = tex_id
    return lin
