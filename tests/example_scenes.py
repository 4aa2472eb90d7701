"""The reference's example scene scripts re-expressed in the test DSL (test infrastructure).

Scene constants are taken from the reference's scene scripts (data, not code):
  examples/single-triangle.rs:17-58, examples/primitives-simple.rs:17-76,
  examples/macho-cows.rs:17-128, examples/entering-the-mirror-dimension.rs:17-188,
  examples/big-scene.rs:26-109, examples/smooth-shading.rs:17-100, examples/glossy-reflection.rs:17-87,
  examples/soft-shadows.rs:17-95, examples/hier.rs:17-101, examples/instance.rs:17-95, examples/antialiasing.rs:18-50, examples/fish.rs:17-63, examples/transmission-refraction.rs:20-264, examples/water-glass.rs:17-117.
The product has its own C++ transliteration of the same scripts (examples/*.cpp); the tests check
that both produce identical node matrices.
"""
from __future__ import annotations

import os

import numpy as np

from rand07 import StdRng
from scene_dsl import (ASSETS, Camera, Cone, Cube, Cylinder, Light, Material, Mesh, MeshData, Node, Plane, Scene, Sphere,
                       Triangle, to_radians)

_mesh_cache = {}


def load_mesh(name: str) -> MeshData:
    if name not in _mesh_cache:
        _mesh_cache[name] = MeshData.load_obj(os.path.join(ASSETS, name))
    return _mesh_cache[name]


def single_triangle():
    mat1 = Material(diffuse=(0.541, 0.169, 0.886), specular=(0.5, 0.7, 0.5), shininess=25.0)
    tri = Triangle((-1.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.5, 0.0))
    scene = Scene(root=Node.group([Node.geo(tri, mat1)]),
                  lights=[Light(position=(1.0, 1.0, 10.0), color=(0.5, 0.5, 0.5))],
                  ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 0.5, 4.0), center=(0.0, 0.5, 0.0), fovy_degrees=50.0)
    return scene, cam, (640, 480)


def primitives_simple():
    mat_grass = Material(diffuse=(0.173224, 0.8, 0.226505))
    mat_cylinder = Material(diffuse=(0.139339, 0.435762, 0.8), specular=(0.3, 0.3, 0.3), shininess=25.0)
    mat_cone = Material(diffuse=(0.8, 0.047361, 0.04305), specular=(0.3, 0.3, 0.3), shininess=25.0)
    scene = Scene(
        root=Node.group([
            Node.geo(Cylinder(), mat_cylinder).scaled(2.0).translated((-2.0, 1.0, 0.0)),
            Node.geo(Cone(), mat_cone).scaled(2.0).translated((2.0, 1.0, 0.0)),
            Node.geo(Plane(), mat_grass).scaled(10.0),
        ]),
        lights=[Light(position=(0.0, 10.0, 9.0), color=(0.9, 0.9, 0.9))],
        ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.760838, 8.095396, 10.50759), center=(-0.41716, -3.477774, -5.761218), fovy_degrees=25.0)
    return scene, cam, (910, 512)


def macho_cows():
    stone = Material(diffuse=(0.8, 0.7, 0.7))
    grass = Material(diffuse=(0.1, 0.7, 0.1))
    cow_hide = Material(diffuse=(0.84, 0.6, 0.53), specular=(0.3, 0.3, 0.3), shininess=20.0)
    cow_model, plane, buckyball = load_mesh("cow.obj"), load_mesh("plane.obj"), load_mesh("buckyball.obj")
    arc = Node.group([
        Node.geo(Cube(), stone).scaled((0.8, 4.0, 0.8)).translated((-2.0, 2.0, 0.0)),
        Node.geo(Cube(), stone).scaled((0.8, 4.0, 0.8)).translated((2.0, 2.0, 0.0)),
        Node.geo(Sphere(), stone).scaled((4.0, 0.6, 0.6)).translated((0.0, 4.0, 0.0)),
    ]).translated((0.0, 0.0, -10.0))
    nodes = [Node.group([arc]).rotated_y(to_radians(60.0 * float(i - 1))) for i in range(1, 7)]
    cow = Node.geo(Mesh(cow_model), cow_hide).translated((0.0, 3.637, 0.0)).scaled(2.0 / (2.76 + 3.637)).translated((0.0, -1.0, 0.0))
    cows = [((1.0, 1.3, 14.0), 20.0), ((5.0, 1.3, -11.0), 180.0), ((-5.5, 1.3, -3.0), -60.0)]
    for pos, rot in cows:
        nodes.append(Node.group([cow]).scaled(1.4).rotated_y(to_radians(rot)).translated(pos))
    nodes.append(Node.geo(Mesh(plane), grass).scaled(30.0))
    nodes.append(Node.geo(Mesh(buckyball), stone).scaled(1.5))
    scene = Scene(root=Node.group(nodes).rotated_x(to_radians(23.0)),
                  lights=[Light(position=(200.0, 202.0, 430.0), color=(0.8, 0.8, 0.8))],
                  ambient=(0.4, 0.4, 0.4))
    cam = Camera(eye=(0.0, 2.0, 30.0), center=(0.0, 2.0, 29.0), fovy_degrees=50.0)
    return scene, cam, (256, 256)


def mirror_dimension():
    mat_mirror_frame = Material(diffuse=(0.29, 0.204, 0.145), shininess=1.0)
    mat_mirror = Material(diffuse=(0.0, 0.0, 0.0), specular=(0.8, 0.8, 0.8), shininess=1000.0, reflectivity=1.0)
    mat_floor = Material(diffuse=(0.016, 0.384, 0.0), specular=(0.8, 0.8, 0.8), shininess=25.0)
    mat_body = Material(diffuse=(0.906, 0.22, 0.282), specular=(0.8, 0.8, 0.8), shininess=25.0)
    mat_head = Material(diffuse=(0.086, 0.671, 0.906), specular=(0.8, 0.8, 0.8), shininess=50.0)
    mat_eyes = Material(diffuse=(0.3, 0.3, 0.3), specular=(0.8, 0.8, 0.8), shininess=1000.0, reflectivity=0.9)
    mat_arms = Material(diffuse=(0.345, 0.588, 0.906), specular=(0.8, 0.8, 0.8), shininess=1.0)
    monkey, plane = load_mesh("monkey.obj"), load_mesh("plane.obj")
    deg = lambda v: tuple(to_radians(a) for a in v)
    mirror = Node.group([
        Node.geo(Cube(), mat_mirror_frame).scaled((3.96, 5.5, 0.4)).translated((0.0, 2.75, 0.0)),
        Node.geo(Cube(), mat_mirror).scaled((3.6, 5.0, 0.1)).translated((0.0, 2.75, 0.2)),
    ]).translated((0.0, 0.0, -1.3))
    monkey_character = Node.group([
        Node.geo(Cube(), mat_body).scaled((0.545055, 2.6, 0.545055)).translated((0.0, 1.3, 0.0)),
        Node.geo(Mesh(monkey), mat_head).scaled((1.0, 1.0, 1.0)).rotated_y(to_radians(180.0)).translated((0.0, 2.7, 0.0))
            .with_children([
                Node.geo(Sphere(), mat_eyes).scaled((0.1, 0.1, 0.05)).translated((0.35, 0.24, 0.8)),
                Node.geo(Sphere(), mat_eyes).scaled((0.1, 0.1, 0.05)).translated((-0.35, 0.24, 0.8)),
            ]),
        Node.geo(Sphere(), mat_arms).scaled((0.2, 0.63, 0.2)).rotated_xzy(deg((161.156, 107.062, -133.944))).translated((-0.388703, 1.715599, -0.2)),
        Node.geo(Sphere(), mat_arms).scaled((0.2, 0.56, 0.2)).rotated_xzy(deg((127.221, 42.0695, -104.823))).translated((-0.711297, 1.284401, -1.0)),
        Node.geo(Sphere(), mat_mirror).scaled((0.5, 0.5, 0.3)).translated((-0.711297, 1.284401, -1.20)),
        Node.geo(Sphere(), mat_arms).scaled((0.2, 0.63, 0.2)).rotated_xzy(deg((92.3684, -57.6199, 38.2278))).translated((0.581161, 1.984976, -0.2)),
        Node.geo(Sphere(), mat_arms).scaled((0.2, 0.56, 0.2)).rotated_xzy(deg((91.5166, -11.239, 28.419))).translated((1.118839, 2.015024, -1.0)),
        Node.geo(Sphere(), mat_mirror).scaled((0.5, 0.5, 0.3)).translated((1.118839, 2.015024, -1.20)),
    ])
    floor = Node.geo(Mesh(plane), mat_floor).scaled(20.0)
    scene = Scene(root=Node.group([mirror, floor, monkey_character]),
                  lights=[Light(position=(2.5, 3.5, -1.0), color=(0.9, 0.9, 0.9)),
                          Light(position=(10.0, 10.0, 0.0), color=(0.9, 0.9, 0.9)),
                          Light(position=(-9.0, 4.0, 0.0), color=(0.406471, 0.901283, 1.0))],
                  ambient=(0.2, 0.2, 0.2))
    cam = Camera(eye=(5.545485, 2.966984, 1.795613), center=(-4.348584, 2.148794, -3.057839), fovy_degrees=30.0)
    return scene, cam, (800, 600)


def big_scene(n: int = 10, prims=None):
    """examples/big-scene.rs:26-109; `prims` replaces the primitive list (synthetic variants)."""
    rng = StdRng.seed_from_u64(1234939301)
    materials = []
    for _ in range(15):
        r = rng.gen_f64(); g = rng.gen_f64(); b = rng.gen_f64()
        materials.append(Material(diffuse=(r, g, b), specular=(0.3, 0.3, 0.3), shininess=25.0))
    primitives = prims if prims is not None else [Sphere, Cube, Cone, Cylinder]
    width = length = height = 800.0
    nodes = []
    for i in range(n):
        x = float(i) / float(n - 1) * width - width / 2.0
        for j in range(n):
            y = float(j) / float(n - 1) * length - length / 2.0
            for k in range(n):
                z = float(k) / float(n - 1) * height - height / 2.0
                prim = rng.choose(primitives)
                mat = rng.choose(materials)
                scale = 30.0 * rng.gen_f64() + 30.0
                angle = to_radians(360.0 * rng.gen_f64())
                yy = y + rng.gen_f64() * 50.0
                nodes.append(Node.geo(prim(), mat).scaled(scale).rotated_xzy((angle, angle, angle)).translated((x, yy, z)))
    scene = Scene(root=Node.group(nodes),
                  lights=[Light(position=(-100.0, 150.0, 400.0), color=(0.9, 0.9, 0.9)),
                          Light(position=(100.0, -150.0, 800.0), color=(0.7, 0.7, 0.7)),
                          Light(position=(400.0, 100.0, 150.0), color=(0.7, 0.0, 0.7))],
                  ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 0.0, 1200.0), center=(0.0, 0.0, 0.0), fovy_degrees=50.0)
    return scene, cam, (1980, 1020)


def smooth_shading():
    mat_rock = Material(diffuse=(0.256361, 0.256361, 0.256361), specular=(0.6, 0.6, 0.6), shininess=50.0)
    mat_cow = Material(diffuse=(0.692066, 0.477245, 0.293336), specular=(0.3, 0.3, 0.3), shininess=25.0)
    mat_monkey = Material(diffuse=(0.261829, 0.8, 0.310477), specular=(0.3, 0.3, 0.3), shininess=25.0)
    monkey, cow, flat_rock, smooth_rock = (load_mesh(f) for f in ("monkey.obj", "cow.obj", "flat_rock.obj", "smooth_rock.obj"))
    scene = Scene(
        root=Node.group([
            Node.geo(Mesh(monkey), mat_monkey).rotated_y(to_radians(45.0)).translated((-1.904434, 1.4, 0.0)),
            Node.geo(Mesh(cow), mat_cow).scaled(0.5).rotated_y(to_radians(-15.0)).translated((-4.2, 1.8, 4.0)),
            Node.geo(Mesh(flat_rock), mat_rock).translated((-3.396987, -1.4, 2.286671)),
            Node.geo(Mesh(monkey, smooth=True), mat_monkey).rotated_y(to_radians(-45.0)).translated((1.242585, 1.4, 0.0)),
            Node.geo(Mesh(cow, smooth=True), mat_cow).scaled(0.5).rotated_y(to_radians(205.0)).translated((3.8, 1.8, 4.0)),
            Node.geo(Mesh(smooth_rock, smooth=True), mat_rock).translated((3.271008, -1.406423, 2.372513)),
        ]),
        lights=[Light(position=(0.0, 5.0, 10.0), color=(0.9, 0.9, 0.9))],
        ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(1.062382, 0.54746, 22.827951), center=(-0.813817, 0.424462, -8.112782), fovy_degrees=24.0)
    return scene, cam, (910, 512)


def glossy_reflection():
    non_glossy_ball = Material(diffuse=(0.146505, 0.314666, 0.170564), specular=(0.3, 0.3, 0.3), shininess=100.0, reflectivity=0.4)
    glossy_ball = Material(diffuse=(0.146505, 0.314666, 0.170564), specular=(0.3, 0.3, 0.3), shininess=100.0, reflectivity=0.4,
                           glossy_side_length=2.0)
    center_ball = Material(diffuse=(0.8, 0.0, 0.023362), specular=(0.3, 0.3, 0.3), shininess=25.0)
    table = Material(diffuse=(1.0, 0.6, 0.1), specular=(0.3, 0.3, 0.3), shininess=25.0)
    scene = Scene(
        root=Node.group([
            Node.geo(Sphere(), non_glossy_ball).translated((-1.1, 1.3, 0.0)),
            Node.geo(Sphere(), glossy_ball).translated((1.1, 1.3, 0.0)),
            Node.geo(Sphere(), center_ball).scaled(0.5).translated((0.0, 0.8, 1.8)),
            Node.geo(Cube(), table).scaled((10.0, 0.6, 5.0)),
        ]),
        lights=[Light(position=(0.0, 6.0, 3.0), color=(0.9, 0.9, 0.9)), Light(position=(0.0, 1.0, 12.0), color=(0.7, 0.7, 0.7))],
        ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 2.562834, 8.863271), center=(0.0, -1.083779, -11.817695), fovy_degrees=20.0)
    return scene, cam, (910, 512)


def soft_shadows():
    mat_cow = Material(diffuse=(0.37168, 0.236767, 0.692066), specular=(0.3, 0.3, 0.3), shininess=25.0)
    mat_wall_floor = Material(diffuse=(0.627459, 0.8, 0.589836), specular=(0.3, 0.3, 0.3), shininess=25.0)
    cow = load_mesh("cow.obj")
    scene = Scene(
        root=Node.group([
            Node.geo(Plane(), mat_wall_floor).scaled(30.0),
            Node.geo(Cube(), mat_wall_floor).scaled((0.2, 20.0, 20.0)).translated((0.0, 8.0, 8.0)),
            Node.geo(Cube(), mat_wall_floor).scaled((30.0, 30.0, 0.4)).translated((0.0, 8.0, -2.0)),
            Node.geo(Mesh(cow, smooth=True), mat_cow).scaled(0.5).rotated_y(to_radians(-15.0)).translated((-4.2, 1.8, 4.0)),
            Node.geo(Mesh(cow, smooth=True), mat_cow).scaled(0.5).rotated_y(to_radians(195.0)).translated((4.2, 1.8, 4.0)),
        ]),
        lights=[Light(position=(-2.0, 2.0, 16.0), color=(0.5, 0.5, 0.5)),
                Light(position=(2.0, 2.0, 16.0), color=(0.5, 0.5, 0.5), area_a=(0.0, 0.5, 0.0), area_b=(0.5, 0.0, 0.0))],
        ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 5.04746, 24.827951), center=(0.012231, -0.459716, -15.800501), fovy_degrees=25.0)
    return scene, cam, (910, 512)


def _arch(mat):
    return [Node.geo(Cube(), mat).scaled((0.8, 4.0, 0.8)).translated((-2.0, 2.0, 0.0)),
            Node.geo(Cube(), mat).scaled((0.8, 4.0, 0.8)).translated((2.0, 2.0, 0.0)),
            Node.geo(Sphere(), mat).scaled((4.0, 0.6, 0.6)).translated((0.0, 4.0, 0.0))]


def hier():
    gold = Material(diffuse=(0.9, 0.8, 0.4), specular=(0.8, 0.8, 0.4), shininess=25.0)
    grass = Material(diffuse=(0.1, 0.7, 0.1))
    blue = Material(diffuse=(0.7, 0.6, 1.0), specular=(0.5, 0.4, 0.8), shininess=25.0)
    arc = Node.group(_arch(gold)).translated((0.0, 0.0, -10.0)).rotated_y(to_radians(60.0))
    floor = Node.geo(Mesh(load_mesh("plane.obj")), grass).scaled(30.0)
    poly = Node.geo(Mesh(load_mesh("dodeca.obj")), blue).translated((-2.0, 1.618034, 0.0))
    scene = Scene(root=Node.group([arc, floor, poly]).rotated_x(to_radians(23.0)).translated((6.0, -2.0, -15.0)),
                  lights=[Light(position=(200.0, 200.0, 400.0), color=(0.8, 0.8, 0.8)), Light(position=(0.0, 5.0, -20.0), color=(0.4, 0.4, 0.8))],
                  ambient=(0.4, 0.4, 0.4))
    cam = Camera(eye=(0.0, 0.0, 0.0), center=(0.0, 0.0, -1.0), fovy_degrees=50.0)
    return scene, cam, (256, 256)


def instance():
    stone = Material(diffuse=(0.8, 0.7, 0.7))
    grass = Material(diffuse=(0.1, 0.7, 0.1))
    arc = Node.group(_arch(stone)).translated((0.0, 0.0, -10.0))
    nodes = [Node.group([arc]).rotated_y(to_radians(60.0 * float(i))) for i in range(1, 7)]
    nodes.append(Node.geo(Mesh(load_mesh("plane.obj")), grass).scaled(30.0))
    nodes.append(Node.geo(Sphere(), stone).scaled(2.5))
    scene = Scene(root=Node.group(nodes).rotated_x(to_radians(23.0)),
                  lights=[Light(position=(200.0, 202.0, 430.0), color=(0.8, 0.8, 0.8))],
                  ambient=(0.4, 0.4, 0.4))
    cam = Camera(eye=(0.0, 2.0, 30.0), center=(0.0, 2.0, 29.0), fovy_degrees=50.0)
    return scene, cam, (256, 256)


def antialiasing():
    mat_monkey = Material(diffuse=(0.961, 0.573, 0.259), specular=(0.3, 0.3, 0.3), shininess=25.0)
    scene = Scene(root=Node.group([Node.geo(Mesh(load_mesh("monkey.obj")), mat_monkey)]),
                  lights=[Light(position=(0.0, 0.0, 10.0), color=(0.5, 0.5, 0.5))],
                  ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 0.0, 6.5), center=(0.0, 0.0, 0.0), fovy_degrees=20.0)
    return scene, cam, (300, 250)


# more reference scene scripts (not in BASELINE.json's configs): pins for smooth shading, glossy reflection, area lights
MORE_EXAMPLES = {"smooth-shading": smooth_shading, "glossy-reflection": glossy_reflection, "soft-shadows": soft_shadows,
                 "hier": hier, "instance": instance, "antialiasing": antialiasing}


EXAMPLES = {
    "single-triangle": single_triangle,
    "primitives-simple": primitives_simple,
    "macho-cows": macho_cows,
    "entering-the-mirror-dimension": mirror_dimension,
    "big-scene": big_scene,
}


def normal_mapping(light_pos=(0.0, 8.0, 10.0)):
    """examples/normal-mapping.rs:20-171 (texture + normal maps on Plane, Cube, Sphere)."""
    import os
    from scene_dsl import Texture
    tex = lambda name: Texture.open(os.path.join(ASSETS, name))
    tex_plane, nrm_plane = tex("Terracotta_Tiles_002_Base_Color.jpg"), tex("Terracotta_Tiles_002_Normal.jpg")
    tex_sphere, nrm_sphere = tex("Rock_033_baseColor_2.jpg"), tex("Rock_033_normal_2.jpg")
    tex_cube, nrm_cube = tex("Stone_Wall_007_COLOR_cubemap.jpg"), tex("Stone_Wall_007_NORM_cubemap.jpg")
    d = (0.37168, 0.236767, 0.692066)
    mat_tex_plane = Material(diffuse=d, specular=(0.4, 0.4, 0.4), shininess=25.0, texture=tex_plane)
    mat_tex_plane_norm = Material(diffuse=d, specular=(0.4, 0.4, 0.4), shininess=25.0, texture=tex_plane, normals=nrm_plane)
    mat_tex_sphere = Material(diffuse=d, specular=(0.6, 0.6, 0.6), shininess=25.0, texture=tex_sphere)
    mat_tex_sphere_norm = Material(diffuse=d, specular=(0.6, 0.6, 0.6), shininess=25.0, texture=tex_sphere, normals=nrm_sphere)
    mat_tex_cube = Material(diffuse=d, specular=(0.3, 0.3, 0.3), shininess=25.0, texture=tex_cube)
    mat_tex_cube_norm = Material(diffuse=d, specular=(0.3, 0.3, 0.3), shininess=25.0, texture=tex_cube, normals=nrm_cube)
    mat_wall_floor = Material(diffuse=(0.424858, 0.531206, 0.8), specular=(0.3, 0.3, 0.3), shininess=25.0)
    root = Node.group([
        Node.geo(Plane(), mat_wall_floor).scaled(40.0).translated((0.0, -1.0, 0.0)),
        Node.geo(Plane(), mat_tex_plane).scaled(6.0).rotated_x(to_radians(90.0)).translated((-4.0, 2.0, -6.0)),
        Node.geo(Cube(), mat_tex_cube).scaled(2.0).translated((-7.0, 0.0, -1.0)),
        Node.geo(Sphere(), mat_tex_sphere).translated((-7.0, 2.0, -1.0)),
        Node.geo(Cube(), mat_tex_cube).scaled(2.0).translated((-2.0, 0.0, 3.0)),
        Node.geo(Sphere(), mat_tex_sphere).translated((-2.0, 2.0, 3.0)),
        Node.geo(Plane(), mat_tex_plane_norm).scaled(6.0).rotated_x(to_radians(90.0)).translated((4.0, 2.0, -6.0)),
        Node.geo(Cube(), mat_tex_cube_norm).scaled(2.0).translated((7.0, 0.0, -1.0)),
        Node.geo(Sphere(), mat_tex_sphere_norm).translated((7.0, 2.0, -1.0)),
        Node.geo(Cube(), mat_tex_cube_norm).scaled(2.0).translated((2.0, 0.0, 3.0)),
        Node.geo(Sphere(), mat_tex_sphere_norm).translated((2.0, 2.0, 3.0)),
    ])
    scene = Scene(root=root, lights=[Light(position=light_pos, color=(0.9, 0.9, 0.9))], ambient=(0.2, 0.2, 0.2))
    cam = Camera(eye=(0.0, 8.07551, 23.078941), center=(0.0, -2.854475, -16.437334), fovy_degrees=22.0)
    return scene, cam, (910, 512)


def fish():
    from scene_dsl import Texture
    mat_fish = Material(diffuse=(0.8, 0.8, 0.8), specular=(0.3, 0.3, 0.3), shininess=25.0, texture=Texture.open(os.path.join(ASSETS, "fish.png")))
    model = load_mesh("fish.obj")
    scene = Scene(root=Node.group([Node.geo(Mesh(model, smooth=True), mat_fish).rotated_y(to_radians(30.0)),
                                   Node.geo(Mesh(model, smooth=True), mat_fish).rotated_y(to_radians(210.0))]),
                  lights=[Light(position=(0.0, 0.0, 10.0), color=(0.9, 0.9, 0.9))],
                  ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 0.0, 11.0), center=(0.0, 0.0, 0.0), fovy_degrees=25.0)
    return scene, cam, (910, 512)


def transmission_refraction():
    """examples/transmission-refraction.rs:20-264: a glass pane in front of a tiled water tank with two textured
    KDMesh fish, a wooden table (texture + normal map), a glass of water with a straw."""
    from scene_dsl import KDMesh, Texture
    tex = lambda name: Texture.open(os.path.join(ASSETS, name))
    WINDOW_GLASS, WATER = 1.51, 1.33  # material.rs:12-16
    mat_glass = Material(diffuse=(0.0, 0.0, 0.0), specular=(0.3, 0.3, 0.3), shininess=25.0, reflectivity=1.0, refraction_index=WINDOW_GLASS)
    # room()
    mat_walls = Material(diffuse=(0.607917, 0.8, 0.551884), specular=(0.3, 0.3, 0.3), shininess=25.0)
    mat_table = Material(specular=(0.5, 0.5, 0.5), shininess=100.0, texture=tex("Wood_018_basecolor_cubemap.jpg"), normals=tex("Wood_018_normal_cubemap.jpg"))
    room = Node.group([
        Node.geo(Cube(), mat_table).scaled((20.0, 5.0, 2.5)).translated((0.0, -2.0, 1.3)),
        Node.geo(Plane(), mat_walls).scaled((20.0, 1.0, 20.0)).rotated_x(to_radians(90.0)).translated((0.0, 3.0, -10.0)),
        Node.geo(Plane(), mat_walls).scaled((20.0, 1.0, 12.0)).rotated_z(to_radians(90.0)).translated((10.0, 3.0, -6.0)),
        Node.geo(Plane(), mat_walls).scaled((20.0, 1.0, 12.0)).rotated_z(to_radians(-90.0)).translated((-10.0, 3.0, -6.0)),
        Node.geo(Plane(), mat_walls).scaled((12.1, 1.0, 20.0)).rotated_x(to_radians(90.0)).translated((16.0, 3.0, 0.0)),
        Node.geo(Plane(), mat_walls).scaled((12.1, 1.0, 20.0)).rotated_x(to_radians(90.0)).translated((-16.0, 3.0, 0.0)),
    ])
    # tank()
    mat_tank = Material(specular=(0.5, 0.5, 0.5), shininess=100.0, texture=tex("Tiles_017_basecolor_cubemap.jpg"), normals=tex("Tiles_017_normal_cubemap.jpg"))
    nodes = []
    for i in range(4):
        nodes.append(Node.geo(Cube(), mat_tank).scaled((5.0, 5.0, 0.2)).translated((float(i) * 5.0 - 7.5, -2.0, -10.0)))
        nodes.append(Node.geo(Cube(), mat_tank).scaled((5.0, 5.0, 0.2)).translated((float(i) * 5.0 - 7.5, -2.0, 0.0)))
    for i in range(2):
        nodes.append(Node.geo(Cube(), mat_tank).scaled((0.2, 5.0, 5.0)).translated((-10.0, -2.0, -(float(i) * 5.0 + 2.5))))
        nodes.append(Node.geo(Cube(), mat_tank).scaled((0.2, 5.0, 5.0)).translated((10.0, -2.0, -(float(i) * 5.0 + 2.5))))
    for x in range(4):
        for y in range(2):
            nodes.append(Node.geo(Cube(), mat_tank).scaled((5.0, 0.2, 5.0)).translated((float(x) * 5.0 - 7.5, -4.0, -(float(y) * 5.0 + 2.5))))
    tank = Node.group(nodes)
    # water()
    mat_water = Material(diffuse=(0.0, 0.0, 0.1), specular=(0.3, 0.3, 0.3), shininess=25.0, reflectivity=0.9, refraction_index=WATER)
    mat_fish = Material(diffuse=(0.8, 0.8, 0.8), specular=(0.3, 0.3, 0.3), shininess=25.0, texture=tex("fish.png"))
    fish_model = load_mesh("fish.obj")
    deg = lambda v: tuple(to_radians(a) for a in v)
    water = Node.group([
        Node.geo(Cube(), mat_water).scaled((19.799999, 3.8, 9.8)).translated((0.0, -2.0, -5.0)),
        Node.geo(KDMesh(fish_model, True), mat_fish).rotated_xzy(deg((0.0, -71.8181, 30.8927))).translated((-4.798946, -0.970323, -5.246493)),
        Node.geo(KDMesh(fish_model, True), mat_fish).rotated_xzy(deg((0.0, 108.666, -23.084))).translated((3.110451, -2.562474, -6.838645)),
    ])
    # drink()
    mat_water2 = Material(diffuse=(0.0, 0.0, 0.1), specular=(0.3, 0.3, 0.3), shininess=25.0, reflectivity=0.9, refraction_index=WATER)
    mat_straw = Material(diffuse=(0.8, 0.0, 0.0), specular=(0.3, 0.3, 0.3), shininess=25.0)
    drink = Node.group([
        Node.geo(Cylinder(), mat_water2).scaled((1.0, 1.4, 1.0)).translated((-7.4, 1.2, 1.2)),
        Node.geo(Cylinder(), mat_straw).scaled((0.1, 2.0, 0.1)).rotated_z(to_radians(28.4282)).translated((-7.565556, 1.411109, 1.1)),
    ])
    front_glass = Node.geo(Cube(), mat_glass).scaled((20.0, 10.0, 0.2)).translated((0.0, 5.0, 0.0))
    scene = Scene(root=Node.group([front_glass, room, tank, water, drink]),
                  lights=[Light(position=(0.0, 27.0, 5.0), color=(0.5, 0.5, 0.5))],
                  ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 14.658033, 27.19817), center=(0.0, -6.058867, -24.828854), fovy_degrees=23.0)
    return scene, cam, (910, 512)


def water_glass():
    """examples/water-glass.rs:17-117."""
    from scene_dsl import Texture
    tex = lambda name: Texture.open(os.path.join(ASSETS, name))
    mat_wall = Material(specular=(0.3, 0.3, 0.3), shininess=25.0, texture=tex("Brick_Wall_013_COLOR.jpg"), normals=tex("Brick_Wall_013_NORM.jpg"))
    mat_table = Material(specular=(0.5, 0.5, 0.5), shininess=100.0, reflectivity=0.2, glossy_side_length=2.0,
                         texture=tex("Wood_018_basecolor_cubemap.jpg"), normals=tex("Wood_018_normal_cubemap.jpg"))
    room = Node.group([
        Node.geo(Plane(), mat_wall).scaled(10.0).rotated_x(to_radians(90.0)).translated((0.0, 1.0, -2.0)),
        Node.geo(Cube(), mat_table).scaled((8.0, 0.4, 4.0)).translated((0.0, 0.0, -0.2)),
    ])
    mat_water = Material(diffuse=(0.0, 0.0, 0.1), specular=(0.3, 0.3, 0.3), shininess=25.0, reflectivity=0.9, refraction_index=1.33)
    mat_straw = Material(diffuse=(0.8, 0.0, 0.0), specular=(0.3, 0.3, 0.3), shininess=25.0)
    drink = Node.group([
        Node.geo(Cylinder(), mat_water).scaled((1.0, 1.4, 1.0)).translated((0.0, 0.7, 0.0)),
        Node.geo(Cylinder(), mat_straw).scaled((0.1, 2.0, 0.1)).rotated_z(to_radians(28.4282)).translated((-0.165556, 0.911109, 0.1)),
    ]).translated((0.0, 0.2, 0.0))
    scene = Scene(root=Node.group([room, drink]), lights=[Light(position=(0.0, 27.0, 5.0), color=(0.5, 0.5, 0.5))], ambient=(0.3, 0.3, 0.3))
    cam = Camera(eye=(0.0, 3.2, 7.151111), center=(0.0, 0.091525, -5.719519), fovy_degrees=23.0)
    return scene, cam, (910, 512)


TEXTURED_EXAMPLES = {"normal-mapping": normal_mapping, "fish": fish, "transmission-refraction": transmission_refraction, "water-glass": water_glass}


def big_mesh(n: int = 6):
    """SURVEY §8(d) synthetic variant "big-mesh-N": the big-scene generator with the primitive list
    replaced by Mesh(cow.obj): n = 6 -> 216 instances = 1,253,664 instanced triangles (one 5,804-triangle
    mesh, 216 transforms). Not a reference scene."""
    cow = load_mesh("cow.obj")
    scene, cam, size = big_scene(n, prims=[lambda: Mesh(cow)])
    # cow.obj spans about 5 units; the generator scales by 30..60, far larger than the grid pitch: shrink
    for node in scene.root.children:
        node.ops[0] = ("s", tuple(v / 4.0 for v in node.ops[0][1]))
    return scene, cam, size


def big_soup(n: int = 6):
    """SURVEY §8(d) synthetic variant "big-soup": the same instances baked to ONE world-space triangle
    mesh (n = 6: 1,253,664 triangles, 90 MB of vertex records) — the input where the scene no longer fits
    the caches. Not a reference scene."""
    import oracle_lib as O
    scene, cam, size = big_mesh(n)
    cow = load_mesh("cow.obj")
    pos, tris = [], []
    for k, node in enumerate(scene.root.children):
        m = O.compose(node.ops)
        p = cow.positions
        w = np.stack([((m[r, 0] * p[:, 0] + m[r, 1] * p[:, 1]) + m[r, 2] * p[:, 2]) + m[r, 3] for r in range(3)], axis=1)
        pos.append(w)
        tris.append(cow.triangles + np.uint32(k * len(p)))
    soup = MeshData(np.concatenate(pos), np.concatenate(tris).astype(np.uint32), None, "soup")
    mat = Material(diffuse=(0.7, 0.6, 0.5), specular=(0.3, 0.3, 0.3), shininess=25.0)
    return Scene(root=Node.group([Node.geo(Mesh(soup), mat)]), lights=scene.lights, ambient=scene.ambient), cam, size


SYNTHETIC = {"big-mesh": big_mesh, "big-soup": big_soup}
