"""The reference's own nine unit tests, restated against the oracle.

Each test names the reference test it restates (file:line). These pin the oracle where the
reference pins itself."""
import math
import os

import numpy as np
import pytest

from example_scenes import load_mesh
from scene_dsl import Camera, KDMesh, Light, Material, Mesh, Node, Plane, Scene, to_radians


def test_solve_quadratic_equations(oracle):  # src/math.rs:159-171
    got = oracle.quadratic(2.0, 8.0, 3.0)
    exp = [-2.0 - math.sqrt(5.0 / 2.0), math.sqrt(5.0 / 2.0) - 2.0]
    assert len(got) == 2 and all(abs(a - b) < 1e-6 for a, b in zip(got, exp))
    got = oracle.quadratic(4.0, -4.0, 1.0)
    assert len(got) == 1 and abs(got[0] - 0.5) < 1e-6
    assert oracle.quadratic(3.0, 4.0, 2.0) == []


def test_solution_order(oracle):  # src/math.rs:173-179
    got = oracle.quadratic(-2.0, 8.0, 3.0)
    exp = [2.0 - math.sqrt(11.0 / 2.0), 2.0 + math.sqrt(11.0 / 2.0)]
    assert len(got) == 2 and got[0] < got[1] and all(abs(a - b) < 1e-6 for a, b in zip(got, exp))


def test_rotated_plane_bounds_90(oracle):  # src/bounding_box.rs:171-182
    trans = oracle.compose([("x", (to_radians(90.0),))])
    mn, mx = oracle.transform_bounds(trans, (-0.5, 0.0, -0.5), (0.5, 0.0, 0.5))
    assert np.array_equal(np.round(mn * 10.0) / 10.0 + 0.0, [-0.5, -0.5, 0.0])
    assert np.array_equal(np.round(mx * 10.0) / 10.0 + 0.0, [0.5, 0.5, 0.0])


def test_rotated_cube_bounds_60(oracle):  # src/bounding_box.rs:184-195
    trans = oracle.compose([("s", (8.0, 0.25, 5.0)), ("x", (to_radians(60.0),))])
    mn, mx = oracle.transform_bounds(trans, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    assert np.array_equal(np.round(mn * 1000.0) / 1000.0, [-4.0, -2.228, -1.358])
    assert np.array_equal(np.round(mx * 1000.0) / 1000.0, [4.0, 2.228, 1.358])


def _edge_case_scene(sign):
    """node.rs:240-262 (sign=+1) and :302-324 (sign=-1): planes B and C."""
    mat_b, mat_c = Material(diffuse=(1, 0, 0)), Material(diffuse=(0, 0, 1))
    node_b = Node.geo(Plane(), mat_b).scaled(2.0).rotated_x(to_radians(sign * 90.0)).translated((0.0, 1.2, sign * -0.4))
    node_c = Node.geo(Plane(), mat_c).scaled(2.0).rotated_x(to_radians(sign * 50.0)).translated((0.0, 0.0, sign * -0.3))
    return Scene(root=Node.group([node_b, node_c]), lights=[], ambient=(0, 0, 0))  # flat ids: B = 0, C = 1


@pytest.mark.parametrize("flipped", [False, True])
def test_ray_cast_edge_case(oracle, flipped):  # src/kdtree/node.rs:219-293 and :295-351
    sign = -1.0 if flipped else 1.0
    scene = _edge_case_scene(sign)
    flat = oracle.flatten(scene)
    b = flat["bounds"]
    root_bounds = np.concatenate([np.minimum(b[0, :3], b[1, :3]), np.maximum(b[0, 3:], b[1, 3:])])
    # Split on z = 0; not flipped: front leaf [C], back leaf [C, B]; flipped: front [C, B], back [C]
    front_items, back_items = ([1], [1, 0]) if not flipped else ([1, 0], [1])
    tree = dict(kind=[0, 1, 1], axis=[2, -1, -1], plane=[0.0, 0.0, 0.0], front=[1, -1, -1], back=[2, -1, -1],
                first=[0, 0, len(front_items)], count=[0, len(front_items), len(back_items)], items=front_items + back_items)
    origin = [[0.0, 0.5, sign * 0.9]]
    direction = [[0.0, 0.0, sign * -1.0]]
    t, ids = oracle.kd_cast_custom(scene, tree, root_bounds, origin, direction)
    assert ids[0] == 0, "must return polygon B (nearer, only on the far side), not C"
    # brute force agrees
    t2, ids2, _, _ = oracle.cast_rays(scene, origin, direction, mode=oracle.MODE_FLAT)
    assert ids2[0] == 0 and t2[0] == t[0]


def _plane_boxes(oracle, xs):
    """leaf.rs:260-264: Plane rotated_z(90 deg) then translated to x."""
    mat = Material()
    scene = Scene(root=Node.group([Node.geo(Plane(), mat).rotated_z(to_radians(90.0)).translated((x, 0.0, 0.0)) for x in xs]),
                  lights=[], ambient=(0, 0, 0))
    return oracle.flatten(scene)["bounds"]


def _leaves(tree):
    out = {}
    for i, k in enumerate(tree["kind"]):
        if k == 1:
            out[i] = list(tree["items"][tree["first"][i]:tree["first"][i] + tree["count"][i]])
    return out


def test_single_axis_center_partition(oracle):  # src/kdtree/leaf.rs:248-300
    b = _plane_boxes(oracle, [-8.0, -5.0, 3.0, 5.0, 8.0])
    tree = oracle.kd_partition_boxes(b[:, :3], b[:, 3:], 5, target_max_nodes=3, target_max_merit=3, max_tries=10)
    assert list(tree["kind"]) == [0, 1, 1]
    assert tree["axis"][0] == 0 and tree["plane"][0] == 0.0
    leaves = _leaves(tree)
    assert leaves[tree["front"][0]] == [2, 3, 4]  # C, D, E
    assert leaves[tree["back"][0]] == [0, 1]      # A, B


def test_single_axis_uneven_partition(oracle):  # src/kdtree/leaf.rs:302-360
    b = _plane_boxes(oracle, [-8.0, 0.0, 3.0, 5.0, 8.0])
    tree = oracle.kd_partition_boxes(b[:, :3], b[:, 3:], 5, target_max_nodes=3, target_max_merit=2, max_tries=10)
    assert list(tree["kind"]) == [0, 1, 1]
    assert tree["axis"][0] == 0 and tree["plane"][0] == 4.0
    leaves = _leaves(tree)
    assert leaves[tree["front"][0]] == [3, 4]     # D, E
    assert leaves[tree["back"][0]] == [0, 1, 2]   # A, B, C


def test_mesh_equivalence(oracle):  # src/kdtree/kdmesh.rs:99-166
    model = load_mesh("castle.obj")
    mat = Material(diffuse=(1.0, 0.0, 0.0), specular=(0.3, 0.3, 0.3), shininess=25.0)
    lights = [Light(position=(50.0, 110.0, -120.0), color=(0.9, 0.9, 0.9))]

    def make(prim):
        return Scene(root=Node.geo(prim, mat).scaled(1.4).translated((0.0, 0.0, -229.0)), lights=lights, ambient=(0.3, 0.3, 0.3))

    cam = Camera(eye=(0.0, 120.0, 240.0), center=(0.0, 100.0, -24.0), fovy_degrees=25.0)
    width, height, n = 533.0, 300.0, 100000
    i = np.arange(n, dtype=np.float64)
    xy = np.stack([width * i / float(n), height * i / float(n)], axis=1)
    o, d = oracle.camera_rays(cam, width, height, xy)
    c_mesh = oracle.color_rays(make(Mesh(model)), o, d, mode=oracle.MODE_HIER)
    c_kd = oracle.color_rays(make(KDMesh(model)), o, d, mode=oracle.MODE_HIER)
    assert (c_mesh != 0).any(), "the diagonal must cross the castle"
    assert np.array_equal(c_mesh, c_kd)


def test_texel_edge_counters_of_sphere_texture_coordinates(oracle):
    """po_stats.tex_sphere_lookups / tex_sphere_near_edge (round 5): the GPU tests use them to PROVE that no sphere texture coordinate - the one place where the
    device's libm (atan2 / acos within 2 / 1 ulp of glibc's) could pick another texel than the reference - comes within 4096 ulps of a texel edge in the renders
    they compare exactly. Here: the counters count - a textured sphere fetches texels (none of them near an edge in this render), an untextured scene none."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_textures import textured_scene
    from example_scenes import EXAMPLES
    sc, cam = textured_scene(1)
    r = oracle.render(sc, cam, 96, 64, mode=oracle.MODE_FLAT, kd_depth=5)
    assert r.stats["tex_sphere_lookups"] > 100 and r.stats["tex_sphere_near_edge"] == 0
    scene, cam2, _ = EXAMPLES["primitives-simple"]()
    r2 = oracle.render(scene, cam2, 64, 48, mode=oracle.MODE_FLAT)
    assert r2.stats["tex_sphere_lookups"] == 0 and r2.stats["tex_sphere_near_edge"] == 0
