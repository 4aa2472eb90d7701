"""CPU tests of the product's host layer (C++ library behind include/portrayer_host.h) against the
oracle and the test DSL: scene-script transliterations, builder-call composition, flattening,
bounding boxes, the k-d build, the camera, the OBJ reader, the PNG codec, and that both shared
libraries load and export every symbol their headers declare (no GPU needed)."""
import os
import sys
import re

import numpy as np
import pytest

import device_glue
import host_glue
from example_scenes import EXAMPLES as _BASELINE_EXAMPLES, MORE_EXAMPLES, big_scene

EXAMPLES = {**_BASELINE_EXAMPLES, **MORE_EXAMPLES}
from scene_dsl import ASSETS, MeshData

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(EXAMPLES)


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p[th]_[a-z0-9_]+)\s*\(", text)))


def test_libraries_export_every_declared_symbol():
    from portrayer_amd import _hip, host
    hip_fns = declared_functions("portrayer_hip.h")
    assert len(hip_fns) >= 18 and sorted(_hip.EXPORTS) == hip_fns
    for f in hip_fns:
        getattr(_hip.lib(), f)
    host_fns = [f for f in declared_functions("portrayer_host.h") if f.startswith("ph_")]
    assert sorted(host.EXPORTS) == host_fns
    for f in host_fns:
        getattr(host.lib(), f)
    import __graft_entry__
    assert _hip.lib().pt_abi_version() == __graft_entry__.header_abi_version() == 8


def test_driver_build_entry_point_passes():
    """__graft_entry__.build() is what the driver calls each round; nothing else in the suite did, and a stale ABI
    assertion in it once went unnoticed. force=False: an incremental make (the objects are up to date in a test run),
    then its checks - ABI version of the built library against the header, every declared symbol exported."""
    import __graft_entry__
    __graft_entry__.build(force=False)


def assert_same_scene(a, b):
    for k in ("node_trans", "prim_type", "prim_data", "prim_flags", "child_off", "mesh_vert_off", "mesh_tri_off", "ambient"):
        assert np.array_equal(np.asarray(a[k]).reshape(-1), np.asarray(b[k]).reshape(-1)), k
    n_ch = int(a["child_off"][-1])
    assert np.array_equal(a["children"][:n_ch], b["children"][:n_ch])
    for k, n in (("materials", "n_materials"), ("lights", "n_lights"), ("tri_vertices", "n_triangles"), ("tri_normals", "n_triangles")):
        assert int(a[n]) == int(b[n]), n
        assert np.array_equal(np.asarray(a[k])[:int(a[n])], np.asarray(b[k])[:int(a[n])]), k
    geo = a["prim_type"] >= 0
    assert np.array_equal(a["material"][geo], b["material"][geo])
    nv, nt = int(a["mesh_vert_off"][-1]), int(a["mesh_tri_off"][-1])
    assert np.array_equal(a["mesh_positions"][:nv], b["mesh_positions"][:nv])
    assert np.array_equal(a["mesh_indices"][:nt], b["mesh_indices"][:nt])
    assert np.array_equal(a["mesh_has_normals"][:int(a["n_meshes"])], b["mesh_has_normals"][:int(a["n_meshes"])])


@pytest.mark.parametrize("name", NAMES)
def test_cpp_example_equals_dsl_example(oracle, name):
    """examples/<name>.cpp (product) and tests/example_scenes.py (test DSL + oracle matrices) are two
    independent transliterations of the reference's scene script: every array must be identical."""
    from portrayer_amd import host
    cpp = host.Scene.example(name, assets=ASSETS)
    scene, cam, size = EXAMPLES[name]()
    ref, _ = oracle.arrays_from_dsl(scene)
    assert_same_scene(cpp.export(), ref)
    assert cpp.size == size
    assert np.array_equal(cpp.camera, host_glue.cam10(cam))


@pytest.mark.parametrize("name", ["big-mesh", "big-soup"])
def test_cpp_synthetic_scene_equals_dsl_scene(oracle, name):
    """bench.py takes the SURVEY 8(d) synthetic workloads from the product's C++ generator (examples/big-scene.cpp);
    the GPU parity tests take them from the test DSL. Same arrays (checked at n = 2: 8 cows)."""
    from example_scenes import SYNTHETIC
    from portrayer_amd import host
    cpp = host.Scene.example("synthetic:" + name, assets=ASSETS, n=2)
    scene, cam, size = SYNTHETIC[name](2)
    ref, _ = oracle.arrays_from_dsl(scene)
    assert_same_scene(cpp.export(), ref)
    assert np.array_equal(cpp.camera, host_glue.cam10(cam))


@pytest.mark.parametrize("name", NAMES)
def test_builder_call_replay_matches_oracle_composition(oracle, name):
    scene, _, _ = EXAMPLES[name]()
    hs = host_glue.host_scene(scene)
    ref, _ = oracle.arrays_from_dsl(scene)
    assert_same_scene(hs.export(), ref)


@pytest.mark.parametrize("name", NAMES)
def test_flatten_matches_oracle(oracle, name):
    from portrayer_amd import host
    cpp = host.Scene.example(name, assets=ASSETS)
    got = cpp.flatten()
    scene, _, _ = EXAMPLES[name]()
    ref = oracle.flatten(scene)
    for k in ("trans", "invtrans", "normal_trans", "prim_type", "bounds"):
        assert np.array_equal(got[k], ref[k]), k
    # material ids are numbered by first use (flat order here, DFS order in the oracle's input): same partition
    assert len(set(zip(got["material"], ref["material"]))) == len(set(got["material"])) == len(set(ref["material"]))


@pytest.mark.parametrize("name,depth", [("big-scene", 10), ("big-scene", 4), ("macho-cows", 10), ("primitives-simple", 10)])
def test_kdtree_matches_oracle(oracle, name, depth):
    from portrayer_amd import host
    cpp = host.Scene.example(name, assets=ASSETS)
    got = cpp.kdtree(depth)
    scene, _, _ = EXAMPLES[name]()
    ref = oracle.kd_scene_dump(scene, depth)
    assert np.array_equal(got["axis"], ref["axis"])
    split = ref["kind"] == 0
    assert np.array_equal(got["plane"][split], ref["plane"][split])
    assert np.array_equal(got["front"][split], ref["front"][split]) and np.array_equal(got["back"][split], ref["back"][split])
    assert np.array_equal(got["first"][~split], ref["first"][~split]) and np.array_equal(got["count"][~split], ref["count"][~split])
    assert np.array_equal(got["items"], ref["items"])
    assert np.array_equal(got["root_bounds"], ref["root_bounds"])
    assert got["max_depth"] <= depth


def test_reference_partition_known_answers_on_product_build(oracle):
    """leaf.rs:248-360 through the product's k-d build: planes at x = -8, 0, 3, 5, 8 do not stop at
    the midpoint; big-scene depth 10 has 1023 splits / 1024 leaves / 6806 references (SURVEY App.C)."""
    from portrayer_amd import host
    t = host.Scene.example("big-scene").kdtree(10)
    assert (t["axis"] >= 0).sum() == 1023 and (t["axis"] < 0).sum() == 1024 and len(t["items"]) == 6806


@pytest.mark.parametrize("name", NAMES)
def test_camera_matches_oracle_rays(oracle, name):
    """Camera::new on the host (camera.rs:34-45) feeds the kernel's ray_at: rebuild ray_at here from
    the product's pt_camera and compare with the oracle's primary rays bit for bit."""
    from portrayer_amd import host
    scene, cam, (w, h) = EXAMPLES[name]()
    pc = host.camera(host_glue.cam10(cam), w, h)
    ref = device_glue.camera_struct(cam, w, h)
    assert list(pc.view_to_world) == list(ref.view_to_world) and list(pc.eye) == list(ref.eye)
    assert (pc.fov_factor, pc.aspect_ratio, pc.width, pc.height) == (ref.fov_factor, ref.aspect_ratio, ref.width, ref.height)
    xy = np.array([[0.5, 0.5], [w - 0.5, h - 0.5], [w / 2.0, h / 3.0]])
    o, d = oracle.camera_rays(cam, w, h, xy)
    m = np.array(list(pc.view_to_world)).reshape(4, 4)
    for (x, y), oo, dd in zip(xy, o, d):
        vy = (1.0 - 2.0 * (y / pc.height)) * pc.fov_factor
        vx = (2.0 * (x / pc.width) - 1.0) * pc.aspect_ratio * pc.fov_factor
        world = [((m[r][0] * vx + m[r][1] * vy) + m[r][2] * -1.0) + m[r][3] for r in range(3)]
        diff = [world[k] - pc.eye[k] for k in range(3)]
        mag = float(np.sqrt((diff[0] * diff[0] + diff[1] * diff[1]) + diff[2] * diff[2]))
        assert [diff[k] / mag for k in range(3)] == list(dd) and list(pc.eye) == list(oo)


@pytest.mark.parametrize("obj", ["cow.obj", "plane.obj", "buckyball.obj", "monkey.obj", "castle.obj"])
def test_obj_loader_matches_python_reader(obj):
    from portrayer_amd import host
    pos, nrm, idx = host.load_obj(os.path.join(ASSETS, obj))
    ref = MeshData.load_obj(os.path.join(ASSETS, obj))
    assert np.array_equal(pos, ref.positions) and np.array_equal(idx, ref.triangles)
    assert (nrm is None) == (ref.normals is None)
    if nrm is not None:
        assert np.array_equal(nrm, ref.normals)
    assert {"cow.obj": 5804, "plane.obj": 2, "buckyball.obj": 116, "monkey.obj": 967}.get(obj, len(idx)) == len(idx)


def test_png_codec_roundtrip(tmp_path):
    from PIL import Image
    from portrayer_amd import host
    import ctypes as C
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    assert host.lib().ph_png_write(p.encode(), 53, 37, img.ctypes.data_as(host._u8p)) == 0
    assert np.array_equal(np.array(Image.open(p).convert("RGB")), img)
    q = str(tmp_path / "b.png")
    Image.fromarray(img).save(q)  # PIL picks its own filters: exercises every unfilter path
    size = np.zeros(2, dtype=np.uint32); out = np.zeros_like(img)
    assert host.lib().ph_png_read(q.encode(), size.ctypes.data_as(host._up), out.ctypes.data_as(host._u8p), out.size) == 0
    assert tuple(size) == (53, 37) and np.array_equal(out, img)
    golden = os.path.join(ROOT, "tests", "golden", "render", "01a_primitives-simple.png")
    g = np.array(Image.open(golden).convert("RGB"))
    out = np.zeros_like(g)
    assert host.lib().ph_png_read(golden.encode(), size.ctypes.data_as(host._up), out.ctypes.data_as(host._u8p), out.size) == 0
    assert np.array_equal(out, g)


def _python_paths(arr):
    """Independent of the C++ host: geometry-bearing paths of the hierarchy in breadth-first (FlatScene::from,
    flat_scene.rs:24-46) and in depth-first, node-before-children (SceneNode::ray_cast, scene.rs:95-117) order."""
    off, kids, ptype, root = arr["child_off"], arr["children"], arr["prim_type"], int(arr["root"])
    bfs, queue = [], [(root, (root,), ())]
    while queue:
        node, chain, path = queue.pop(0)
        if ptype[node] >= 0:
            bfs.append((chain, path))
        for k, c in enumerate(kids[off[node]:off[node + 1]]):
            queue.append((int(c), chain + (int(c),), path + (k,)))
    dfs = []

    def walk(node, path):
        if ptype[node] >= 0:
            dfs.append(path)
        for k, c in enumerate(kids[off[node]:off[node + 1]]):
            walk(int(c), path + (k,))

    walk(root, ())
    return bfs, {p: r for r, p in enumerate(dfs)}


@pytest.mark.parametrize("name", ["entering-the-mirror-dimension", "macho-cows", "hier", "instance", "water-glass", "random-3", "random-5"])
def test_scene_graph_packing_for_hierarchical_traversal(oracle, name):
    """pt_scene's ABI-4 arrays (PT_TRAVERSE_HIER) as the C++ host packs them, against a Python walk of the same
    hierarchy and the oracle's restated vek matrices: every flattened node's chain names exactly the SceneNodes on
    its path with their OWN matrices (not composed), inverses and normal matrices are the node's own, and the
    depth-first rank is the order scene.rs tests candidates in - including subtrees shared by two parents."""
    from example_scenes import TEXTURED_EXAMPLES
    if name.startswith("random-"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_render_parity import random_scene
        scene = random_scene(int(name.split("-")[1]))[0]
    else:
        scene = {**EXAMPLES, **TEXTURED_EXAMPLES}[name]()[0]
    arr, _ = oracle.arrays_from_dsl(scene)
    own = np.asarray(arr["node_trans"], dtype=np.float64).reshape(-1, 4, 4)
    bfs, rank = _python_paths(arr)
    g = host_glue.host_scene(scene).graph()
    assert len(bfs) == len(g["dfs_rank"]) > 0
    assert len(set(rank.values())) == len(bfs) and sorted(g["dfs_rank"].tolist()) == list(range(len(bfs)))
    for i, (chain, path) in enumerate(bfs):
        got = g["chain"][g["chain_off"][i]:g["chain_off"][i + 1]]
        assert len(got) == len(chain), (i, chain)
        for level, (gid, node) in enumerate(zip(got, chain)):
            assert np.array_equal(g["trans"][gid], own[node]), (i, level)
            assert np.array_equal(g["invtrans"][gid], oracle.mat4_inverse(own[node])), (i, level)
            assert np.array_equal(g["normal_trans"][gid], oracle.mat4_inverse(own[node]).T), (i, level)
        assert int(g["dfs_rank"][i]) == rank[path], (i, path)
    # a SceneNode reached along two paths (instancing) is ONE entry of the matrix table
    assert g["trans"].shape[0] <= len({n for chain, _ in bfs for n in chain})


def _image_read(path):
    from portrayer_amd import host
    size = np.zeros(2, dtype=np.uint32)
    assert host.lib().ph_image_read(path.encode(), size.ctypes.data_as(host._up), None, 0) == 0
    out = np.zeros((int(size[1]), int(size[0]), 3), dtype=np.uint8)
    assert host.lib().ph_image_read(path.encode(), size.ctypes.data_as(host._up), out.ctypes.data_as(host._u8p), out.size) == 0
    return out


@pytest.mark.parametrize("name", ["Brick_Wall_013_COLOR.jpg",             # baseline, 4:2:0, 1025 x 1025 (odd: partial MCUs, replicated chroma edges)
                                  "Rock_033_baseColor_2.jpg",             # baseline, 4:4:4
                                  "Terracotta_Tiles_002_Base_Color.jpg",  # progressive, 4:2:0
                                  "Terracotta_Tiles_002_Normal.jpg",      # progressive, 4:2:0
                                  "Stone_Wall_007_NORM_cubemap.jpg"])     # baseline, 4:4:4, 4096 x 3072
def test_jpeg_decoder_equals_libjpeg(name):
    """The C++ host's JPEG reader (portrayer_amd/host/jpeg.cpp) follows libjpeg's default decompression path
    (islow IDCT, fancy upsampling, fixed-point YCbCr tables): the texels it hands to the device are the ones
    Pillow hands to the oracle in the textured parity tests, bit for bit."""
    from PIL import Image
    path = os.path.join(ROOT, "tests", "golden", "assets", name)
    assert np.array_equal(_image_read(path), np.array(Image.open(path).convert("RGB")))


def test_jpeg_decoder_synthetic_variants(tmp_path):
    """Grey, 4:2:2, restart intervals, tiny and odd sizes, progressive with many scans - written by Pillow."""
    from PIL import Image
    rng = np.random.default_rng(3)
    base = np.clip(rng.normal(128, 50, (67, 45, 3)).cumsum(axis=1) / 6 % 256, 0, 255).astype(np.uint8)
    cases = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=1, progressive=True),
             dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=0, quality=30, optimize=True), dict(subsampling=2, quality=98, progressive=True, restart_marker_rows=1)]
    for i, kw in enumerate(cases):
        for img in (base, base[:9, :3], base[:16, :16], base[:1, :1]):
            p = str(tmp_path / f"v{i}.jpg")
            Image.fromarray(img).save(p, **{"quality": 85, **kw})
            assert np.array_equal(_image_read(p), np.array(Image.open(p).convert("RGB"))), (kw, img.shape)
        p = str(tmp_path / f"g{i}.jpg")
        Image.fromarray(base[:, :, 0]).save(p, quality=80, progressive=bool(kw.get("progressive")))
        assert np.array_equal(_image_read(p), np.array(Image.open(p).convert("RGB")))


def test_no_gpu_means_loud_failure():
    """Without an MI355X the product must raise, never fall back to a CPU path."""
    from portrayer_amd import _hip, host
    if _hip.lib().pt_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_hip.PortrayerHipError):
        _hip.Context(0)
    with pytest.raises(host.PortrayerHostError):
        host.Renderer(host.Scene.example("single-triangle"))
