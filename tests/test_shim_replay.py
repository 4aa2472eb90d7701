"""The Rust shim (shim/: build.rs, src/hip_ffi.rs, src/hip_pack.rs, overlay/*.rs.append, apply.py) cannot be compiled
here (no rustc / cargo). What CAN be checked:

  CPU  shim/apply.py overlays a copy of the reference checkout without touching anything but the places it names, and
       the structs of shim/src/hip_ffi.rs match include/portrayer_hip.h field for field;
  GPU  tests/shim_replay.c - the shim's call sequence and array layouts in plain C (triangle-list meshes, materials by
       first use, breadth-first nodes, the scene-graph arrays of an instanced subtree) - renders through
       libportrayer_hip.so exactly what the oracle renders from the same scene description."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"


@pytest.mark.parametrize("c_name,rust_name", [("pt_scene", "PtScene"), ("pt_kdtree", "PtKdTree"), ("pt_camera", "PtCamera"), ("pt_rect", "PtRect"),
                                             ("pt_render_params", "PtRenderParams"), ("pt_stats", "PtStats")])
def test_hip_ffi_structs_match_the_header(c_name, rust_name):
    """Same fields, same types, same order (the parsers of test_integration_doc.py, which checks INTEGRATION.md the same way)."""
    from test_integration_doc import c_structs, rust_structs
    header = open(os.path.join(ROOT, "include", "portrayer_hip.h")).read()
    ffi = open(os.path.join(ROOT, "shim", "src", "hip_ffi.rs")).read()
    c, r = c_structs(header), rust_structs(ffi)
    assert r[rust_name] == c[c_name], f"{rust_name} differs from {c_name}:\n rust {r[rust_name]}\n c    {c[c_name]}"


def test_hip_ffi_functions_exist_in_the_header():
    from test_integration_doc import strip_comments
    header = strip_comments(open(os.path.join(ROOT, "include", "portrayer_hip.h")).read())
    ffi = open(os.path.join(ROOT, "shim", "src", "hip_ffi.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', ffi, flags=re.S).group(1)
    fns = re.findall(r"pub fn (\w+)\((.*?)\)", re.sub(r"//[^\n]*", "", block), flags=re.S)
    assert len(fns) >= 12
    for name, args in fns:
        m = re.search(r"\b" + name + r"\((.*?)\);", header, flags=re.S)
        assert m, f"{name} is not declared in portrayer_hip.h"
        n_c = 0 if m.group(1).strip() == "void" else m.group(1).count(",") + 1
        assert args.count(":") == n_c, f"{name}: {args.count(':')} arguments in hip_ffi.rs, {n_c} in the header"
    assert int(re.search(r"PT_ABI_VERSION: c_int = (\d+)", ffi).group(1)) == int(re.search(r"#define PT_ABI_VERSION (\d+)", header).group(1))


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout only exists in the build container")
def test_apply_overlays_a_reference_checkout(tmp_path):
    dst = str(tmp_path / "portrayer-hip")
    subprocess.run([sys.executable, os.path.join(ROOT, "shim", "apply.py"), REFERENCE, dst], check=True)
    render = open(os.path.join(dst, "src", "render.rs")).read()
    ref_render = open(os.path.join(REFERENCE, "src", "render.rs")).read()
    # exactly one delegation inserted, everything else of the reference's file intact, our impl appended
    inserted = '        #[cfg(feature = "hip")]\n        { return self.render_hip::<R, T>(scene, camera, background); }\n'
    assert render.count(inserted) == 1
    assert render.replace(inserted, "").startswith(ref_render)
    assert "fn render_hip<" in render[len(ref_render):]
    for f in ("src/primitive/mesh.rs", "src/kdtree/kdmesh.rs", "src/kdtree/node.rs", "src/bounding_box.rs", "src/camera.rs", "src/texture.rs"):
        new, old = open(os.path.join(dst, f)).read(), open(os.path.join(REFERENCE, f)).read()
        assert new.startswith(old) and "hip_" in new[len(old):], f
    assert "mod hip_pack;" in open(os.path.join(dst, "src", "lib.rs")).read()
    cargo = open(os.path.join(dst, "Cargo.toml")).read()
    assert "hip = []" in cargo and 'build = "build.rs"' in cargo
    for f in ("build.rs", "src/hip_ffi.rs", "src/hip_pack.rs"):
        assert os.path.exists(os.path.join(dst, f))


def replay_scene():
    """The scene of tests/shim_replay.c in the test DSL (every number a short dyadic fraction)."""
    from scene_dsl import Camera, Cube, Light, Material, Mesh, MeshData, Node, Plane, Scene, Sphere, Triangle
    mat0 = Material(diffuse=(0.75, 0.25, 0.125), specular=(0.5, 0.5, 0.5), shininess=32.0)
    mat1 = Material(diffuse=(0.25, 0.5, 0.75), specular=(0.25, 0.25, 0.25), shininess=16.0)
    mat2 = Material(diffuse=(0.5, 0.75, 0.25), specular=(0.0, 0.0, 0.0), shininess=0.0)
    mat3 = Material(diffuse=(0.125, 0.125, 0.125), specular=(0.5, 0.5, 0.5), shininess=64.0, reflectivity=0.5)
    tent = MeshData(np.array([[-1.0, 0, -1], [1, 0, -1], [0, 2, 1], [0, 0, 3]]), np.array([[0, 1, 2], [1, 3, 2]], dtype=np.uint32), None, "tent")
    b = Node.group([Node.geo(Cube(), mat1).scaled(2.0), Node.geo(Mesh(tent), mat2).translated((0.0, 2.0, 0.0))]).scaled(0.5).translated((2.0, 0.0, -1.0))
    root = Node.group([
        Node.geo(Sphere(), mat0).scaled(2.0).translated((-2.5, 1.5, 0.0)),
        b,
        Node.group([b]).translated((-1.0, 3.0, -2.0)),
        Node.geo(Triangle((-4.0, -0.5, -3.0), (-2.0, -0.5, -3.0), (-3.0, 2.5, -3.0)), mat0),
        Node.geo(Plane(), mat3).scaled(16.0).translated((0.0, -1.0, 0.0)),
    ]).translated((0.0, -0.5, 0.0))
    scene = Scene(root=root, lights=[Light(position=(-4.0, 8.0, 6.0), color=(0.75, 0.75, 0.75)), Light(position=(6.0, 4.0, 8.0), color=(0.5, 0.25, 0.5))],
                  ambient=(0.25, 0.25, 0.25))
    cam = Camera(eye=(0.0, 1.0, 16.0), center=(0.0, 1.0, 0.0), fovy_degrees=32.0)
    return scene, cam


def test_replay_scene_flattens_as_the_c_file_assumes(oracle):
    """The orders tests/shim_replay.c hard-codes: breadth-first nodes A T P B0 B1 (C/B)0 (C/B)1."""
    scene, _ = replay_scene()
    flat = oracle.flatten(scene)
    assert list(flat["prim_type"]) == [0, 1, 4, 5, 2, 5, 2]


@pytest.mark.gpu
def test_shim_call_sequence_and_layout_render_the_oracle_image(oracle, tmp_path):
    exe, out = str(tmp_path / "shim_replay"), str(tmp_path / "out.rgb")
    lib_dir = os.path.join(ROOT, "portrayer_amd")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "shim_replay.c"), "-L", lib_dir, "-lportrayer_hip",
                    "-Wl,-rpath," + lib_dir, "-lm", "-o", exe], check=True)
    w, h, s = 200, 120, 3
    r = subprocess.run([exe, out, str(w), str(h), str(s)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.uint8).reshape(h, w, 3)
    scene, cam = replay_scene()
    ref = oracle.render(scene, cam, w, h, samples=s, seed=0, jitter=oracle.JITTER_RNG, mode=oracle.MODE_HIER)
    st = dict(zip(r.stdout.split()[0::2], map(int, r.stdout.split()[1::2])))
    assert st["primary"] == ref.stats["primary"] and st["shadow"] == ref.stats["shadow"] and st["reflect"] == ref.stats["reflect"] > 0
    assert np.array_equal(got, ref.rgb)
