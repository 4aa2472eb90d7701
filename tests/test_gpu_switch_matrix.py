"""The kernel instantiations only an environment switch reaches, against the oracle - inside the GPU suite (round 4 ran this matrix by hand,
profiles/r04/calls/c55.sh, and it was the matrix that found the one wrong instantiation, pt_render_kernel<2, true, *, 0> under PORTRAYER_PARK=0).

Every switch of pt_render_common (csrc/pt_api.hip) x four scenes that between them reach every kernel family - dielectric recursion (interpreter, parked
frame / fork), opaque mirrors (chain kernel), a mesh-free scene (straight-line kernels at 3 - 6 waves per SIMD), meshes (walks inside instances; the host's
and the device's tree builder) - x the crate's three traversal semantics, counting AND plain instantiation each: u8 pixels, f64 means and ray counts."""
import numpy as np
import pytest

import host_glue
from scene_dsl import ASSETS, default_background
from test_gpu_render_parity import random_scene
from ulp import assert_ulp

pytestmark = pytest.mark.gpu

SWITCHES = ["PORTRAYER_WAVES=3", "PORTRAYER_WAVES=4", "PORTRAYER_WAVES=5", "PORTRAYER_KD_WAVES=3", "PORTRAYER_KD_WAVES=4", "PORTRAYER_CHAIN_WAVES=3",
            "PORTRAYER_CHAIN=0", "PORTRAYER_MESH_OCT=0", "PORTRAYER_FORK=1", "PORTRAYER_PARK=0", "PORTRAYER_PARK=0 PORTRAYER_FORK=1",
            "PORTRAYER_FINE_QUEUES=0", "PORTRAYER_LANE_CHUNKS=1", "PORTRAYER_BUILD=host", "PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16", "PORTRAYER_KD_CULL=0"]

_ORACLE = {}  # (scene, mode) -> oracle render: the same for every switch


def _scenes(host):
    """name -> (host scene, camera (10 doubles), oracle scene or None (= from the host scene's export), w, h, samples, kd_depth)"""
    from example_scenes import TEXTURED_EXAMPLES
    out = {}
    for name, (seed, dielectric) in {"glass": (2, True), "mirrors": (101, False)}.items():
        scene, cam = random_scene(seed, dielectric=dielectric)
        out[name] = (host_glue.host_scene(scene), host_glue.cam10(cam), (scene, cam), 80, 56, 2, 6)
    for name, (w, h, s) in {"big-scene": (64, 36, 8), "macho-cows": (64, 36, 2)}.items():
        sc = host.Scene.example(name, assets=ASSETS)
        out[name] = (sc, sc.camera, None, w, h, s, 10)
    scene, cam = TEXTURED_EXAMPLES["transmission-refraction"]()[:2]  # textured KDMesh fish in glass and water: the textured interpreter instantiations
    out["aquarium"] = (host_glue.host_scene(scene), host_glue.cam10(cam), (scene, cam), 64, 36, 8, 5)
    return out


@pytest.fixture(scope="module")
def scenes():
    from portrayer_amd import host
    return _scenes(host)


@pytest.mark.parametrize("switch", SWITCHES)
def test_every_switch_renders_what_the_oracle_renders(oracle, scenes, monkeypatch, switch):
    from example_scenes import EXAMPLES
    from portrayer_amd import _hip as H
    from portrayer_amd import host
    for k in ("PORTRAYER_WAVES", "PORTRAYER_KD_WAVES", "PORTRAYER_CHAIN_WAVES", "PORTRAYER_CHAIN", "PORTRAYER_MESH_OCT", "PORTRAYER_FORK", "PORTRAYER_PARK",
              "PORTRAYER_FINE_QUEUES", "PORTRAYER_LANE_CHUNKS", "PORTRAYER_BUILD", "PORTRAYER_BUILD_MIN", "PORTRAYER_KD_CULL"):
        monkeypatch.delenv(k, raising=False)
    for kv in switch.split():
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    seen = set()
    for name, (hs, cam10, dsl, w, h, samples, kd_depth) in scenes.items():
        bg = default_background(w, h)
        for mode, tr, om in (("flat", H.TRAVERSE_FLAT, oracle.MODE_FLAT), ("kd", H.TRAVERSE_KD, oracle.MODE_KD), ("hier", H.TRAVERSE_HIER, oracle.MODE_HIER)):
            key = (name, mode)
            if key not in _ORACLE:
                if dsl is None:
                    _ORACLE[key] = oracle.render(oracle.pack_arrays(hs.export()), EXAMPLES[name]()[1], w, h, samples=samples, seed=9, jitter=oracle.JITTER_RNG, mode=om, kd_depth=kd_depth)
                else:
                    packed = oracle.pack(dsl[0]) if name != "aquarium" else dsl[0]
                    _ORACLE[key] = oracle.render(packed, dsl[1], w, h, samples=samples, seed=9, jitter=oracle.JITTER_RNG, mode=om, kd_depth=kd_depth)
            ref = _ORACLE[key]
            r = host.Renderer(hs, tr, kd_depth=kd_depth)  # (a new renderer: PORTRAYER_BUILD / MESH_OCT / KD_CULL act at upload)
            kw = dict(samples=samples, seed=9, sample_mode=H.SAMPLE_RNG)
            rgb, linear, st = r.render(cam10, w, h, bg, stats=True, **kw)
            plain, plain_linear, st0 = r.render(cam10, w, h, bg, **kw)
            r.close()
            where = f"{switch}: {name} / {mode} (kernel mode {st['kernel_mode']}, variant {st['kernel_variant']})"
            assert st["kernel_variant"] & H.KERNEL_COUNTING and not st0["kernel_variant"] & H.KERNEL_COUNTING, where
            for k in ("primary", "shadow", "reflect", "refract", "hits"):
                assert st[k] == ref.stats[k], (where, k)
            assert ref.stats["tex_sphere_near_edge"] == 0  # (no sphere texture coordinate near a texel edge: tests/test_gpu_textures.py::texel_edge_proof - so exact)
            assert np.array_equal(rgb, ref.rgb), where
            assert np.array_equal(plain, rgb), where
            assert_ulp(linear, ref.linear, 0)
            assert_ulp(plain_linear, ref.linear, 0)
            seen.add((st["kernel_mode"], st["kernel_variant"]))
            seen.add((st0["kernel_mode"], st0["kernel_variant"]))
    assert len(seen) >= 12  # (counting + plain) of several families: the switch did not collapse everything onto one kernel
    if switch == "PORTRAYER_PARK=0":  # the instantiation round 4 found wrong and round 5 root-caused: it did run
        assert any(m == 2 and (v & H.KERNEL_INTERPRETER) and not (v & H.KERNEL_PARK) and (v & H.KERNEL_COUNTING) for (m, v) in seen)
