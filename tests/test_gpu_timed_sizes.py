"""Parity where bench.py times: the secondary workloads of the JSON line at THEIR size and on THEIR kernel instantiation.

  big-soup (1,253,664 baked triangles, tree built on the device) and big-mesh (216 instances of cow.obj) at 1920x1080 SAMPLES=64:
      the straight-line kernel at 4 waves per SIMD, a wavefront = one pixel's 64 samples (pt_render_simple.h, pt_trace_packet_mesh);
  transmission-refraction ("aquarium") at 1920x1080 SAMPLES=16: the interpreter kernel with a parked frame in LDS, textured.

The smaller renders of test_gpu_render_parity.py compare every pixel; at these sizes the oracle can afford a sample: 32 pixels per
scene at the full sample count, half of them where the image changes fastest. pt_stats.kernel_mode / kernel_variant (ABI 6) prove
that the launch checked here is the instantiation the bench runs by default.
reference: mesh.rs:146-167 (box, then every triangle), material.rs:216-303 (recursion), render.rs:22-51 (pixel pipeline)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from scene_dsl import ASSETS, default_background
from test_gpu_config_sizes import pick_pixels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from portrayer_amd import _hip
    return _hip


@pytest.fixture(scope="module")
def host():
    from portrayer_amd import host
    return host


def check_pixels_in_parallel(oracle, ps, cam, rgb, w, h, samples, mode, pixels, workers=32):
    """One oracle call per pixel (a pixel of the 1.25 M-triangle soup costs the CPU several seconds: the reference scans every
    triangle of a mesh, mesh.rs:157-166), the calls side by side on the host's cores (ctypes releases the GIL)."""
    def one(xy):
        x, y = xy
        ref = oracle.render(ps, cam, w, h, samples=samples, seed=0, jitter=oracle.JITTER_RNG, mode=mode, rect=(x, y, x, y), threads=1)
        return (xy, tuple(int(v) for v in ref.rgb[y, x]), tuple(int(v) for v in rgb[y, x]))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(one, pixels))
    bad = [r for r in res if r[1] != r[2]]
    assert not bad, f"{len(bad)} of {len(pixels)} sampled pixels differ from the oracle: {bad[:5]}"


@pytest.mark.parametrize("name,mode_id", [("synthetic:big-soup", 1), ("synthetic:big-mesh", 1)])
def test_synthetic_million_triangle_scenes_at_the_timed_size(oracle, host, H, name, mode_id):
    w, h, samples = 1920, 1080, 64
    sc = host.Scene.example(name, n=6, assets=ASSETS)
    r = host.Renderer(sc, H.TRAVERSE_FLAT)
    bg = default_background(w, h)
    counted, _, st = r.render(sc.camera, w, h, bg, samples=samples, seed=0, sample_mode=H.SAMPLE_RNG, want_linear=False, stats=True)
    rgb, _, timed = r.render(sc.camera, w, h, bg, samples=samples, seed=0, sample_mode=H.SAMPLE_RNG, want_linear=False)
    r.close()
    assert np.array_equal(counted, rgb), "the counting and the timed instantiation must render the same image"
    assert timed["kernel_mode"] == mode_id and timed["kernel_variant"] == 4, "the default for these scenes in flat_scene: straight-line kernel, 4 waves per SIMD (round 5, c47; the hierarchical semantics take 5), untextured"
    assert st["kernel_variant"] == (4 | H.KERNEL_COUNTING)
    assert st["primary"] == w * h * samples and st["shadow"] == 3 * st["hits"] and st["reflect"] == 0 and st["stack_overflow"] == 0
    ps = oracle.pack_arrays(sc.export())
    check_pixels_in_parallel(oracle, ps, sc.camera, rgb, w, h, samples, oracle.MODE_FLAT, pick_pixels(rgb, 16, 16))


def test_transmission_refraction_at_the_timed_size(oracle, host, H):
    w, h, samples = 1920, 1080, 16
    sc = host.Scene.example("transmission-refraction", assets=ASSETS)
    r = host.Renderer(sc, H.TRAVERSE_FLAT)
    bg = default_background(w, h)
    rgb, _, st = r.render(sc.camera, w, h, bg, samples=samples, seed=0, sample_mode=H.SAMPLE_RNG, want_linear=False, stats=True)
    again, _, timed = r.render(sc.camera, w, h, bg, samples=samples, seed=0, sample_mode=H.SAMPLE_RNG, want_linear=False)
    r.close()
    assert np.array_equal(rgb, again)
    assert timed["kernel_variant"] == (3 | H.KERNEL_INTERPRETER | H.KERNEL_PARK | H.KERNEL_TEXTURED), "reflective, textured: the interpreter with a parked frame in LDS (fork / join is opt-in)"
    assert st["primary"] == w * h * samples and st["reflect"] > 0 and st["refract"] > 0 and st["stack_overflow"] == 0
    ps = oracle.pack_arrays(sc.export())
    check_pixels_in_parallel(oracle, ps, sc.camera, rgb, w, h, samples, oracle.MODE_FLAT, pick_pixels(rgb, 16, 16))
