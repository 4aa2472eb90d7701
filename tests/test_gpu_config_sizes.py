"""Parity at the sizes BASELINE.json's configs state (the smaller renders of test_gpu_render_parity.py compare
every pixel with the oracle; here the launch is the real one - every wavefront slot of the chip busy, large work
batches, partial last chunks - and the oracle can only afford a sample of pixels):

  C3 macho-cows 1280x720 SAMPLES=16, C4 entering-the-mirror-dimension 1920x1080 SAMPLES=64,
  C5 big-scene 3840x2160 SAMPLES=256 (its 8-way tile partition assembled like the 8-GPU run assembles it).

Checked: >= 64 oracle pixels per scene at the full sample count (half of them where the image changes fastest:
mesh silhouettes, mirror edges, shadow boundaries), ray accounting identities, determinism, halves == whole.
reference: render.rs:22-51 (pixel pipeline), :56-66 (slices), material.rs:149-179 (one shadow ray per hit and light)."""
import ctypes as C

import numpy as np
import pytest

from example_scenes import EXAMPLES
from scene_dsl import ASSETS, default_background

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from portrayer_amd import _hip
    return _hip


@pytest.fixture(scope="module")
def host():
    from portrayer_amd import host
    return host


def pick_pixels(rgb, n_edge=32, n_any=32, seed=11):
    """Pixels where the picture changes fastest (silhouettes, mirror edges, shadow boundaries) + random ones."""
    g = rgb.astype(np.int32)
    grad = np.zeros(g.shape[:2], dtype=np.int64)
    grad[:, 1:] += np.abs(g[:, 1:] - g[:, :-1]).sum(axis=2)
    grad[1:, :] += np.abs(g[1:, :] - g[:-1, :]).sum(axis=2)
    h, w = grad.shape
    order = np.argsort(grad.reshape(-1))[::-1]
    rng = np.random.default_rng(seed)
    edge = rng.choice(order[:max(n_edge * 50, n_edge)], size=n_edge, replace=False)  # spread over the 1600 strongest edges
    anyp = rng.choice(h * w, size=n_any, replace=False)
    return [(int(i % w), int(i // w)) for i in np.concatenate([edge, anyp])]


def check_against_oracle(oracle, ps, cam, rgb, w, h, samples, mode, pixels):
    bad = []
    for (x, y) in pixels:
        ref = oracle.render(ps, cam, w, h, samples=samples, seed=0, jitter=oracle.JITTER_RNG, mode=mode, rect=(x, y, x, y), threads=8)
        if not np.array_equal(ref.rgb[y, x], rgb[y, x]):
            bad.append(((x, y), tuple(ref.rgb[y, x]), tuple(rgb[y, x])))
    assert not bad, f"{len(bad)} of {len(pixels)} sampled pixels differ from the oracle: {bad[:5]}"


def render(host, H, r, sc, w, h, samples, rect=None, into=None, stats=False):
    rgb, _, st = r.render(sc.camera, w, h, default_background(w, h), samples=samples, seed=0, sample_mode=H.SAMPLE_RNG, rect=rect, into=into,
                          want_linear=False, stats=stats)
    return rgb, st


@pytest.mark.parametrize("name,w,h,samples,lights", [("macho-cows", 1280, 720, 16, 1), ("entering-the-mirror-dimension", 1920, 1080, 64, 3)])
@pytest.mark.parametrize("mode", ["flat", "hier", "kd"])
def test_config_size_render(oracle, host, H, name, w, h, samples, lights, mode):
    sc = host.Scene.example(name, assets=ASSETS)
    r = host.Renderer(sc, {"flat": H.TRAVERSE_FLAT, "hier": H.TRAVERSE_HIER, "kd": H.TRAVERSE_KD}[mode])
    rgb, st = render(host, H, r, sc, w, h, samples, stats=True)
    # ray accounting: one primary ray per pixel and sample; one shadow ray per shaded hit and light
    assert st["primary"] == w * h * samples
    assert st["shadow"] == lights * st["hits"]
    assert st["stack_overflow"] == 0
    # the instantiation bench.py times by default: the straight-line kernel - with the loop over the depth for the mirror scene (C4)
    assert not st["kernel_variant"] & H.KERNEL_INTERPRETER
    assert bool(st["kernel_variant"] & H.KERNEL_CHAIN) == (name == "entering-the-mirror-dimension")
    # determinism, and the reference's slice API (render.rs:56-66): two half renders into one image == the whole
    again, _ = render(host, H, r, sc, w, h, samples)
    assert np.array_equal(rgb, again)
    halves = np.zeros_like(rgb)
    render(host, H, r, sc, w, h, samples, rect=(0, 0, w - 1, h // 2 - 3), into=halves)
    render(host, H, r, sc, w, h, samples, rect=(0, h // 2 - 2, w - 1, h - 1), into=halves)
    assert np.array_equal(rgb, halves)
    r.close()
    cam = EXAMPLES[name]()[1]
    ps = oracle.pack_arrays(sc.export())
    if mode == "kd":  # the k-d semantics of scenes with plain Mesh instances: their own instantiation (9), one walk per wavefront
        assert st["kernel_mode"] == 9
    check_against_oracle(oracle, ps, cam, rgb, w, h, samples, {"flat": oracle.MODE_FLAT, "hier": oracle.MODE_HIER, "kd": oracle.MODE_KD}[mode], pick_pixels(rgb))


@pytest.mark.parametrize("mode,kernel_mode,waves", [("flat", 3, 6), ("hier", 6, 6), ("kd", 7, 5)])
def test_headline_frame_big_scene_1920x1080x64(oracle, host, H, mode, kernel_mode, waves):
    """The frame bench.py's headline (flat_scene), its default-semantics line (hierarchical) and its k-d line time: big-scene 1920x1080 SAMPLES=64, on the
    instantiation that is timed (asserted through pt_stats) - 64 oracle pixels at the full sample count, half of them on the strongest edges (VERDICT r03:
    the metric's own frame was only bracketed by smaller and larger ones; the k-d semantics were never checked at the size they are timed)."""
    w, h, samples = 1920, 1080, 64
    sc = host.Scene.example("big-scene", assets=ASSETS)
    r = host.Renderer(sc, {"flat": H.TRAVERSE_FLAT, "hier": H.TRAVERSE_HIER, "kd": H.TRAVERSE_KD}[mode])
    rgb, st = render(host, H, r, sc, w, h, samples)
    assert st["kernel_mode"] == kernel_mode and st["kernel_variant"] == waves, "not the instantiation bench.py times"
    counted, stc = render(host, H, r, sc, w, h, samples, stats=True)
    r.close()
    assert np.array_equal(counted, rgb)
    assert stc["primary"] == w * h * samples and stc["shadow"] == 3 * stc["hits"] and stc["stack_overflow"] == 0 and stc["kd_plane_miss"] == 0
    ps = oracle.pack_arrays(sc.export())
    check_against_oracle(oracle, ps, EXAMPLES["big-scene"]()[1], rgb, w, h, samples, {"flat": oracle.MODE_FLAT, "hier": oracle.MODE_HIER, "kd": oracle.MODE_KD}[mode],
                         pick_pixels(rgb))


def test_config5_big_scene_4k_256_samples_and_its_8_way_partition(oracle, host, H):
    """C5: big-scene 3840x2160 SAMPLES=256 (5.9e9 rays per frame). The single launch, the 8-way tile partition the 8-GPU run
    uses (each rank's compact buffer rendered here one after the other, concatenated as the gather delivers them,
    pt_untile_device) and 64 oracle pixels at the full 256 samples."""
    w, h, samples, ranks = 3840, 2160, 256, 8
    sc = host.Scene.example("big-scene", assets=ASSETS)
    r = host.Renderer(sc, H.TRAVERSE_FLAT)
    rgb, st = render(host, H, r, sc, w, h, samples, stats=True)
    assert st["primary"] == w * h * samples and st["shadow"] == 3 * st["hits"] and st["reflect"] == 0 and st["stack_overflow"] == 0
    again, _ = render(host, H, r, sc, w, h, samples)
    assert np.array_equal(rgb, again)
    lib, ctx = H.lib(), r.context
    cam = host.camera(sc.camera, w, h)
    bg = default_background(w, h)

    def check(rc, what):
        assert rc == 0, f"{what}: {rc} {lib.pt_last_error(ctx).decode()}"

    d_bg, d_gath, d_full = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.pt_device_alloc(ctx, bg.nbytes, C.byref(d_bg)), "alloc")
    check(lib.pt_copy_to_device(ctx, d_bg, bg.ctypes.data_as(C.c_void_p), bg.nbytes), "copy")
    p0 = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), samples, 0, H.SAMPLE_RNG, 1, 0, ranks, 0)
    per = int(lib.pt_compact_bytes(C.byref(p0)))
    check(lib.pt_device_alloc(ctx, per * ranks, C.byref(d_gath)), "alloc")
    check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_full)), "alloc")
    rays = 0
    for k in range(ranks):
        p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), samples, 0, H.SAMPLE_RNG, 1, k, ranks, 1)
        s1 = H.PtStats()
        check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(p), 1, C.c_void_p(d_gath.value + k * per), None), "pt_render_device")
        check(lib.pt_render_finish(ctx, C.byref(s1)), "pt_render_finish")
        rays += s1.primary + s1.shadow
    check(lib.pt_untile_device(ctx, C.byref(p0), d_gath, d_full, None), "pt_untile_device")
    check(lib.pt_synchronize(ctx), "sync")
    img = np.zeros((h, w, 3), dtype=np.uint8)
    check(lib.pt_copy_from_device(ctx, img.ctypes.data_as(C.c_void_p), d_full, img.nbytes), "copy back")
    for d in (d_bg, d_gath, d_full):
        lib.pt_device_free(ctx, d)
    assert np.array_equal(img, rgb), "the 8-way partition does not assemble to the single-launch image"
    assert rays == st["primary"] + st["shadow"]
    r.close()
    ps = oracle.pack_arrays(sc.export())
    check_against_oracle(oracle, ps, EXAMPLES["big-scene"]()[1], rgb, w, h, samples, oracle.MODE_FLAT, pick_pixels(rgb))
