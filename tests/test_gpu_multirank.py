"""The multi-rank PRODUCT path on one GPU: pt_render_device(compact=1) for every rank of a tile partition,
the rank-major concatenation a gather produces, pt_untile_device -> must equal the single-launch image
(the compact indexing of pt_finish_kernel, pt_render_finish and pt_untile_kernel are otherwise reached
only through bench.py). Also: the launch-size guard and the unconditional stack-overflow report."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from scene_dsl import ASSETS, default_background

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def H():
    from portrayer_amd import _hip
    return _hip


@pytest.fixture(scope="module")
def host():
    from portrayer_amd import host
    return host


def render_partition(H, host, renderer, scene, w, h, rect, ranks, samples, bg):
    """Every rank's compact buffer rendered on this GPU, concatenated rank-major, untiled on the device."""
    lib, ctx = H.lib(), renderer.context
    cam = host.camera(scene.camera, w, h)

    def check(rc, what):
        assert rc == 0, f"{what}: {rc} {lib.pt_last_error(ctx).decode()}"

    d_bg = C.c_void_p()
    check(lib.pt_device_alloc(ctx, bg.nbytes, C.byref(d_bg)), "alloc")
    check(lib.pt_copy_to_device(ctx, d_bg, bg.ctypes.data_as(C.c_void_p), bg.nbytes), "copy")
    p0 = H.PtRenderParams(w, h, H.PtRect(*rect), samples, 3, H.SAMPLE_RNG, 1, 0, ranks, 0)
    per = int(lib.pt_compact_bytes(C.byref(p0)))
    d_gath, d_full = C.c_void_p(), C.c_void_p()
    check(lib.pt_device_alloc(ctx, per * ranks, C.byref(d_gath)), "alloc")
    check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_full)), "alloc")
    start = np.full((h, w, 3), 7, dtype=np.uint8)  # pixels outside the slice must keep these bytes
    check(lib.pt_copy_to_device(ctx, d_full, start.ctypes.data_as(C.c_void_p), start.nbytes), "copy")
    rays = 0
    for r in range(ranks):
        p = H.PtRenderParams(w, h, H.PtRect(*rect), samples, 3, H.SAMPLE_RNG, 1, r, ranks, 1)
        st = H.PtStats()
        check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(p), 1, C.c_void_p(d_gath.value + r * per), None), "pt_render_device")
        check(lib.pt_render_finish(ctx, C.byref(st)), "pt_render_finish")
        rays += st.primary + st.shadow + st.reflect + st.refract
    check(lib.pt_untile_device(ctx, C.byref(p0), d_gath, d_full, None), "pt_untile_device")
    check(lib.pt_synchronize(ctx), "sync")
    img = np.zeros((h, w, 3), dtype=np.uint8)
    check(lib.pt_copy_from_device(ctx, img.ctypes.data_as(C.c_void_p), d_full, img.nbytes), "copy back")
    for d in (d_bg, d_gath, d_full):
        lib.pt_device_free(ctx, d)
    return img, rays


@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("w,h,rect", [(200, 120, (0, 0, 199, 119)), (157, 93, (11, 5, 149, 90))])
def test_partition_assembles_to_the_single_launch_image(H, host, ranks, w, h, rect):
    sc = host.Scene.example("entering-the-mirror-dimension", assets=ASSETS)
    r = host.Renderer(sc, H.TRAVERSE_FLAT)
    bg = default_background(w, h)
    img, rays = render_partition(H, host, r, sc, w, h, rect, ranks, 9, bg)  # 9 samples: two chunks, the second partial
    one = np.full((h, w, 3), 7, dtype=np.uint8)
    _, _, st = r.render(sc.camera, w, h, bg, samples=9, seed=3, sample_mode=H.SAMPLE_RNG, rect=rect, into=one, want_linear=False, stats=True)
    assert np.array_equal(img, one)
    assert rays == st["primary"] + st["shadow"] + st["reflect"] + st["refract"]
    x0, y0, x1, y1 = rect
    outside = np.ones((h, w), dtype=bool); outside[y0:y1 + 1, x0:x1 + 1] = False
    assert (img[outside] == 7).all()
    r.close()


def test_launch_size_guard(H, host):
    """pixel slots x ceil(samples / 8) is a 32-bit index: what would wrap is refused, not rendered wrongly."""
    sc = host.Scene.example("single-triangle", assets=ASSETS)
    r = host.Renderer(sc, H.TRAVERSE_FLAT)
    lib, ctx = H.lib(), r.context
    w, h = 7680, 4320
    cam = host.camera(sc.camera, w, h)
    p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), 1100, 0, H.SAMPLE_CENTRE, 1, 0, 1, 0)  # 33.2 M slots x 138 chunks > 2^32
    rc = lib.pt_render_device(ctx, C.byref(cam), C.c_void_p(16), C.byref(p), 0, C.c_void_p(16), None)
    assert rc == -1 and b"too large" in lib.pt_last_error(ctx)
    r.close()


_OVERFLOW = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from portrayer_amd import _hip as H, host
from scene_dsl import ASSETS, default_background
sc = host.Scene.example("big-scene", assets=ASSETS)
r = host.Renderer(sc, H.TRAVERSE_FLAT)
try:
    r.render(sc.camera, 64, 48, default_background(64, 48), stats=False, want_linear=False)
    print("NO ERROR")
except Exception as e:
    print("ERR", e)
"""


def test_stack_overflow_is_reported_without_the_counting_build(H):
    """PORTRAYER_STACK_CAP=2 makes the walk of a 1000-node scene run out of stack: the plain (non-STATS)
    kernel must fail the render with PT_ERR_TRAVERSAL (-6)."""
    env = dict(os.environ, PORTRAYER_STACK_CAP="2")
    out = subprocess.run([sys.executable, "-c", _OVERFLOW % (ROOT, HERE)], env=env, capture_output=True, text=True, timeout=300)
    assert "ERR" in out.stdout and "-6" in out.stdout and "overflow" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------------------------------------------
# pt_node_*: the product's own multi-GPU path (one context per rank, gather, untile) - here with every
# rank on GPU 0 (RCCL needs distinct GPUs; with shared devices the gather is device-to-device copies)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ranks", [2, 3])
def test_node_render_equals_single_context(oracle, H, ranks):
    import device_glue
    from example_scenes import EXAMPLES
    scene, cam, _ = EXAMPLES["entering-the-mirror-dimension"]()
    ds = device_glue.DeviceScene(scene, H.TRAVERSE_FLAT)
    lib = H.lib()
    w, h, rect = 173, 101, (3, 2, 170, 99)
    bg = default_background(w, h)
    ctx = H.Context()
    ds.upload(ctx)
    one = np.full((h, w, 3), 9, dtype=np.uint8)
    device_glue.render(ctx, cam, w, h, samples=5, seed=2, sample_mode=H.SAMPLE_RNG, rect=rect, into=one)
    ctx.close()
    node = C.c_void_p()
    devs = (C.c_int32 * ranks)(*([0] * ranks))
    assert lib.pt_node_create(ranks, devs, C.byref(node)) == 0
    assert lib.pt_node_ranks(node) == ranks and lib.pt_node_uses_rccl(node) == 0
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0, lib.pt_node_last_error(node)
    img = np.full((h, w, 3), 9, dtype=np.uint8)
    p = H.PtRenderParams(w, h, H.PtRect(*rect), 5, 2, H.SAMPLE_RNG, 1, 0, 1, 1)
    st = H.PtStats()
    camera = device_glue.camera_struct(cam, w, h)
    rc = lib.pt_node_render(node, C.byref(camera), bg.ctypes.data_as(H._dp), C.byref(p), img.ctypes.data_as(H._u8p), C.byref(st))
    assert rc == 0, lib.pt_node_last_error(node)
    assert np.array_equal(img, one)
    ref = oracle.render(scene, cam, w, h, samples=5, seed=2, jitter=oracle.JITTER_RNG, mode=oracle.MODE_FLAT, rect=rect)
    assert st.primary == ref.stats["primary"] and st.shadow == ref.stats["shadow"] and st.reflect == ref.stats["reflect"]
    lib.pt_node_destroy(node)


def test_node_gather_through_rccl_with_one_rank(H, monkeypatch):
    """The RCCL leg of pt_node_render (dlopen of librccl, ncclCommInitAll, ncclGather inside a group, untile of what the
    gather delivered) as far as a 1-GPU box can run it: a communicator of ONE rank (PORTRAYER_NODE_RCCL=1 - without it a
    single-GPU node skips RCCL). The gather between separate GPUs stays unexecuted here."""
    import device_glue
    from example_scenes import EXAMPLES
    scene, cam, _ = EXAMPLES["macho-cows"]()
    ds = device_glue.DeviceScene(scene, H.TRAVERSE_FLAT)
    lib = H.lib()
    w, h = 131, 77
    bg = default_background(w, h)
    ctx = H.Context()
    ds.upload(ctx)
    one, _, _ = device_glue.render(ctx, cam, w, h, samples=3, seed=5, sample_mode=H.SAMPLE_RNG)
    ctx.close()
    monkeypatch.setenv("PORTRAYER_NODE_RCCL", "1")
    node = C.c_void_p()
    devs = (C.c_int32 * 1)(0)
    assert lib.pt_node_create(1, devs, C.byref(node)) == 0
    assert lib.pt_node_ranks(node) == 1 and lib.pt_node_uses_rccl(node) == 1
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0, lib.pt_node_last_error(node)
    camera = device_glue.camera_struct(cam, w, h)
    for _ in range(2):  # twice: buffers and communicator are reused
        img = np.zeros((h, w, 3), dtype=np.uint8)
        p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), 3, 5, H.SAMPLE_RNG, 1, 0, 1, 1)
        rc = lib.pt_node_render(node, C.byref(camera), bg.ctypes.data_as(H._dp), C.byref(p), img.ctypes.data_as(H._u8p), None)
        assert rc == 0, lib.pt_node_last_error(node)
        assert np.array_equal(img, one)
    lib.pt_node_destroy(node)


@pytest.mark.parametrize("ranks,threads", [(3, "1"), (8, "1"), (4, "0")])
def test_node_frames_in_a_pipeline(H, monkeypatch, ranks, threads):
    """pt_node_frame_begin / pt_node_frame_end (ABI 7): two frames open at a time - frame k + 1 is rendered into the other set of tile
    buffers while frame k is gathered and untiled on the gather streams, every rank launched by its own host thread (PORTRAYER_NODE_THREADS=0:
    by the caller). Frames with DIFFERENT cameras, so that a frame assembled from the wrong buffer set cannot pass: every image == the
    single-context render of its camera, a third begin without an end is refused, the stats are each frame's own."""
    import device_glue
    from example_scenes import EXAMPLES
    from scene_dsl import Camera
    monkeypatch.setenv("PORTRAYER_NODE_THREADS", threads)
    scene, cam0, _ = EXAMPLES["macho-cows"]()
    cams = [cam0] + [Camera(eye=(cam0.eye[0] + 0.7 * k, cam0.eye[1] + 0.3 * k, cam0.eye[2] - 0.5 * k), center=cam0.center, up=cam0.up, fovy_degrees=cam0.fovy_degrees) for k in (1, 2, 3, 4)]
    ds = device_glue.DeviceScene(scene, H.TRAVERSE_FLAT)
    lib = H.lib()
    w, h = 200, 120
    bg = default_background(w, h)
    ctx = H.Context()
    ds.upload(ctx)
    singles = []
    for cam in cams:
        img, _, st1 = device_glue.render(ctx, cam, w, h, samples=2, seed=4, sample_mode=H.SAMPLE_RNG, stats=True)
        singles.append((img, st1))
    ctx.close()
    node = C.c_void_p()
    devs = (C.c_int32 * ranks)(*([0] * ranks))
    assert lib.pt_node_create(ranks, devs, C.byref(node)) == 0
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0, lib.pt_node_last_error(node)
    p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), 2, 4, H.SAMPLE_RNG, 1, 0, 1, 1)
    assert lib.pt_node_upload_background(node, bg.ctypes.data_as(H._dp), C.byref(p), None) == 0, lib.pt_node_last_error(node)
    cs = [device_glue.camera_struct(cam, w, h) for cam in cams]
    st = H.PtStats()
    hm = (C.c_double * 5)()

    def check(k):  # close the oldest frame: it must be frame k
        assert lib.pt_node_frame_end(node, C.byref(st)) == 0, lib.pt_node_last_error(node)
        assert (st.primary, st.shadow, st.hits) == (singles[k][1]["primary"], singles[k][1]["shadow"], singles[k][1]["hits"]), k
        assert lib.pt_node_last_frame_host_ms(node, C.byref(hm)) == 0 and hm[0] > 0.0 and hm[4] > 0.0

    assert lib.pt_node_frame_end(node, None) != 0  # nothing open
    assert lib.pt_node_frame_begin(node, C.byref(cs[0]), C.byref(p)) == 0, lib.pt_node_last_error(node)
    assert lib.pt_node_frame_begin(node, C.byref(cs[1]), C.byref(p)) == 0, lib.pt_node_last_error(node)
    assert lib.pt_node_frames_in_flight(node) == 2
    assert lib.pt_node_frame_begin(node, C.byref(cs[2]), C.byref(p)) != 0  # a third open frame is refused ...
    img = np.zeros((h, w, 3), dtype=np.uint8)
    assert lib.pt_node_download_image(node, C.byref(p), img.ctypes.data_as(H._u8p)) != 0  # ... and so is reading the image under open frames
    check(0)
    for k in (2, 3, 4):  # begin k, close k - 1: always two open
        assert lib.pt_node_frame_begin(node, C.byref(cs[k]), C.byref(p)) == 0, lib.pt_node_last_error(node)
        check(k - 1)
    check(4)
    assert lib.pt_node_frames_in_flight(node) == 0
    assert lib.pt_node_download_image(node, C.byref(p), img.ctypes.data_as(H._u8p)) == 0, lib.pt_node_last_error(node)
    assert np.array_equal(img, singles[4][0]), "the last frame's image is not its camera's"
    # and frame by frame: the image of every camera
    for k in (1, 3, 0):
        assert lib.pt_node_frame_begin(node, C.byref(cs[k]), C.byref(p)) == 0
        check(k)
        assert lib.pt_node_download_image(node, C.byref(p), img.ctypes.data_as(H._u8p)) == 0
        assert np.array_equal(img, singles[k][0]), k
    lib.pt_node_destroy(node)


@pytest.mark.parametrize("scene_name,traverse", [("macho-cows", "flat"), ("entering-the-mirror-dimension", "kd"), ("transmission-refraction", "flat")])
@pytest.mark.parametrize("two", ["0", "1"])
def test_two_frames_in_flight_on_two_streams_of_one_context(H, monkeypatch, scene_name, traverse, two):
    """ABI 8: the two renders a context may have in flight own their work buffers (chunk sums, recursion frames, stack columns, counters, work queues) and
    may run on the context's two streams at once - the next frame's persistent wavefronts start where this frame's tail frees places. Frames with DIFFERENT
    cameras and sample seeds, alternating slots, the older frame closed after the newer one is queued: every image and every frame's own ray counts == the
    same render done alone (straight-line, chain and interpreter kernels: the last two park recursion frames in the slot's HBM lines)."""
    import device_glue
    from example_scenes import EXAMPLES, TEXTURED_EXAMPLES
    from scene_dsl import Camera
    scene, cam0 = (TEXTURED_EXAMPLES if scene_name in TEXTURED_EXAMPLES else EXAMPLES)[scene_name]()[:2]
    tr = H.TRAVERSE_KD if traverse == "kd" else H.TRAVERSE_FLAT
    if two == "1":  # two streams forced for every scene: the recursion frames of two launches must not meet either
        monkeypatch.setenv("PORTRAYER_TWO_STREAMS", "1")
    cams = [cam0] + [Camera(eye=(cam0.eye[0] + 0.4 * k, cam0.eye[1] + 0.2 * k, cam0.eye[2] - 0.3 * k), center=cam0.center, up=cam0.up, fovy_degrees=cam0.fovy_degrees) for k in (1, 2, 3, 4, 5)]
    ds = device_glue.DeviceScene(scene, tr)
    lib = H.lib()
    w, h, samples = 160, 96, 4
    bg = default_background(w, h)
    ctx = H.Context()
    ds.upload(ctx)
    alone = [device_glue.render(ctx, cam, w, h, samples=samples, seed=10 + k, sample_mode=H.SAMPLE_RNG, stats=True) for k, cam in enumerate(cams)]
    c = ctx._h
    d_bg = C.c_void_p()
    assert lib.pt_device_alloc(c, bg.nbytes, C.byref(d_bg)) == 0
    assert lib.pt_copy_to_device(c, d_bg, bg.ctypes.data_as(C.c_void_p), bg.nbytes) == 0
    d_img = [C.c_void_p(), C.c_void_p()]
    for d in d_img:
        assert lib.pt_device_alloc(c, w * h * 3, C.byref(d)) == 0
    # two streams where the scene's kernels keep no recursion frames in HBM, one for both slots where they do (csrc/pt_api.hip: pt_context_stream)
    assert lib.pt_context_stream(c, 0) and lib.pt_context_stream(c, 1)
    assert (lib.pt_context_stream(c, 0) != lib.pt_context_stream(c, 1)) == (scene_name == "macho-cows" or two == "1")
    got, slots = [], []
    st = H.PtStats()

    def close(k):  # the OLDEST open frame is frame k: its image is in the buffer of the slot it took
        assert lib.pt_render_finish(c, C.byref(st)) == 0, lib.pt_last_error(c)
        img = np.zeros((h, w, 3), dtype=np.uint8)
        assert lib.pt_copy_from_device(c, img.ctypes.data_as(C.c_void_p), d_img[slots[k]], img.nbytes) == 0
        got.append((img, st.as_dict()))

    for k, cam in enumerate(cams):
        slot = int(lib.pt_context_next_slot(c))
        slots.append(slot)
        p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), samples, 10 + k, H.SAMPLE_RNG, 1, 0, 1, 1)
        camera = device_glue.camera_struct(cam, w, h)
        assert lib.pt_render_device(c, C.byref(camera), d_bg, C.byref(p), 0, d_img[slot], C.c_void_p(lib.pt_context_stream(c, slot))) == 0, lib.pt_last_error(c)
        if k > 0:
            close(k - 1)
    close(len(cams) - 1)
    assert slots == [0, 1, 0, 1, 0, 1]
    p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), samples, 10, H.SAMPLE_RNG, 1, 0, 1, 1)
    for k, ((img, stk), (ref_img, _, ref_st)) in enumerate(zip(got, alone)):
        assert np.array_equal(img, ref_img), f"frame {k} differs from the same render done alone"
        for key in ("primary", "shadow", "reflect", "refract", "hits"):
            assert stk[key] == ref_st[key], (k, key)
    for d in d_img + [d_bg]:
        lib.pt_device_free(c, d)
    ctx.close()


def test_node_failure_behind_a_launch_and_uploads_under_open_frames(H, monkeypatch):
    """(1) a HIP call that fails BEHIND a rank's launch must come back as one of the header's negative codes with its own
    text (the old sentinel PT_ERR_DEVICE + 1000 = 998 was positive and never recognised), the launches that did go out closed, and the
    node usable afterwards; (2) pt_node_upload_background / pt_node_scene_upload are refused while frames are open BEFORE they touch a
    buffer - the open frame still delivers its image."""
    import device_glue
    from example_scenes import EXAMPLES
    scene, cam, _ = EXAMPLES["macho-cows"]()
    ds = device_glue.DeviceScene(scene, H.TRAVERSE_FLAT)
    lib = H.lib()
    w, h = 96, 64
    bg = default_background(w, h)
    ctx = H.Context()
    ds.upload(ctx)
    one, _, _ = device_glue.render(ctx, cam, w, h, samples=2, seed=4, sample_mode=H.SAMPLE_RNG)
    ctx.close()
    p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), 2, 4, H.SAMPLE_RNG, 1, 0, 1, 1)
    big = H.PtRenderParams(2 * w, 2 * h, H.PtRect(0, 0, 2 * w - 1, 2 * h - 1), 2, 4, H.SAMPLE_RNG, 1, 0, 1, 1)
    bg_big = default_background(2 * w, 2 * h)
    camera = device_glue.camera_struct(cam, w, h)
    devs = (C.c_int32 * 3)(0, 0, 0)
    img = np.zeros((h, w, 3), dtype=np.uint8)
    # (1) the injected failure
    monkeypatch.setenv("PORTRAYER_NODE_FAIL_AFTER_LAUNCH", "1")
    node = C.c_void_p()
    assert lib.pt_node_create(3, devs, C.byref(node)) == 0
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0
    assert lib.pt_node_upload_background(node, bg.ctypes.data_as(H._dp), C.byref(p), None) == 0
    rc = lib.pt_node_frame_begin(node, C.byref(camera), C.byref(p))
    assert rc == H.ERR_DEVICE and rc < 0, rc
    msg = lib.pt_node_last_error(node).decode()
    assert "rank 1" in msg and "behind its launch" in msg, msg
    assert lib.pt_node_frames_in_flight(node) == 0  # drained and closed
    lib.pt_node_destroy(node)
    monkeypatch.delenv("PORTRAYER_NODE_FAIL_AFTER_LAUNCH")
    # (2) uploads under an open frame
    node = C.c_void_p()
    assert lib.pt_node_create(3, devs, C.byref(node)) == 0
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0
    assert lib.pt_node_upload_background(node, bg.ctypes.data_as(H._dp), C.byref(p), None) == 0
    assert lib.pt_node_frame_begin(node, C.byref(camera), C.byref(p)) == 0, lib.pt_node_last_error(node)
    assert lib.pt_node_upload_background(node, bg_big.ctypes.data_as(H._dp), C.byref(big), None) == H.ERR_ARGUMENT  # a LARGER image: would have reallocated the open frame's target
    assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == H.ERR_ARGUMENT
    assert b"in flight" in lib.pt_node_last_error(node)
    assert lib.pt_node_frame_end(node, None) == 0, lib.pt_node_last_error(node)
    assert lib.pt_node_download_image(node, C.byref(p), img.ctypes.data_as(H._u8p)) == 0
    assert np.array_equal(img, one)
    lib.pt_node_destroy(node)


def test_node_rccl_leg_inside_a_process_that_carries_torch(H):
    """bench.py drives pt_node from a process that has imported torch - which brings its own bundled RCCL - while pt_node dlopens the
    system's librccl: the combination the driver's multi-GPU run uses and no test had executed. A subprocess (so that this
    test process's own state does not matter) imports torch, initialises its CUDA side, then runs the one-rank RCCL leg of the node twice,
    pipelined, and compares with the single-context image."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
assert torch.cuda.is_available()
x = torch.ones(1024, device="cuda:0"); torch.cuda.synchronize()
import torch.distributed  # (the process group machinery that loads torch's RCCL)
import numpy as np
os.environ["PORTRAYER_NODE_RCCL"] = "1"
from portrayer_amd import _hip as H
import device_glue
from example_scenes import EXAMPLES
from scene_dsl import default_background
scene, cam, _ = EXAMPLES["primitives-simple"]()
ds = device_glue.DeviceScene(scene, H.TRAVERSE_FLAT)
lib = H.lib()
w, h = 160, 96
bg = default_background(w, h)
ctx = H.Context(); ds.upload(ctx)
one, _, _ = device_glue.render(ctx, cam, w, h, samples=2, seed=1, sample_mode=H.SAMPLE_RNG)
ctx.close()
node = C.c_void_p(); devs = (C.c_int32 * 1)(0)
assert lib.pt_node_create(1, devs, C.byref(node)) == 0
assert lib.pt_node_uses_rccl(node) == 1
assert lib.pt_node_scene_upload(node, C.byref(ds.struct), H.TRAVERSE_FLAT, None) == 0
p = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), 2, 1, H.SAMPLE_RNG, 1, 0, 1, 0)
assert lib.pt_node_upload_background(node, bg.ctypes.data_as(H._dp), C.byref(p), None) == 0
camera = device_glue.camera_struct(cam, w, h)
assert lib.pt_node_frame_begin(node, C.byref(camera), C.byref(p)) == 0, lib.pt_node_last_error(node)
assert lib.pt_node_frame_begin(node, C.byref(camera), C.byref(p)) == 0, lib.pt_node_last_error(node)
assert lib.pt_node_frame_end(node, None) == 0 and lib.pt_node_frame_end(node, None) == 0, lib.pt_node_last_error(node)
img = np.zeros((h, w, 3), dtype=np.uint8)
assert lib.pt_node_download_image(node, C.byref(p), img.ctypes.data_as(H._u8p)) == 0
assert np.array_equal(img, one)
lib.pt_node_destroy(node)
y = (x * 2).sum().item(); assert y == 2048.0   # torch's side still works afterwards
print("rccl beside torch ok")
""" % (root, root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl beside torch ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_host_renderer_on_a_node(host, H, monkeypatch):
    """PORTRAYER_DEVICES puts detail::Renderer (what Image::render builds) on a node: same picture."""
    sc = host.Scene.example("macho-cows", assets=ASSETS)
    w, h = 160, 90
    bg = default_background(w, h)
    r1 = host.Renderer(sc, H.TRAVERSE_HIER)
    a, _, _ = r1.render(sc.camera, w, h, bg, samples=2, seed=1, sample_mode=H.SAMPLE_RNG, want_linear=False)
    r1.close()
    monkeypatch.setenv("PORTRAYER_DEVICES", "0,0,0,0")
    r4 = host.Renderer(sc, H.TRAVERSE_HIER)
    assert r4.ranks == 4
    b, _, _ = r4.render(sc.camera, w, h, bg, samples=2, seed=1, sample_mode=H.SAMPLE_RNG, want_linear=False)
    r4.close()
    assert np.array_equal(a, b)


def _bench_line(*args):
    """bench.py as the driver starts it (a subprocess; with --gpus N > 1 it spawns its own rank processes), the JSON line parsed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "2", "--warmup", "1", *args],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_two_ranks_go_through_pt_node():
    """`bench.py --gpus 2` times the product's own multi-GPU path: rank 0 drives pt_node_render_resident, the JSON line says which
    communicator carried the frame. Two ranks on one GPU here (device-to-device copies instead of RCCL: RCCL needs distinct GPUs)."""
    d = _bench_line("--gpus", "2", "--same-device", "--workload", "primitives", "--check")
    c = d["config"]["collective"]
    assert d["n_gpus"] == 2 and c["via"].startswith("pt_node_render_resident") and c["ranks_in_group"] == 2 and c["devices"] == [0, 0]
    assert c["uses_rccl"] is False and c["rank_processes"] == 2
    assert d["config"]["assembled_image_equals_single_gpu_render"] is True
    assert d["value"] > 0 and d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1.5
    assert d["config"]["rays"]["primary"] == 800 * 600


def test_bench_two_ranks_via_torch_cross_check():
    """The same partition with one process per GPU and torch.distributed's gather (gloo here: two ranks share the GPU)."""
    d = _bench_line("--gpus", "2", "--same-device", "--via", "torch", "--backend", "gloo", "--workload", "primitives", "--check")
    c = d["config"]["collective"]
    assert c["via"].startswith("torch.distributed") and c["ranks_in_group"] == 2
    assert d["config"]["assembled_image_equals_single_gpu_render"] is True


def test_bench_single_gpu_line_carries_the_contract_fields():
    d = _bench_line("--workload", "cows")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"].startswith("T lane-ops/s") and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert "pt_render_simple_kernel<1, false, false, 4, false>" in r["kernel"], "macho-cows: plain meshes, nothing reflective, untextured: 4 waves"
    assert r["hbm"]["needed_bytes"] > 0 and d["config"]["collective"] is None
