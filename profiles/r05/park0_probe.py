"""Round 5: the wrong counting render of round 4 (mode 2 = k-d semantics with KDMesh trees, interpreter variant 0 = PORTRAYER_PARK=0, counting
instantiation, built with -DPT_ARGS_AGAIN_EVERYWHERE) under one diagnostic build per call of this script: hits of the PARK=0 render against the
default (parked) render of the same scene, twice each (run-to-run variation), with the work queues and with batches."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import host_glue
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene
def run(seed, env, w=16, h=8, stats=True):
    for k in ("PORTRAYER_PARK", "PORTRAYER_FINE_QUEUES"): os.environ.pop(k, None)
    os.environ.update(env)
    scene, cam = random_scene(seed)
    r = host.Renderer(host_glue.host_scene(scene), H.TRAVERSE_KD, kd_depth=6)
    out = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=1, seed=seed, sample_mode=H.SAMPLE_CENTRE, stats=stats)
    r.close()
    return out
tag = sys.argv[1]
sizes = [(16, 8)] if len(sys.argv) < 3 else [tuple(int(v) for v in s.split("x")) for s in sys.argv[2:]]
for (w, h) in sizes:
    for seed in (2, 3):
        b_rgb, _, b = run(seed, {}, w, h)
        row = []
        for env in ({"PORTRAYER_PARK": "0"}, {"PORTRAYER_PARK": "0"}, {"PORTRAYER_PARK": "0", "PORTRAYER_FINE_QUEUES": "0"}):
            rgb, _, st = run(seed, env, w, h)
            row.append("%d/%d px %d" % (st["hits"], b["hits"], int((rgb != b_rgb).any(axis=2).sum())))
        p_rgb, _, p = run(seed, {"PORTRAYER_PARK": "0"}, w, h, stats=False)
        print("%-12s %dx%d seed %d variant %d  counting: %s | plain: px %d" % (tag, w, h, seed, st["kernel_variant"], " ; ".join(row), int((p_rgb != b_rgb).any(axis=2).sum())), flush=True)
