"""Round 5: every counter of the wrong counting render beside the right one's (same scene, default parked variant), five runs, 8x8 (one work item) and 16x8."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import host_glue
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene
def run(seed, env, w, h):
    for k in ("PORTRAYER_PARK", "PORTRAYER_FINE_QUEUES"): os.environ.pop(k, None)
    os.environ.update(env)
    scene, cam = random_scene(seed)
    r = host.Renderer(host_glue.host_scene(scene), H.TRAVERSE_KD, kd_depth=6)
    out = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=1, seed=seed, sample_mode=H.SAMPLE_CENTRE, stats=True)
    r.close()
    return out
keys = ["primary", "shadow", "reflect", "refract", "hits", "n_inner", "n_leaf", "n_analytic", "n_tri", "n_bbox", "kd_plane_miss", "depth11_skipped", "stack_overflow"]
for (w, h) in ((8, 8), (16, 8)):
    for seed in (2, 3):
        _, _, b = run(seed, {}, w, h)
        print(sys.argv[1], "%dx%d seed %d" % (w, h, seed), "right:", " ".join("%s=%d" % (k, b[k]) for k in keys), flush=True)
        for rep in range(5):
            _, _, st = run(seed, {"PORTRAYER_PARK": "0"}, w, h)
            print("      PARK=0 run %d:" % rep, " ".join("%s=%d" % (k, st[k]) for k in keys), flush=True)
