"""Round 4, call c58: c56 / c57's wrong counting render (mode 2, interpreter variant 0) under three builds: arguments not re-read, per-lane walks, -O1."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import host_glue
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene
def run(seed, env, w=16, h=8):
    os.environ.pop("PORTRAYER_PARK", None); os.environ.update(env)
    scene, cam = random_scene(seed)
    r = host.Renderer(host_glue.host_scene(scene), H.TRAVERSE_KD, kd_depth=6)
    return r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=1, seed=seed, sample_mode=H.SAMPLE_CENTRE, stats=True)
for seed in (2, 3):
    b_rgb, _, b = run(seed, {})
    rgb, _, st = run(seed, {"PORTRAYER_PARK": "0"})
    print(sys.argv[1], "seed", seed, "variant", st["kernel_variant"], "hits", st["hits"], "vs", b["hits"], "pixels differing", int((rgb != b_rgb).any(axis=2).sum()), flush=True)
