"""Round 4, call c57: narrowing down c56 (mode 2 = k-d semantics with KDMesh trees, interpreter variant 0 = PORTRAYER_PARK=0 loses hits)."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import host_glue
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene

KEYS = ("PORTRAYER_PARK", "PORTRAYER_LDS_STACK", "PORTRAYER_LDS_BUDGET_KB", "PORTRAYER_KD_CULL", "PORTRAYER_FINE_QUEUES", "PORTRAYER_STACK_CAP", "PORTRAYER_BLOCKS_PER_CU", "PORTRAYER_LANE_CHUNKS")
def run(seed, env, stats=True, w=112, h=80):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    scene, cam = random_scene(seed)
    r = host.Renderer(host_glue.host_scene(scene), H.TRAVERSE_KD, kd_depth=6)
    return r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=1, seed=seed, sample_mode=H.SAMPLE_CENTRE, stats=stats)

seed = 2
base_rgb, _, base = run(seed, {})
print("base", {k: base[k] for k in ("primary", "shadow", "reflect", "refract", "hits", "kernel_mode", "kernel_variant")})
for env in ({"PORTRAYER_PARK": "0"}, {"PORTRAYER_PARK": "0", "PORTRAYER_KD_CULL": "0"}, {"PORTRAYER_PARK": "0", "PORTRAYER_FINE_QUEUES": "0"}, {"PORTRAYER_PARK": "0", "PORTRAYER_BLOCKS_PER_CU": "1"},
            {"PORTRAYER_PARK": "0", "PORTRAYER_LDS_BUDGET_KB": "160"}, {"PORTRAYER_PARK": "0", "PORTRAYER_LANE_CHUNKS": "1"}):
    for stats in (True, False):
        try:
            rgb, _, st = run(seed, env, stats)
            bad = int((rgb != base_rgb).any(axis=2).sum())
            print(env, "stats" if stats else "plain", "variant", st["kernel_variant"], {k: st[k] for k in ("shadow", "reflect", "refract", "hits")} if stats else "", "pixels differing from the default render:", bad, flush=True)
        except Exception as e:
            print(env, stats, "ERROR", str(e)[:160], flush=True)
# tiny frames: one wavefront's worth
for (w, h) in ((8, 8), (16, 8), (64, 8)):
    b_rgb, _, b = run(seed, {}, True, w, h)
    rgb, _, st = run(seed, {"PORTRAYER_PARK": "0"}, True, w, h)
    print((w, h), "hits", st["hits"], "vs", b["hits"], "pixels differing", int((rgb != b_rgb).any(axis=2).sum()), flush=True)
