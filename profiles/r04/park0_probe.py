"""Round 4, call c56: PORTRAYER_PARK=0 (every parked frame in HBM) in the k-d semantics lost hits in tests/test_gpu_render_parity.py::test_random_scene_matches_oracle[kd-2] (c55).
Which scenes / kernels, and does the size of the LDS stack area matter? usage: python3 profiles/r04/park0_probe.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import host_glue
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import default_background
from test_gpu_render_parity import random_scene

def run(seed, env):
    for k in ("PORTRAYER_PARK", "PORTRAYER_LDS_STACK", "PORTRAYER_LDS_BUDGET_KB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    scene, cam = random_scene(seed)
    hs = host_glue.host_scene(scene)
    r = host.Renderer(hs, H.TRAVERSE_KD, kd_depth=6)
    w, h = 112, 80
    rgb, linear, st = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=1, seed=seed, sample_mode=H.SAMPLE_CENTRE, stats=True)
    return rgb, st

for seed in range(8):
    base_rgb, base = run(seed, {})
    for env in ({"PORTRAYER_PARK": "0"}, {"PORTRAYER_PARK": "0", "PORTRAYER_LDS_STACK": "8"}, {"PORTRAYER_PARK": "0", "PORTRAYER_LDS_STACK": "24"}, {"PORTRAYER_LDS_STACK": "40"}, {"PORTRAYER_LDS_BUDGET_KB": "80"}):
        try:
            rgb, st = run(seed, env)
            same = bool((rgb == base_rgb).all()) and all(st[k] == base[k] for k in ("primary", "shadow", "reflect", "refract", "hits"))
            print(seed, env, "mode", st["kernel_mode"], "variant", st["kernel_variant"], "OK" if same else "DIFFERENT: shadow %d vs %d, hits %d vs %d" % (st["shadow"], base["shadow"], st["hits"], base["hits"]), flush=True)
        except Exception as e:
            print(seed, env, "ERROR", str(e)[:120], flush=True)
